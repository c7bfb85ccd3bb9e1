/*
 * snappymt_engine.c -- the host side of snappy-mt on MI355X: SNAPPYMT_* (include/snappy-mt.h) over
 * gpumt_*: mt16_engine.inc with snappy's constants (reference lib/snappy-mt_compress.c,
 * lib/snappy-mt_decompress.c).
 */
#include "snappy-mt.h"

#define MTP(x) SNAPPYMT_##x
#define MT_CODEC "snappy"
#define MT_LEVEL_OK(level) ((void)(level), 1)         /* "None level", lib/snappy-mt_compress.c:96 */
#define MT_DEFAULT_CHUNK(level) ((void)(level), 1024 * 64) /* SNAPPY_IN_ALLOC_SIZE, :12,:102 */
#define MT_SLOT_STRIDE(chunk) gpumt_snappy_slot_stride(chunk)
#define MT_COMPRESS_BATCH(g, in, n, chunk, slots, stride, lens, level, stream) \
	((void)(level), gpumt_snappy_compress_batch(g, in, n, chunk, slots, stride, lens, stream))
#define MT_DECOMPRESS_BATCH gpumt_snappy_decompress_batch
#define MT_CAP_FROM_PREAMBLE 1 /* snappy_uncompressed_length, lib/snappy-mt_decompress.c:262-267 */

#include "mt16_engine.inc"
