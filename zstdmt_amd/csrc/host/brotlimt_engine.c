/*
 * brotlimt_engine.c -- the host side of brotli-mt on MI355X: BROTLIMT_* (include/brotli-mt.h) over
 * gpumt_*: mt16_engine.inc with brotli's constants (reference lib/brotli-mt_compress.c,
 * lib/brotli-mt_decompress.c).
 */
#include "brotli-mt.h"

#define MTP(x) BROTLIMT_##x
#define MT_CODEC "brotli"
#define MT_LEVEL_OK(level) ((level) >= BROTLIMT_LEVEL_MIN && (level) <= BROTLIMT_LEVEL_MAX)
#define MT_DEFAULT_CHUNK(level) (1024 * 1024 * ((level) ? (level) : 1)) /* lib/brotli-mt_compress.c:105-109 */
#define MT_SLOT_STRIDE(chunk) gpumt_zstd_slot_stride(chunk)
/* the quality reaches the encoder as the reference hands it to BrotliEncoderCompress (lib/brotli-mt_compress.c:269-272) */
#define MT_COMPRESS_BATCH gpumt_brotli_compress_batch_level
#define MT_DECOMPRESS_BATCH gpumt_brotli_decompress_batch
#define MT_CAP_FROM_PREAMBLE 0 /* capacity = hint << 16, lib/brotli-mt_decompress.c:236-239 */

#include "mt16_engine.inc"
