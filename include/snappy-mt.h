/*
 * snappy-mt.h -- drop-in C API of the snappy-mt library, served by the MI355X engine.
 *
 * ABI-compatible with mcmilk/zstdmt's lib/snappy-mt.h (reference: /root/reference/lib/snappy-mt.h:
 * 14-143): same symbols (SNAPPYMT_*), struct layouts, enum values, callback protocol and wire format
 * (records = 16-byte header `LE32 0x184D2A50 | LE32 8 | LE32 csize | LE16 "SP" | LE16 hint` + one
 * raw snappy stream, lib/snappy-mt_compress.c:280-300; the decoder sizes a record's output from the
 * stream's own length preamble and does not read the hint, lib/snappy-mt_decompress.c:234-238,262-267).
 *
 *   - SNAPPYMT_decompressDCtx decodes raw snappy streams on the device, byte-identical to a snappy
 *     decoder (zstdmt_amd/csrc/hip/snappy.hip; oracle/snappy_oracle.c pinned against libsnappy 1.1.8);
 *   - SNAPPYMT_compressCCtx writes valid raw snappy that any snappy decoder decodes to the input.  The
 *     reference's snappy library (a C port vendored by the zstdmt repository) is not part of its
 *     lib/ tree, so its bytes cannot be pinned: the bar is decompress-identical.  `level` is accepted
 *     and ignored, as in the reference (lib/snappy-mt_compress.c:80,96).
 */
#ifndef SNAPPYMT_H
#define SNAPPYMT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* limits and magics -- reference lib/snappy-mt.h:14-17 */
#define SNAPPY_OK 0
#define SNAPPYMT_THREAD_MAX      128
#define SNAPPYMT_MAGICNUMBER     0x5053U /* "SP" */
#define SNAPPYMT_MAGIC_SKIPPABLE 0x184D2A50U

/* ---- errors: a size_t result r is an error iff SNAPPYMT_isError(r); code = (size_t)-enum --------
 * reference lib/snappy-mt.h:20-38, lib/snappy-mt_common.c:8-45 */
typedef enum {
	SNAPPYMT_error_no_error,
	SNAPPYMT_error_memory_allocation,
	SNAPPYMT_error_read_fail,
	SNAPPYMT_error_write_fail,
	SNAPPYMT_error_data_error,
	SNAPPYMT_error_frame_compress,
	SNAPPYMT_error_frame_decompress,
	SNAPPYMT_error_compressionParameter_unsupported,
	SNAPPYMT_error_compression_library,
	SNAPPYMT_error_canceled,
	SNAPPYMT_error_maxCode
} SNAPPYMT_ErrorCode;

#define SNAPPYMT_PREFIX(name) SNAPPYMT_error_##name
#define SNAPPYMT_ERROR(name)  ((size_t)-SNAPPYMT_PREFIX(name))
extern unsigned SNAPPYMT_isError(size_t code);
extern const char *SNAPPYMT_getErrorString(size_t code);

/* ---- buffers and callbacks -- reference lib/snappy-mt.h:46-68, lib/README.md:19-24 --------------
 * fn_read : the library sets in->size to the bytes it wants in in->buf; the callee stores what it
 *           got in in->size (0 = end of input).  fn_write: the callee must take out->size bytes.
 * Return 0 on success, -1 read/write error, -2 cancelled, -3 out of memory. */
typedef struct {
	void *buf;
	size_t size;
	size_t allocated;
} SNAPPYMT_Buffer;

typedef int (fnRead)(void *args, SNAPPYMT_Buffer *in);
typedef int (fnWrite)(void *args, SNAPPYMT_Buffer *out);

typedef struct {
	fnRead *fn_read;
	void *arg_read;
	fnWrite *fn_write;
	void *arg_write;
} SNAPPYMT_RdWr_t;

/* ---- compression -- reference lib/snappy-mt.h:74-107 ------------------------------------------- */
typedef struct SNAPPYMT_CCtx_s SNAPPYMT_CCtx;

/* threads 1..SNAPPYMT_THREAD_MAX, level unused, inputsize = chunk bytes (0 -> 64 KiB,
 * lib/snappy-mt_compress.c:99-102); NULL on invalid arguments or when no MI355X device can be opened */
SNAPPYMT_CCtx *SNAPPYMT_createCCtx(int threads, int level, int inputsize);
size_t SNAPPYMT_compressCCtx(SNAPPYMT_CCtx *ctx, SNAPPYMT_RdWr_t *rdwr);
size_t SNAPPYMT_GetFramesCCtx(SNAPPYMT_CCtx *ctx);
size_t SNAPPYMT_GetInsizeCCtx(SNAPPYMT_CCtx *ctx);
size_t SNAPPYMT_GetOutsizeCCtx(SNAPPYMT_CCtx *ctx);
void SNAPPYMT_freeCCtx(SNAPPYMT_CCtx *ctx);

/* ---- decompression -- reference lib/snappy-mt.h:113-143 ---------------------------------------- */
typedef struct SNAPPYMT_DCtx_s SNAPPYMT_DCtx;

/* NULL on invalid arguments or when no MI355X device can be opened */
SNAPPYMT_DCtx *SNAPPYMT_createDCtx(int threads, int inputsize);
size_t SNAPPYMT_decompressDCtx(SNAPPYMT_DCtx *ctx, SNAPPYMT_RdWr_t *rdwr);
size_t SNAPPYMT_GetFramesDCtx(SNAPPYMT_DCtx *ctx);
size_t SNAPPYMT_GetInsizeDCtx(SNAPPYMT_DCtx *ctx);
size_t SNAPPYMT_GetOutsizeDCtx(SNAPPYMT_DCtx *ctx);
void SNAPPYMT_freeDCtx(SNAPPYMT_DCtx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SNAPPYMT_H */
