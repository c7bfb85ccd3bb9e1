/*
 * brotli-mt.h -- drop-in C API of the brotli-mt library, served by the MI355X engine.
 *
 * ABI-compatible with mcmilk/zstdmt's lib/brotli-mt.h (reference: /root/reference/lib/brotli-mt.h:
 * 27-157): same symbols (BROTLIMT_*), struct layouts, enum values, callback protocol and wire format
 * (records = 16-byte header `LE32 0x184D2A50 | LE32 8 | LE32 csize | LE16 "BR" | LE16 hint` + one
 * raw brotli stream, lib/brotli-mt_compress.c:285-304; hint = 64 KiB units of output the decoder
 * must provide, lib/brotli-mt_decompress.c:236-239).
 *
 *   - BROTLIMT_decompressDCtx decodes what the reference writes, at any level 0..11, on the device
 *     (complete RFC 7932 decoder, zstdmt_amd/csrc/hip/brotli_dec.hip);
 *   - BROTLIMT_compressCCtx writes valid brotli streams that the reference (and any brotli decoder)
 *     decodes to the input: the bar for this codec is decompress-identical (brotli's bytes are version
 *     dependent); `level` is validated (0..11), sets the default chunk size and selects one of the device
 *     encoder's three tiers (0-3 / 4-8 / 9-11, zstdmt_amd/csrc/hip/brotli_enc.hip).
 */
#ifndef BROTLIMT_H
#define BROTLIMT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* limits and magics -- reference lib/brotli-mt.h:27-33 */
#define BROTLIMT_THREAD_MAX      128
#define BROTLIMT_LEVEL_MIN       0
#define BROTLIMT_LEVEL_MAX       11
#define BROTLIMT_MAGICNUMBER     0x5242U /* "BR" */
#define BROTLIMT_MAGIC_SKIPPABLE 0x184D2A50U

/* ---- errors: a size_t result r is an error iff BROTLIMT_isError(r); code = (size_t)-enum --------
 * reference lib/brotli-mt.h:39-59, lib/brotli-mt_common.c:24-58 */
typedef enum {
	BROTLIMT_error_no_error,
	BROTLIMT_error_memory_allocation,
	BROTLIMT_error_read_fail,
	BROTLIMT_error_write_fail,
	BROTLIMT_error_data_error,
	BROTLIMT_error_frame_compress,
	BROTLIMT_error_frame_decompress,
	BROTLIMT_error_compressionParameter_unsupported,
	BROTLIMT_error_compression_library,
	BROTLIMT_error_canceled,
	BROTLIMT_error_maxCode
} BROTLIMT_ErrorCode;

#define BROTLIMT_PREFIX(name) BROTLIMT_error_##name
#define BROTLIMT_ERROR(name)  ((size_t)-BROTLIMT_PREFIX(name))
extern unsigned BROTLIMT_isError(size_t code);
extern const char *BROTLIMT_getErrorString(size_t code);

/* ---- buffers and callbacks -- reference lib/brotli-mt.h:65-87, lib/README.md:19-24 --------------
 * fn_read : the library sets in->size to the bytes it wants in in->buf; the callee stores what it
 *           got in in->size (0 = end of input).  fn_write: the callee must take out->size bytes.
 * Return 0 on success, -1 read/write error, -2 cancelled, -3 out of memory. */
typedef struct {
	void *buf;
	size_t size;
	size_t allocated;
} BROTLIMT_Buffer;

typedef int (fn_read)(void *args, BROTLIMT_Buffer *in);
typedef int (fn_write)(void *args, BROTLIMT_Buffer *out);

typedef struct {
	fn_read *fn_read;
	void *arg_read;
	fn_write *fn_write;
	void *arg_write;
} BROTLIMT_RdWr_t;

/* ---- compression -- reference lib/brotli-mt.h:93-126 ------------------------------------------ */
typedef struct BROTLIMT_CCtx_s BROTLIMT_CCtx;

/* threads 1..BROTLIMT_THREAD_MAX, level 0..11, inputsize = chunk bytes (0 -> 1 MiB x max(level, 1),
 * lib/brotli-mt_compress.c:105-109); NULL on invalid arguments or when no MI355X device can be opened */
BROTLIMT_CCtx *BROTLIMT_createCCtx(int threads, int level, int inputsize);
size_t BROTLIMT_compressCCtx(BROTLIMT_CCtx *ctx, BROTLIMT_RdWr_t *rdwr);
size_t BROTLIMT_GetFramesCCtx(BROTLIMT_CCtx *ctx);
size_t BROTLIMT_GetInsizeCCtx(BROTLIMT_CCtx *ctx);
size_t BROTLIMT_GetOutsizeCCtx(BROTLIMT_CCtx *ctx);
void BROTLIMT_freeCCtx(BROTLIMT_CCtx *ctx);

/* ---- decompression -- reference lib/brotli-mt.h:132-157 ---------------------------------------- */
typedef struct BROTLIMT_DCtx_s BROTLIMT_DCtx;

/* NULL on invalid arguments or when no MI355X device can be opened */
BROTLIMT_DCtx *BROTLIMT_createDCtx(int threads, int inputsize);
size_t BROTLIMT_decompressDCtx(BROTLIMT_DCtx *ctx, BROTLIMT_RdWr_t *rdwr);
size_t BROTLIMT_GetFramesDCtx(BROTLIMT_DCtx *ctx);
size_t BROTLIMT_GetInsizeDCtx(BROTLIMT_DCtx *ctx);
size_t BROTLIMT_GetOutsizeDCtx(BROTLIMT_DCtx *ctx);
void BROTLIMT_freeDCtx(BROTLIMT_DCtx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* BROTLIMT_H */
