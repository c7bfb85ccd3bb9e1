/*
 * lz4-mt.h -- drop-in C API of the lz4-mt library, served by the MI355X engine.
 *
 * ABI-compatible with mcmilk/zstdmt's lib/lz4-mt.h (reference: /root/reference/lib/lz4-mt.h:27-159):
 * same symbols, same struct layouts, same enum values, same callback protocol, same wire format
 * (records = 12-byte 0x184D2A50 skippable header + one LZ4 frame, README.md:8-17).  A program
 * written against the reference header (e.g. programs/lz4-mt.c:13-43) compiles against this one
 * unchanged and links to libzstdmt_amd.so instead of the pthread library.
 *
 * Differences that a caller can observe are listed in INTEGRATION.md:
 *   - `threads` is validated (1..LZ4MT_THREAD_MAX) but the work runs on the GPU (the device named by
 *     the environment variable GPUMT_DEVICE, default 0).  Threading contract of the callbacks: fn_read
 *     is called from ONE library thread (the reader) and fn_write from ONE other library thread (the
 *     writer); a read and a write may run at the same time, two reads or two writes never do -- the
 *     reference's own contract (lib/lz4-mt_compress.c:256-277 vs :301-303).  With threads == 1
 *     LZ4MT_decompressDCtx runs every callback on the calling thread, as the reference does
 *     (lib/lz4-mt_decompress.c:528-534);
 *   - every level runs on the device and is bit-identical to the reference at the same level:
 *     1-2 (LZ4 "fast"), 3-9 (LZ4HC hash-chain parser), 10-12 (LZ4HC optimal parser; slow);
 *   - plain .lz4 input (no skippable frames; reference: st_decompress, lib/lz4-mt_decompress.c:391-483)
 *     is decoded on the device as well: the frames are split on the host and handed over in batches.
 */
#ifndef LZ4MT_H
#define LZ4MT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* limits and magics -- reference lib/lz4-mt.h:27-33 */
#define LZ4MT_THREAD_MAX        128
#define LZ4MT_LEVEL_MIN         1
#define LZ4MT_LEVEL_MAX         12
#define LZ4FMT_MAGICNUMBER      0x184D2204U
#define LZ4FMT_MAGIC_SKIPPABLE  0x184D2A50U

/* ---- errors: a size_t result r is an error iff LZ4MT_isError(r); code = (size_t)-enum ---------
 * reference lib/lz4-mt.h:39-61, lib/lz4-mt_common.c:16-63 */
typedef enum {
	LZ4MT_error_no_error,
	LZ4MT_error_memory_allocation,
	LZ4MT_error_read_fail,
	LZ4MT_error_write_fail,
	LZ4MT_error_data_error,
	LZ4MT_error_frame_compress,
	LZ4MT_error_frame_decompress,
	LZ4MT_error_compressionParameter_unsupported,
	LZ4MT_error_compression_library,
	LZ4MT_error_canceled,
	LZ4MT_error_maxCode
} LZ4MT_ErrorCode;

#ifdef ERROR
#undef ERROR
#endif
#define PREFIX(name) LZ4MT_error_##name
#define ERROR(name)  ((size_t)-PREFIX(name))

extern size_t lz4mt_errcode;  /* last codec-level failure (here: a GPUMT_ST_* status word) */
extern unsigned LZ4MT_isError(size_t code);
extern const char *LZ4MT_getErrorString(size_t code);

/* ---- buffers and callbacks -- reference lib/lz4-mt.h:67-89, lib/README.md:19-24 ---------------
 * fn_read : the library sets in->size to the bytes it wants in in->buf; the callee stores what it
 *           got in in->size (0 = end of input).  fn_write: the callee must take out->size bytes.
 * Return 0 on success, -1 read/write error, -2 cancelled, -3 out of memory. */
typedef struct {
	void *buf;
	size_t size;
	size_t allocated;
} LZ4MT_Buffer;

typedef int (fn_read)(void *args, LZ4MT_Buffer *in);
typedef int (fn_write)(void *args, LZ4MT_Buffer *out);

typedef struct {
	fn_read *fn_read;
	void *arg_read;
	fn_write *fn_write;
	void *arg_write;
} LZ4MT_RdWr_t;

/* ---- compression -- reference lib/lz4-mt.h:95-124 ------------------------------------------ */
typedef struct LZ4MT_CCtx_s LZ4MT_CCtx;

/* threads 1..LZ4MT_THREAD_MAX, level LZ4MT_LEVEL_MIN..MAX, inputsize = chunk bytes (0 -> 4 MiB).
 * NULL on invalid arguments or when no MI355X device can be opened. */
LZ4MT_CCtx *LZ4MT_createCCtx(int threads, int level, int inputsize);
size_t LZ4MT_compressCCtx(LZ4MT_CCtx *ctx, LZ4MT_RdWr_t *rdwr);
size_t LZ4MT_GetFramesCCtx(LZ4MT_CCtx *ctx);
size_t LZ4MT_GetInsizeCCtx(LZ4MT_CCtx *ctx);
size_t LZ4MT_GetOutsizeCCtx(LZ4MT_CCtx *ctx);
void LZ4MT_freeCCtx(LZ4MT_CCtx *ctx);

/* ---- decompression -- reference lib/lz4-mt.h:130-159 ---------------------------------------- */
typedef struct LZ4MT_DCtx_s LZ4MT_DCtx;

LZ4MT_DCtx *LZ4MT_createDCtx(int threads, int inputsize);
size_t LZ4MT_decompressDCtx(LZ4MT_DCtx *ctx, LZ4MT_RdWr_t *rdwr);
size_t LZ4MT_GetFramesDCtx(LZ4MT_DCtx *ctx);
size_t LZ4MT_GetInsizeDCtx(LZ4MT_DCtx *ctx);
size_t LZ4MT_GetOutsizeDCtx(LZ4MT_DCtx *ctx);
void LZ4MT_freeDCtx(LZ4MT_DCtx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* LZ4MT_H */
