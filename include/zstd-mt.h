/*
 * zstd-mt.h -- drop-in C API of the zstd-mt library, served by the MI355X engine.
 *
 * ABI-compatible with mcmilk/zstdmt's lib/zstd-mt.h (reference: /root/reference/lib/zstd-mt.h:27-205):
 * same symbols (ZSTDCB_*), struct layouts, enum values, callback protocol and wire format
 * (records = 12-byte 0x184D2A50 skippable header + one zstd frame, lib/zstd-mt_compress.c:296-302).
 * A program written against the reference header (programs/zstd-mt.c:10-43) compiles against this
 * one unchanged and links to libzstdmt_amd.so instead of the pthread library + libzstd.
 *
 * The bar for this codec is decompress-identical, not byte-identical (zstd's compressed bytes are
 * version dependent):
 *   - ZSTDCB_decompressDCtx decodes what the reference writes, at any level, on the device;
 *   - ZSTDCB_compressCCtx writes valid zstd frames that the reference (and any zstd) decodes to the
 *     input; `level` is validated (1..22) and selects one of the device encoder's three tiers (1-2 / 3-9 /
 *     10-22: table size and hash length, zstdmt_amd/csrc/hip/zstd_enc.hip) and the default chunk size.
 * Differences a caller can observe are listed in INTEGRATION.md.
 */
#ifndef ZSTDCB_H
#define ZSTDCB_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* limits and magics -- reference lib/zstd-mt.h:27-35 */
#define ZSTDCB_THREAD_MAX       128
#define ZSTDCB_LEVEL_MIN        1
#define ZSTDCB_LEVEL_MAX        22
#define ZSTDCB_MAGICNUMBER_V01  0x1EB52FFDU
#define ZSTDCB_MAGICNUMBER_MIN  0xFD2FB522U
#define ZSTDCB_MAGICNUMBER_MAX  0xFD2FB528U
#define ZSTDCB_MAGIC_SKIPPABLE  0x184D2A50U

/* ---- errors: a size_t result r is an error iff ZSTDCB_isError(r); code = (size_t)-enum ----------
 * reference lib/zstd-mt.h:41-61, lib/zstd-mt_common.c:20-62 */
typedef enum {
	ZSTDCB_error_no_error,
	ZSTDCB_error_memory_allocation,
	ZSTDCB_error_init_missing,
	ZSTDCB_error_read_fail,
	ZSTDCB_error_write_fail,
	ZSTDCB_error_data_error,
	ZSTDCB_error_frame_compress,
	ZSTDCB_error_frame_decompress,
	ZSTDCB_error_compressionParameter_unsupported,
	ZSTDCB_error_compression_library,
	ZSTDCB_error_canceled,
	ZSTDCB_error_maxCode
} ZSTDCB_ErrorCode;

extern size_t zstdmt_errcode; /* last codec-level failure (here: a GPUMT_ST_* status word) */

#define ZSTDCB_PREFIX(name) ZSTDCB_error_##name
#define ZSTDCB_ERROR(name)  ((size_t)-ZSTDCB_PREFIX(name))
extern unsigned ZSTDCB_isError(size_t code);
extern const char *ZSTDCB_getErrorString(size_t code);

/* ---- buffers and callbacks -- reference lib/zstd-mt.h:67-93, lib/README.md:19-24 ---------------
 * fn_read : the library sets in->size to the bytes it wants in in->buf; the callee stores what it
 *           got in in->size (0 = end of input).  fn_write: the callee must take out->size bytes.
 * Return 0 on success, -1 read/write error, -2 cancelled, -3 out of memory. */
typedef struct {
	void *buf;
	size_t size;
	size_t allocated;
} ZSTDCB_Buffer;

typedef int (fn_read)(void *args, ZSTDCB_Buffer *in);
typedef int (fn_write)(void *args, ZSTDCB_Buffer *out);

typedef struct {
	fn_read *fn_read;
	void *arg_read;
	fn_write *fn_write;
	void *arg_write;
} ZSTDCB_RdWr_t;

/* ---- compression -- reference lib/zstd-mt.h:99-142 -------------------------------------------- */
typedef struct ZSTDCB_CCtx_s ZSTDCB_CCtx;

/* threads 1..ZSTDCB_THREAD_MAX, level ZSTDCB_LEVEL_MIN..MAX, inputsize = chunk bytes
 * (0 -> 1 << (windowLog[level] + 1), 1 MiB at level 1: lib/zstd-mt_compress.c:116-127).
 * NULL on invalid arguments or when no MI355X device can be opened. */
ZSTDCB_CCtx *ZSTDCB_createCCtx(int threads, int level, int inputsize);
size_t ZSTDCB_compressCCtx(ZSTDCB_CCtx *ctx, ZSTDCB_RdWr_t *rdwr);
size_t ZSTDCB_GetFramesCCtx(ZSTDCB_CCtx *ctx);
size_t ZSTDCB_GetInsizeCCtx(ZSTDCB_CCtx *ctx);
size_t ZSTDCB_GetOutsizeCCtx(ZSTDCB_CCtx *ctx);
void ZSTDCB_freeCCtx(ZSTDCB_CCtx *ctx);

/* ---- decompression -- reference lib/zstd-mt.h:148-205 ----------------------------------------- */
typedef struct ZSTDCB_DCtx_s ZSTDCB_DCtx;

ZSTDCB_DCtx *ZSTDCB_createDCtx(int threads, int inputsize);
size_t ZSTDCB_decompressDCtx(ZSTDCB_DCtx *ctx, ZSTDCB_RdWr_t *rdwr);
size_t ZSTDCB_GetFramesDCtx(ZSTDCB_DCtx *ctx);
size_t ZSTDCB_GetInsizeDCtx(ZSTDCB_DCtx *ctx);
size_t ZSTDCB_GetOutsizeDCtx(ZSTDCB_DCtx *ctx);
void ZSTDCB_freeDCtx(ZSTDCB_DCtx *ctx);

/* ---- the names /root/reference/lib/README.md:36-76 documents (ZSTDMT_*).  libzstd itself exports ZSTDMT_createCCtx /
 * ZSTDMT_compressCCtx / ZSTDMT_freeCCtx with other signatures (its multithreading API), which is why upstream renamed
 * the prefix to ZSTDCB_ and why this library does NOT export ZSTDMT_* symbols: a process that links both would bind one
 * library's calls to the other's functions.  A caller written against the old README opts in to header-only names, as
 * the reference's own front end does with macros (/root/reference/programs/zstd-mt.c:13-43): */
#ifdef ZSTDCB_LEGACY_ZSTDMT_NAMES
typedef ZSTDCB_Buffer ZSTDMT_Buffer;
typedef ZSTDCB_RdWr_t ZSTDMT_RdWr_t;
typedef ZSTDCB_CCtx ZSTDMT_CCtx;
typedef ZSTDCB_DCtx ZSTDMT_DCtx;
#define ZSTDMT_THREAD_MAX ZSTDCB_THREAD_MAX
#define ZSTDMT_LEVEL_MIN ZSTDCB_LEVEL_MIN
#define ZSTDMT_LEVEL_MAX ZSTDCB_LEVEL_MAX
#define ZSTDMT_isError ZSTDCB_isError
#define ZSTDMT_getErrorString ZSTDCB_getErrorString
#define ZSTDMT_createCCtx ZSTDCB_createCCtx
#define ZSTDMT_compressCCtx ZSTDCB_compressCCtx
#define ZSTDMT_GetFramesCCtx ZSTDCB_GetFramesCCtx
#define ZSTDMT_GetInsizeCCtx ZSTDCB_GetInsizeCCtx
#define ZSTDMT_GetOutsizeCCtx ZSTDCB_GetOutsizeCCtx
#define ZSTDMT_freeCCtx ZSTDCB_freeCCtx
#define ZSTDMT_createDCtx ZSTDCB_createDCtx
#define ZSTDMT_decompressDCtx ZSTDCB_decompressDCtx
#define ZSTDMT_GetFramesDCtx ZSTDCB_GetFramesDCtx
#define ZSTDMT_GetInsizeDCtx ZSTDCB_GetInsizeDCtx
#define ZSTDMT_GetOutsizeDCtx ZSTDCB_GetOutsizeDCtx
#define ZSTDMT_freeDCtx ZSTDCB_freeDCtx
#endif

#ifdef __cplusplus
}
#endif
#endif /* ZSTDCB_H */
