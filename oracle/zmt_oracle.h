/*
 * zmt_oracle.h -- CPU restatement of the lz4-mt hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (zstdmt_amd/, include/) may link, import or call this code.
 * It is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker for the HIP path.
 *
 * What it restates (reference = /root/reference, mcmilk/zstdmt @ 2025-10-17):
 *   - the per-chunk record framing of lib/lz4-mt_compress.c:207-310 (pt_compress) and the
 *     record parser of lib/lz4-mt_decompress.c:192-388 (pt_read / pt_decompress);
 *   - the third-party arithmetic those call into, which is NOT in the reference tree:
 *       lz4 v1.9.4 (pinned at programs/Makefile:9): LZ4F_compressFrame (lz4-mt_compress.c:281),
 *       LZ4F_decompress (lz4-mt_decompress.c:350), LZ4F_compressFrameBound (lz4-mt_compress.c:232),
 *       and XXH32 underneath.  The published algorithm is restated from the LZ4 block/frame format
 *       specifications and SURVEY.md Appendix A/B.
 *
 * Parity pinning: the oracle is checked against streams produced by the reference's own
 * lib/lz4-mt_*.c compiled in place against the image's liblz4 1.9.3 (oracle/ref/Makefile ->
 * oracle/_ref/), see tests/golden/ and tests/test_oracle_vs_ref.py.  Version skew (1.9.3 here vs
 * the 1.9.4 pin) is stated in DESIGN.md.
 */
#ifndef ZMT_ORACLE_H
#define ZMT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZO_SKIP_MAGIC   0x184D2A50u   /* lib/lz4-mt.h:33  */
#define ZO_LZ4F_MAGIC   0x184D2204u   /* lib/lz4-mt.h:32  */
#define ZO_BLOCK_MAX    65536u        /* LZ4F blockSizeID 4 (default for prefs.blockSizeID = 0) */

/* XXH32, seed-able (upstream xxhash as vendored by lz4). */
uint32_t zo_xxh32(const void *data, size_t len, uint32_t seed);

/* Scratch state for the LZ4 "fast" block encoder: 16 KiB table shared by both table modes. */
typedef struct {
	uint32_t tab[4096];
} zo_lz4_table;

/*
 * Encode one block.  `chunk` is the start of the chunk the block belongs to,
 * [pos, pos+len) the block inside it.  u16_mode selects the byU16 table (single block <= 64 KiB,
 * fresh table) instead of byU32 (linked blocks, table carried across the chunk's blocks).
 * Returns compressed size, or 0 when the output would not fit in `cap` bytes (block is then
 * stored raw by the frame layer).  The table keeps every insertion made even on failure.
 */
size_t zo_lz4_block_encode(zo_lz4_table *t, const uint8_t *chunk, size_t pos, size_t len,
			   uint8_t *dst, size_t cap, int u16_mode);

/*
 * Decode one LZ4 block into out[opos, ...).  Matches may reach back to out[0] (linked blocks);
 * `out_limit` is the highest position that may be written (opos + 64 KiB or the content size).
 * Returns the new output position, or (size_t)-1 on malformed input.
 */
size_t zo_lz4_block_decode(const uint8_t *src, size_t slen, uint8_t *out, size_t opos,
			   size_t out_limit);

/* LZ4F_compressFrameBound for the prefs of lib/lz4-mt_compress.c:141-146. */
size_t zo_lz4f_bound(size_t n);

/* One LZ4 frame exactly as LZ4F_compressFrame emits it for those prefs at level 1..2. */
size_t zo_lz4f_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);

/*
 * Decode one whole LZ4 frame (what LZ4F_decompress does for lz4-mt_decompress.c:349-362).
 * Returns the number of content bytes or (size_t)-1 on any error (bad magic/version/HC, block
 * overrun, content-size mismatch, checksum mismatch, trailing bytes).
 */
size_t zo_lz4f_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap);

/* Content-size field of a frame (LE64 at +6), what lz4-mt_decompress.c:333-334 reads. */
uint64_t zo_lz4f_content_size(const uint8_t *frame, size_t slen);

/*
 * Whole MT stream, in-memory (the pt_compress loop with fn_read = memcpy).
 * Empty input still yields one (empty-frame) record: lz4-mt_compress.c:265.
 * Returns stream bytes, or (size_t)-1 if cap is too small.
 */
size_t zo_lz4mt_compress(const uint8_t *src, size_t n, size_t chunk, uint8_t *dst, size_t cap);
size_t zo_lz4mt_compress_bound(size_t n, size_t chunk);
/* LZ4 "HC" levels (lz4hc_oracle.c): hash-chain parser, levels 3..8 */
int zo_lz4hc_level_supported(int level);
size_t zo_lz4f_compress_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int level);
size_t zo_lz4mt_compress_level(const uint8_t *src, size_t n, size_t chunk, uint8_t *dst, size_t cap, int level);

/* Inverse: walk the records (12-byte skippable header each), decode every frame.
 * Returns content bytes, (size_t)-1 on malformed input, (size_t)-2 if cap too small. */
size_t zo_lz4mt_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap);

/* Multi-threaded variants for the CPU baseline ("port" kind): T pthreads over chunks. */
size_t zo_lz4mt_compress_mt(const uint8_t *src, size_t n, size_t chunk, uint8_t *dst, size_t cap,
			    int threads);
size_t zo_lz4mt_decompress_mt(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap,
			      int threads);

/* ---- zstd-mt decode path (zstd_oracle.c; restates RFC 8878 + lib/zstd-mt_decompress.c) ------ */
#define ZO_ZSTD_MAGIC    0xFD2FB528u            /* lib/zstd-mt.h:33 (MAGICNUMBER_MAX) */
#define ZO_ZSTD_UNKNOWN  0xFFFFFFFFFFFFFFFFull  /* frame carries no content size */
#define ZO_ZSTD_ERROR    0xFFFFFFFFFFFFFFFEull  /* not a zstd frame header */

uint64_t zo_xxh64(const void *data, size_t len, uint64_t seed);

/* Frame_Content_Size of a zstd frame header, ZO_ZSTD_UNKNOWN or ZO_ZSTD_ERROR. */
uint64_t zo_zstd_frame_content_size(const uint8_t *frame, size_t slen);

/* Decode one zstd frame (no dictionary).  Returns content bytes or (size_t)-1; *consumed (may be
 * NULL) receives the frame's length in src. */
size_t zo_zstd_decompress_frame(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap,
				size_t *consumed);

/* Whole zstd-mt stream ("pzstd style" records, what lib/zstd-mt_compress.c:296-302 writes). */
size_t zo_zstdmt_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap);

/* ---- brotli-mt decompress (brotli_oracle.c).  `blob` = zstdmt_amd/csrc/data/brotli_static.bin ---- */
/* one raw brotli stream (RFC 7932); returns decoded size, -1 malformed / truncated, -2 does not fit */
long zo_brotli_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, const uint8_t *blob);
/* dictionary word transform t (RFC 7932 Appendix B); returns bytes written to dst (<= len + 13) */
size_t zo_brotli_transform(const uint8_t *blob, uint8_t *dst, const uint8_t *word, uint32_t len, uint32_t t);
/* whole brotli-mt stream of 16-byte-header records (lib/brotli-mt_decompress.c:187-377) */
size_t zo_brotlimt_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap, const uint8_t *blob);

/* snappy (snappy_oracle.c): varint preamble, one raw stream, the snappy-mt record walk
 * (lib/snappy-mt_decompress.c:185-377); (size_t)-1 / 0 on malformed input */
size_t zo_snappy_uncompressed_length(const uint8_t *src, size_t n, uint32_t *out);
size_t zo_snappy_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
size_t zo_snappymt_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
