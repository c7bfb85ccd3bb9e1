/*
 * gpumt.h -- C ABI of the MI355X device engine that replaces zstdmt's per-chunk codec dispatch.
 *
 * This is the *internal* boundary the reference does not have (last row):
 * the host library behind LZ4MT_* (include/lz4-mt.h) hands batches of chunks / records to the GPU
 * through these calls, exactly where the reference's worker threads call into liblz4:
 *
 *   gpumt_lz4_compress_batch   replaces  LZ4F_compressFrame    @ lib/lz4-mt_compress.c:281  (C2)
 *   gpumt_lz4_compress_batch_level  the same call with prefs.compressionLevel >= 3 (HC, :141-146)
 *                                        + header emit          @ lib/lz4-mt_compress.c:294-298 (F5)
 *   gpumt_lz4_slot_stride      replaces  LZ4F_compressFrameBound@ lib/lz4-mt_compress.c:232,244 (C1)
 *   gpumt_lz4_compact          replaces  pt_write ordering      @ lib/lz4-mt_compress.c:178-205 (F4)
 *   gpumt_lz4_decompress_batch replaces  LZ4F_decompress        @ lib/lz4-mt_decompress.c:349-362 (C3)
 *                                        + size probe           @ lib/lz4-mt_decompress.c:329-334 (F10)
 *
 * Plain C types only: opaque handle, device pointers as void*, sizes as integers.  Nothing here
 * falls back to the CPU: every entry point returns GPUMT_E_NODEVICE/E_HIP when the HIP runtime or
 * the gfx950 device is unavailable.
 *
 * All device pointers must come from gpumt_malloc() of the same handle (or any allocation of the
 * same HIP device).  "stream" arguments are small integers 0..GPUMT_NSTREAMS-1 naming the
 * handle's own HIP streams (0 = compute, 1 = H2D, 2 = D2H in the host pipeline).
 */
#ifndef GPUMT_H
#define GPUMT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* streams 0..3: kernels / H2D / small D2H / bulk D2H of the host engines' pipeline; 4..11: one extra
 * kernel stream per batch slot, so that the batches of a pipeline overlap on the device (every
 * launching stream has its own internal scratch) */
#define GPUMT_NSTREAMS 16

enum {
	GPUMT_OK = 0,
	GPUMT_E_NODEVICE = -1, /* no HIP device / runtime */
	GPUMT_E_HIP = -2,      /* a HIP call failed; see gpumt_last_error() */
	GPUMT_E_ARG = -3,
	GPUMT_E_NOMEM = -4
};

/* per-record status words written by gpumt_lz4_decompress_batch (0 = frame decoded and verified) */
enum {
	GPUMT_ST_OK = 0,
	GPUMT_ST_BAD_RECORD = 1,   /* skippable header wrong (magic / len != 4 / csize)      -> data_error  */
	GPUMT_ST_BAD_FRAME = 2,    /* LZ4F magic / version / reserved bits / header checksum -> compression_library */
	GPUMT_ST_BAD_BLOCK = 3,    /* malformed block (overrun, zero offset, offset too far)  */
	GPUMT_ST_SIZE_MISMATCH = 4,/* decoded bytes != content-size field                     */
	GPUMT_ST_BAD_CHECKSUM = 5, /* XXH32 content checksum mismatch                         */
	GPUMT_ST_TRAILING = 6,     /* record longer than the frame it carries -> frame_decompress */
	GPUMT_ST_UNSUPPORTED = 7   /* valid LZ4F feature the device path does not decode (block checksum, dictID) */
};

typedef struct gpumt_ctx gpumt_ctx;

/* ---- lifetime ---------------------------------------------------------------------------- */
int  gpumt_device_count(void);
/* device: HIP device index, or GPUMT_DEVICE_DEFAULT = the index in the environment variable
 * GPUMT_DEVICE (0 when unset) -- what the LZ4MT_* / ZSTDCB_* / BROTLIMT_* contexts use */
#define GPUMT_DEVICE_DEFAULT (-1)
int  gpumt_open(int device, gpumt_ctx **out);
void gpumt_close(gpumt_ctx *h);
const char *gpumt_last_error(gpumt_ctx *h);
const char *gpumt_device_name(gpumt_ctx *h);
/* NUMA node of the host the device's PCIe function hangs off (sysfs numa_node), -1 when the host does not say.  Pinned
 * memory from gpumt_host_alloc lives on that node: a host thread that fills or drains it copies 25-30 % faster from
 * there (profiles/r06_sweeps/api_numa.txt), which is why the host engines' reader / writer threads bind themselves to it */
int gpumt_host_node(gpumt_ctx *h);

/* ---- memory / transfers / sync ------------------------------------------------------------ */
/* Freed buffers go to process-wide caches (device: GPUMT_DEVICE_CACHE_MB, pinned host:
 * GPUMT_PINNED_CACHE_MB, 16384 each by default; 0 disables) and from there to the next allocation of a
 * similar size, without a device-wide wait: free a buffer only when no queued work uses it. */
/* Debug guard for that contract: with GPUMT_DEBUG_FREE=1 in the environment (or gpumt_set_variant(h, "debug_free", 1))
 * gpumt_free / gpumt_host_free first ask every stream of the context whether work is still queued; if so they count the
 * call (gpumt_debug_free_busy), report it on stderr and wait for the streams before the buffer is recycled. */
void *gpumt_malloc(gpumt_ctx *h, size_t bytes);
void  gpumt_free(gpumt_ctx *h, void *dptr);
void *gpumt_host_alloc(gpumt_ctx *h, size_t bytes);          /* pinned */
void  gpumt_host_free(gpumt_ctx *h, void *hptr);
/* Pin / unpin memory the caller owns (hipHostRegister): an application buffer, or one mapping shared by the ranks
 * of a multi-GPU job into which every GPU copies its segment at its offset (bench.py --gather d2h). */
int gpumt_host_register(gpumt_ctx *h, void *p, size_t bytes);
int gpumt_host_unregister(gpumt_ctx *h, void *p);
/* Freed device and pinned buffers stay in process-wide caches (GPUMT_DEVICE_CACHE_MB / GPUMT_PINNED_CACHE_MB, 16 GiB
 * each by default) for the next context; this releases whatever is idle and returns the bytes given back. */
size_t gpumt_trim_caches(gpumt_ctx *h);
/* calls of gpumt_free / gpumt_host_free the debug guard caught with work queued, process-wide, since the last call */
unsigned long gpumt_debug_free_busy(void);
int   gpumt_memcpy_h2d(gpumt_ctx *h, void *dst, const void *src, size_t n, int stream);
int   gpumt_memcpy_d2h(gpumt_ctx *h, void *dst, const void *src, size_t n, int stream);
int   gpumt_memcpy_d2d(gpumt_ctx *h, void *dst, const void *src, size_t n, int stream);
/* Device -> pinned host memory (from gpumt_host_alloc) by a kernel on `stream`: min(n, *d_n) bytes when
 * d_n (a uint64 in device memory, e.g. the total gpumt_lz4_compact leaves at d_rec_off[nrec]) is
 * given, else n.  Both addresses 16-byte aligned.  What the host engines use for results: the size
 * stays on the device and batches on different streams overlap (see pack.hip). */
int   gpumt_push_host(gpumt_ctx *h, void *dst_host, const void *src, size_t n, const uint64_t *d_n, int stream);
int   gpumt_memset(gpumt_ctx *h, void *dst, int byte, size_t n, int stream);
int   gpumt_stream_sync(gpumt_ctx *h, int stream);
int   gpumt_device_sync(gpumt_ctx *h);
/* make `waiter` wait (on device) for everything queued so far on `signaler` */
int   gpumt_stream_wait(gpumt_ctx *h, int waiter, int signaler);
/* Markers: gpumt_mark(id, stream) records marker id (0..GPUMT_NMARKS-1) behind everything queued so
 * far on `stream`; gpumt_mark_sync(id) blocks the calling host thread until that point is reached
 * (and nothing queued later).  Thread-safe: one thread may wait on a marker while another queues
 * work. */
#define GPUMT_NMARKS 16
int   gpumt_mark(gpumt_ctx *h, int id, int stream);
int   gpumt_mark_sync(gpumt_ctx *h, int id);
/* raw hipStream_t of a stream index, for callers that interoperate (e.g. RCCL via torch) */
void *gpumt_stream_handle(gpumt_ctx *h, int stream);

/* ---- HIP-event timing on the handle's streams (bench.py's roofline leg) -------------------- */
int   gpumt_timer_start(gpumt_ctx *h, int slot, int stream);   /* slot 0..15 */
int   gpumt_timer_stop(gpumt_ctx *h, int slot, int stream);
int   gpumt_timer_ms(gpumt_ctx *h, int slot, float *ms);       /* syncs on the stop event */

/* ---- LZ4 (lz4-mt level 1..2 = LZ4 "fast", acceleration 1) ---------------------------------- */

/* Bytes one record can occupy: 12 + LZ4F_compressFrameBound(chunk), rounded up to 256. */
size_t gpumt_lz4_slot_stride(size_t chunk);

/* Number of records the reference emits for n input bytes (>= 1: empty input -> one empty frame). */
size_t gpumt_lz4_record_count(size_t n, size_t chunk);

/*
 * Compress d_in[0..n) as consecutive chunks of `chunk` bytes (last one ragged).  Record i
 * (12-byte skippable header + one LZ4 frame, byte-identical to the reference's output for that
 * chunk) is written at d_slots + i*slot_stride and its length to d_rec_len[i].
 * Uses internal scratch of 4 bytes per chunk for the XXH32 content checksums.
 */
int gpumt_lz4_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk,
			     void *d_slots, size_t slot_stride, uint32_t *d_rec_len, int stream);

/*
 * The same with lz4-mt's compression level (prefs.compressionLevel, lib/lz4-mt_compress.c:141-146):
 * 1..2 = LZ4 fast (what gpumt_lz4_compress_batch does), 3..9 = LZ4 HC hash-chain parser with
 * 4..256 searches per position (liblz4 1.9.3 lz4hc.c; level 9 with its pattern analysis), 10..12 =
 * the LZ4 HC optimal parser (96 / 512 / 16 384 searches): byte-identical to the reference at that
 * level, and slower with every level (a chain link is a dependent memory access).
 * HC keeps a 320 KiB table set per wave in internal scratch (at most GPUMT_LZ4HC_WAVES of them).
 */
#define GPUMT_LZ4HC_SCRATCH 327936u
#define GPUMT_LZ4HC_WAVES 4096u
int gpumt_lz4_level_supported(int level);
int gpumt_lz4_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk,
				   void *d_slots, size_t slot_stride, uint32_t *d_rec_len, int level,
				   int stream);

/*
 * Ordered concatenation: d_rec_off[i] = sum of d_rec_len[0..i), d_rec_off[nrec] = total, and the
 * records are packed into d_stream at those offsets (the MT stream, ready for fn_write / D2H).
 */
int gpumt_lz4_compact(gpumt_ctx *h, const void *d_slots, size_t slot_stride,
		      const uint32_t *d_rec_len, size_t nrec, void *d_stream,
		      uint64_t *d_rec_off, int stream);

/*
 * For records located at d_stream + d_rec_off[i] (length d_rec_len[i], including the 12-byte
 * skippable header) read each frame's content-size field and produce d_out_len[i] and the
 * exclusive scan d_out_off[0..nrec] (what pt_decompress does per record at :333-334).
 */
int gpumt_lz4_probe_sizes(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
			  const uint32_t *d_rec_len, size_t nrec, uint32_t *d_out_len,
			  uint64_t *d_out_off, int stream);

/*
 * Decode nrec records.  Record i's content goes to d_out + d_out_off[i] and must be exactly
 * d_out_len[i] bytes; header, block structure, content size and XXH32 content checksum are
 * verified.  d_status[i] receives a GPUMT_ST_* code.
 * stream_bytes / out_bytes: upper bounds of the bytes spanned by the records in d_stream and by
 * their contents in d_out (they size the internal token-list scratch, about 0.7 x stream_bytes).
 * The d_stream allocation must extend at least 256 readable bytes past stream_bytes: the parse
 * kernel fetches whole aligned 128-byte lines (their contents past the last record are ignored).
 */
int gpumt_lz4_decompress_batch(gpumt_ctx *h, const void *d_stream, size_t stream_bytes,
			       const uint64_t *d_rec_off, const uint32_t *d_rec_len, size_t nrec,
			       void *d_out, size_t out_bytes, const uint64_t *d_out_off,
			       uint32_t *d_out_len, uint32_t *d_status, int stream);

/* ---- zstd-mt records (12-byte skippable header + one zstd frame, lib/zstd-mt_compress.c:296-302) ----
 *
 * gpumt_zstd_probe_sizes: d_out_len[i] = Frame_Content_Size of record i, d_out_off = exclusive
 * scan, d_status[i] = GPUMT_ST_OK or why the record cannot be decoded on the device (bad record /
 * frame header, or GPUMT_ST_UNSUPPORTED for frames without a content size -- zstd-mt always writes
 * one, it compresses each chunk with one-shot ZSTD_compress, :285).
 *
 * gpumt_zstd_decompress_batch: decode the records whose d_status is GPUMT_ST_OK (replaces
 * ZSTD_decompressStream at lib/zstd-mt_decompress.c:464): raw / RLE / compressed blocks, Huffman
 * literals, FSE sequence tables of every mode, repeat offsets, XXH64 content checksum (verified, a
 * mismatch reports GPUMT_ST_BAD_CHECKSUM); no dictionaries.  Same d_stream slack rule as above.
 * d_out_len[i] is the exact content size for frames that state one (what the probe returns); for a
 * frame without a content size (streaming writers, plain .zst files) the caller passes a capacity
 * there and the decoder replaces it by the decoded size.
 * out_bytes is the size of the output the batch produces (every record's [d_out_off[i], d_out_off[i] + d_out_len[i]) lies
 * inside it).
 * Internal scratch: GPUMT_ZSTD_DEC_SCRATCH (320 KiB) per record of a launch slice (at most 16384
 * records, 5 GiB) + 8 B per record; batches whose records average more than 128 KiB of content (frames of several blocks)
 * take out_bytes + 16 more for the sequence pre-pass (zmt_zstd_seq_kernel: the FSE sequence streams of a frame's blocks
 * decoded side by side ahead of the frame decoder; GPUMT_ZSTD_SEQ=1 or gpumt_set_variant(h, "zstd_seq", 1) turn it off).
 */
/* Bytes one zstd record slot occupies: room for the record + frame header and one padded area per
 * 128 KiB block (the blocks are compressed independently and then moved together), rounded to 256. */
size_t gpumt_zstd_slot_stride(size_t chunk);

/*
 * Compress n bytes at d_in (the allocation must extend 64 readable bytes past n) as independent
 * chunks of `chunk` bytes (last one shorter; n == 0 gives one empty chunk): record i (12-byte
 * skippable header + one zstd frame that decodes to chunk i; replaces ZSTD_compress at
 * lib/zstd-mt_compress.c:285 + header emit :296-302) is written at d_slots + i*slot_stride and its
 * length to d_rec_len[i].  gpumt_lz4_compact then packs the records into the MT stream.
 * The frames are valid RFC 8878 (single segment, content size, no checksum, blocks of at most
 * 128 KiB); their bytes are not those of libzstd -- the bar for zstd is decompress-identical.
 */
int gpumt_zstd_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk,
			      void *d_slots, size_t slot_stride, uint32_t *d_rec_len, int stream);
/* The same at `level` 1..22 (what the reference hands to ZSTD_compress, lib/zstd-mt_compress.c:285): the device encoder
 * has three tiers -- levels 1-2, 3-9, 10-22 (gpumt_zstd_level_tier = 0, 1, 2) -- that trade speed for ratio through
 * the size of the hash table and the number of bytes hashed; gpumt_zstd_compress_batch is level 1. */
int gpumt_zstd_level_tier(int level);
int gpumt_zstd_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				    size_t slot_stride, uint32_t *d_rec_len, int level, int stream);

int gpumt_zstd_probe_sizes(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
			   const uint32_t *d_rec_len, size_t nrec, uint32_t *d_out_len,
			   uint64_t *d_out_off, uint32_t *d_status, int stream);
int gpumt_zstd_decompress_batch(gpumt_ctx *h, const void *d_stream, size_t stream_bytes,
				const uint64_t *d_rec_off, const uint32_t *d_rec_len, size_t nrec,
				void *d_out, size_t out_bytes, const uint64_t *d_out_off,
				uint32_t *d_out_len, uint32_t *d_status, int stream);

/* ---- brotli-mt records (16-byte header + one raw brotli stream, lib/brotli-mt_compress.c:285-304) ----
 *
 * gpumt_brotli_decompress_batch: decode nrec brotli streams (replaces BrotliDecoderDecompress at
 * lib/brotli-mt_decompress.c:344-346).  Stream i is d_stream + d_rec_off[i], d_rec_len[i] bytes (the
 * payload behind the 16-byte header, which the host parses while reading, :187-284); its output goes
 * to d_out + d_out_off[i] and may take d_out_cap[i] bytes (hint << 16, :236-239).  d_out_len[i]
 * receives the decoded size, d_status[i] GPUMT_ST_OK, GPUMT_ST_BAD_BLOCK (malformed / truncated
 * stream) or GPUMT_ST_SIZE_MISMATCH (output exceeds the capacity) -- the reference reports
 * frame_decompress for both.  Complete RFC 7932 decoder incl. the static dictionary.  Same
 * d_stream slack rule as above.  Internal scratch: GPUMT_BROTLI_SCRATCH bytes per resident wave.
 */
/* gpumt_brotli_compress_batch: chunk i = d_in[i*chunk, ...) -> one record (16-byte brotli-mt header with
 * the output hint, lib/brotli-mt_compress.c:285-304, + one raw brotli stream that decodes to the
 * chunk; replaces BrotliEncoderCompress at :269-272) at d_slots + i*slot_stride, its length to
 * d_rec_len[i]; slot_stride >= gpumt_zstd_slot_stride(chunk).  gpumt_lz4_compact packs the records.
 * The streams are valid RFC 7932 (window 2^18, one meta-block per 128 KiB with its own three prefix
 * codes, byte-aligned by empty metadata meta-blocks); their bytes are not libbrotli's -- the bar for
 * brotli is decompress-identical. */
int gpumt_brotli_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				size_t slot_stride, uint32_t *d_rec_len, int stream);
/* The same at quality `level` 0..11 (what the reference hands to BrotliEncoderCompress, lib/brotli-mt_compress.c:269-272):
 * three device tiers -- 0-3, 4-8, 9-11 (gpumt_brotli_level_tier = 0, 1, 2); gpumt_brotli_compress_batch is quality 1. */
int gpumt_brotli_level_tier(int level);
int gpumt_brotli_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				      size_t slot_stride, uint32_t *d_rec_len, int level, int stream);

#define GPUMT_BROTLI_SCRATCH 825856u
/* zstd decode: scratch per record (literals of one 128 KiB unit + 24576 sequences decoded ahead) */
#define GPUMT_ZSTD_DEC_SCRATCH 327936u
int gpumt_brotli_decompress_batch(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
				  const uint32_t *d_rec_len, size_t nrec, void *d_out,
				  const uint64_t *d_out_off, const uint32_t *d_out_cap,
				  uint32_t *d_out_len, uint32_t *d_status, int stream);

/* ---- snappy-mt records (16-byte header + one raw snappy stream, lib/snappy-mt_compress.c:280-300) ----
 *
 * gpumt_snappy_compress_batch: chunk i = d_in[i*chunk, ...) -> one record (header with the payload size,
 * "SP" and the reference's hint, + one raw snappy stream that decodes to the chunk; replaces
 * snappy_compress at lib/snappy-mt_compress.c:264-276) at d_slots + i*slot_stride, its length to
 * d_rec_len[i]; slot_stride >= gpumt_snappy_slot_stride(chunk) (16 + snappy_max_compressed_length).
 * gpumt_lz4_compact packs the records.  The streams are valid raw snappy (copies stay inside 64 KiB
 * blocks); their bytes are not those of the reference's snappy library, which is not part of its tree
 * -- the bar is decompress-identical.  d_in must extend 64 readable bytes past n.
 *
 * gpumt_snappy_decompress_batch: decode nrec raw snappy streams (replaces snappy_uncompress at
 * lib/snappy-mt_decompress.c:350).  Stream i is d_stream + d_rec_off[i], d_rec_len[i] bytes (the
 * payload behind the 16-byte header); its output goes to d_out + d_out_off[i] and may take
 * d_out_cap[i] bytes -- the caller reads the stream's varint preamble for it, as the reference does
 * (snappy_uncompressed_length, :262-267).  d_out_len[i] receives the decoded size, d_status[i]
 * GPUMT_ST_OK, GPUMT_ST_BAD_BLOCK (malformed / truncated stream, or not the size its preamble states)
 * or GPUMT_ST_SIZE_MISMATCH (the preamble exceeds the capacity).  Same d_stream slack rule as above. */
size_t gpumt_snappy_slot_stride(size_t chunk);
int gpumt_snappy_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				size_t slot_stride, uint32_t *d_rec_len, int stream);
int gpumt_snappy_decompress_batch(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
				  const uint32_t *d_rec_len, size_t nrec, void *d_out,
				  const uint64_t *d_out_off, const uint32_t *d_out_cap,
				  uint32_t *d_out_len, uint32_t *d_status, int stream);

/* XXH32 (seed 0) of n items: item i = d_base + d_off[i], d_len[i] bytes -> d_hash[i]. */
int gpumt_xxh32_batch(gpumt_ctx *h, const void *d_base, const uint64_t *d_off,
		      const uint32_t *d_len, size_t n, uint32_t *d_hash, int stream);

/* Developer aid: read-and-clear the 16 phase-cycle counters filled by the profiling decoder
 * (gpumt_set_variant("lz4_dec", 2)). */
int gpumt_debug_counters(gpumt_ctx *h, unsigned long long *dst, int n);

/* Kernel-variant selector for A/B measurements (0 = default). Returns previous value. */
int gpumt_set_variant(gpumt_ctx *h, const char *what, int variant);

#ifdef __cplusplus
}
#endif
#endif
