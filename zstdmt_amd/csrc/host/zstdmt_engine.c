/*
 * zstdmt_engine.c -- the host side of zstd-mt on MI355X: ZSTDCB_* (include/zstd-mt.h) over gpumt_*.
 *
 * Same pipeline as lz4mt_engine.c (batches of records through H2D / kernels / D2H on three
 * streams, four slots: mt_pipe.h), following the callback-visible behaviour of the reference
 * (lib/zstd-mt_compress.c:208-392, lib/zstd-mt_decompress.c:209-549,693-843):
 *   compress   one fn_read of exactly `inputsize` per chunk, one fn_write per record, in order;
 *              counters are reset per call (:337-341); empty input still yields one frame (:264);
 *   decompress a 16-byte sniff, then csize-4 bytes for the first record, then 12 + csize bytes per
 *              record (:221-353); one fn_write per frame in order.
 * Stream layouts accepted by decompress: the one the compressor writes ("pzstd style": skippable
 * frame first, :251-284) and the old "zstdmt style" (a 9-byte empty zstd frame in front, :225-249).
 * Plain .zst streams (the reference's single-threaded path) are split into frames on the
 * host and decoded by the same kernels (plain_decompress below).
 * Plain C, no HIP header.
 */
#include "mt_host.h"
#include "mt_pipe.h"
#include "zstd-mt.h"

size_t zstdmt_errcode;

/* ------------------------------------------------------------------ errors (zstd-mt_common.c) */
unsigned ZSTDCB_isError(size_t code)
{
	return code > ZSTDCB_ERROR(maxCode);
}

const char *ZSTDCB_getErrorString(size_t code)
{
	/* strings of lib/zstd-mt_common.c:40-61 (init_missing and canceled have none there) */
	static const char *const codec[] = {
		"", "device: malformed record header", "device: bad zstd frame header",
		"device: malformed zstd block", "device: content size mismatch",
		"device: content checksum mismatch", "device: trailing bytes after frame",
		"device: unsupported zstd frame feature",
	};
	const size_t idx = (size_t)0 - code;
	if (zstdmt_errcode >= 1 && zstdmt_errcode <= 7 && idx == ZSTDCB_error_compression_library)
		return codec[zstdmt_errcode];
	switch ((ZSTDCB_ErrorCode)idx) {
	case ZSTDCB_error_no_error:
		return "No error detected";
	case ZSTDCB_error_memory_allocation:
		return "Allocation error : not enough memory";
	case ZSTDCB_error_read_fail:
		return "Read failure";
	case ZSTDCB_error_write_fail:
		return "Write failure";
	case ZSTDCB_error_data_error:
		return "Malformed input";
	case ZSTDCB_error_frame_compress:
		return "Could not compress frame at once";
	case ZSTDCB_error_frame_decompress:
		return "Could not decompress frame at once";
	case ZSTDCB_error_compressionParameter_unsupported:
		return "Compression parameter is out of bound";
	case ZSTDCB_error_compression_library:
		return "Compression library reports failure";
	default:
		return "Unspecified zstmt error code"; /* sic, zstd-mt_common.c:34 */
	}
}

/* callback return value -> library error (mt_error, zstd-mt_compress.c:160-173) */
static size_t mt_error(int rv)
{
	switch (rv) {
	case -1:
		return ZSTDCB_ERROR(read_fail);
	case -2:
		return ZSTDCB_ERROR(canceled);
	case -3:
		return ZSTDCB_ERROR(memory_allocation);
	}
	return ZSTDCB_ERROR(read_fail);
}

static int is_zstd_magic(const uint8_t *p) /* IsZstd_Magic, zstd-mt_decompress.c:146-153 */
{
	const uint32_t m = rd32(p);
	return m == ZSTDCB_MAGICNUMBER_V01 || (m >= ZSTDCB_MAGICNUMBER_MIN && m <= ZSTDCB_MAGICNUMBER_MAX);
}

/* Frame_Content_Size of a zstd frame header (RFC 8878 3.1.1.1), ~0 when absent / malformed */
static uint64_t zstd_content_size(const uint8_t *f, size_t n)
{
	if (n < 6 || rd32(f) != ZSTDCB_MAGICNUMBER_MAX)
		return ~(uint64_t)0;
	{
		const unsigned fhd = f[4], fcs = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
		const unsigned did_len = did == 3 ? 4 : did, fcs_len = fcs == 0 ? single : 1u << fcs;
		const size_t hp = 5 + (1 - single) + did_len;
		uint64_t c = 0;
		if (fcs_len == 0 || n < hp + fcs_len)
			return ~(uint64_t)0;
		for (unsigned k = 0; k < fcs_len; k++)
			c |= (uint64_t)f[hp + k] << (8 * k);
		return fcs == 1 ? c + 256 : c;
	}
}

/* =================================================================== compression */
struct cslot {
	dbuf in;      /* chunk data, H2D                       */
	dbuf slots;   /* device only: per-chunk records        */
	dbuf stream;  /* packed records, D2H                   */
	dbuf meta;    /* rec_len[n] u32 | pad | rec_off[n+1] u64, D2H */
	size_t n;     /* bytes in the batch                    */
	size_t nrec;
};

struct ZSTDCB_CCtx_s {
	int level, threads, inputsize;
	size_t insize, outsize, curframe, frames; /* insize / frames: reader; outsize / curframe: writer */
	mt_gpus gpus; /* the devices the batch slots are dealt out to (mt_host.h) */
	struct cslot s[MT_NSLOT];
	ZSTDCB_RdWr_t *io;
	size_t maxrec;
};

ZSTDCB_CCtx *ZSTDCB_createCCtx(int threads, int level, int inputsize)
{
	/* default chunk = 1 << (windowLog[level] + 1), indexed by the level itself as in
	 * zstd-mt_compress.c:116-127 (level 22 reads past that table there; 1 GiB here) */
	static const int window_log[] = {19, 19, 20, 20, 20, 21, 21, 21, 21, 21, 22, 22,
					 22, 22, 22, 23, 23, 23, 23, 25, 26, 27, 29};
	ZSTDCB_CCtx *ctx;
	if (threads < 1 || threads > ZSTDCB_THREAD_MAX)
		return NULL;
	if (level < ZSTDCB_LEVEL_MIN || level > ZSTDCB_LEVEL_MAX)
		return NULL;
	if (inputsize < 0)
		return NULL;
	ctx = (ZSTDCB_CCtx *)calloc(1, sizeof *ctx);
	if (!ctx)
		return NULL;
	ctx->level = level;
	ctx->threads = threads;
	ctx->inputsize = inputsize ? inputsize : 1 << (window_log[level] + 1);
	if (mt_gpus_open(&ctx->gpus)) {
		free(ctx); /* no device: fail loudly, there is no CPU path */
		return NULL;
	}
	return ctx;
}

void ZSTDCB_freeCCtx(ZSTDCB_CCtx *ctx)
{
	if (!ctx)
		return;
	for (int i = 0; i < MT_NSLOT; i++) {
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].in);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].slots);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].stream);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].meta);
	}
	mt_gpus_close(&ctx->gpus);
	free(ctx);
}

/* NULL context: init_missing for the compression getters (zstd-mt_compress.c:395-423) */
size_t ZSTDCB_GetFramesCCtx(ZSTDCB_CCtx *ctx) { return ctx ? ctx->curframe : ZSTDCB_ERROR(init_missing); }
size_t ZSTDCB_GetInsizeCCtx(ZSTDCB_CCtx *ctx) { return ctx ? ctx->insize : ZSTDCB_ERROR(init_missing); }
size_t ZSTDCB_GetOutsizeCCtx(ZSTDCB_CCtx *ctx) { return ctx ? ctx->outsize : ZSTDCB_ERROR(init_missing); }

/* one fn_read of exactly `inputsize` per chunk; EOF = zero-length read once a frame exists
 * (pt_compress, zstd-mt_compress.c:250-277) */
static size_t c_read_batch(ZSTDCB_CCtx *ctx, ZSTDCB_RdWr_t *io, struct cslot *s, size_t maxrec, int *eof)
{
	const size_t chunk = (size_t)ctx->inputsize;
	s->n = 0;
	s->nrec = 0;
	while (s->nrec < maxrec) {
		ZSTDCB_Buffer b;
		int rv;
		b.buf = (uint8_t *)s->in.h + s->n;
		b.size = chunk;
		b.allocated = chunk;
		rv = io->fn_read(io->arg_read, &b);
		if (rv != 0)
			return mt_error(rv);
		if (b.size == 0 && ctx->frames > 0) {
			*eof = 1;
			break;
		}
		if (b.size > chunk)
			return ZSTDCB_ERROR(read_fail);
		ctx->insize += b.size;
		ctx->frames++;
		s->n += b.size;
		s->nrec++;
		if (b.size < chunk)
			break; /* ragged chunk: last one of this device batch */
	}
	return 0;
}

static size_t c_launch(ZSTDCB_CCtx *ctx, struct cslot *s)
{
	gpumt_ctx *g = mt_gpu_of(&ctx->gpus, (int)(s - ctx->s));
	const int ks = mt_stream_of(&ctx->gpus, (int)(s - ctx->s)); /* the slot's own kernel stream: batches overlap on the device */
	const size_t chunk = (size_t)ctx->inputsize;
	const size_t stride = gpumt_zstd_slot_stride(chunk);
	uint32_t *d_len = (uint32_t *)s->meta.d;
	uint64_t *d_off = (uint64_t *)((uint8_t *)s->meta.d + ((s->nrec * 4 + 15) & ~(size_t)15));
	int rc = 0;
	if (s->n)
		rc |= gpumt_memcpy_h2d(g, s->in.d, s->in.h, s->n, 1);
	rc |= gpumt_stream_wait(g, ks, 1);
	/* the level the caller asked for reaches the encoder as the reference hands it to ZSTD_compress
	 * (/root/reference/lib/zstd-mt_compress.c:285): three device tiers, gpumt_zstd_level_tier */
	rc |= gpumt_zstd_compress_batch_level(g, s->in.d, s->n, chunk, s->slots.d, stride, d_len, ctx->level, ks);
	rc |= gpumt_lz4_compact(g, s->slots.d, stride, d_len, s->nrec, s->stream.d, d_off, ks);
	/* sizes, offsets and the packed records go to the pinned mirrors from the slot's own stream, the
	 * byte count of the records read on the device (d_off[nrec]): no host round trip in between, and
	 * the batches of the pipeline overlap (gpumt_push_host) */
	rc |= gpumt_push_host(g, s->meta.h, s->meta.d, ((s->nrec * 4 + 15) & ~(size_t)15) + (s->nrec + 1) * 8, NULL, ks);
	rc |= gpumt_push_host(g, s->stream.h, s->stream.d, s->stream.cap & ~(size_t)15, d_off + s->nrec, ks);
	return rc ? ZSTDCB_ERROR(compression_library) : 0;
}

/* ---- the three roles (mt_pipe.h) ---- */
static void cp_role_start(void *a) { mt_bind_near(&((ZSTDCB_CCtx *)a)->gpus); }
static size_t cp_fill(void *a, int si, int *has_data, int *eof)
{
	ZSTDCB_CCtx *ctx = (ZSTDCB_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const size_t chunk = (size_t)ctx->inputsize, stride = gpumt_zstd_slot_stride(chunk);
	size_t lim = zmt_batch_bytes_for(chunk) / chunk, err;
	if (lim < 1)
		lim = 1;
	if (lim > BATCH_MAXREC)
		lim = BATCH_MAXREC;
	if (ctx->maxrec > lim)
		ctx->maxrec = lim;
	if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, ctx->maxrec * chunk + 512, 1, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->slots, ctx->maxrec * stride, 0, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->stream, ctx->maxrec * stride + 512, 1, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->meta, ctx->maxrec * 12 + 64, 1, 1))
		return ZSTDCB_ERROR(memory_allocation);
	err = c_read_batch(ctx, ctx->io, s, ctx->maxrec, eof);
	*has_data = s->nrec > 0;
	ctx->maxrec *= 4;
	return err;
}

static size_t cp_launch(void *a, int si)
{
	ZSTDCB_CCtx *ctx = (ZSTDCB_CCtx *)a;
	size_t err = c_launch(ctx, &ctx->s[si]);
	if (!err && gpumt_mark(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si), mt_stream_of(&ctx->gpus, si)))
		err = ZSTDCB_ERROR(compression_library);
	return err;
}

static size_t cp_complete(void *a, int si)
{
	ZSTDCB_CCtx *ctx = (ZSTDCB_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const uint64_t *off = (const uint64_t *)((const uint8_t *)s->meta.h + ((s->nrec * 4 + 15) & ~(size_t)15));
	size_t total;
	if (gpumt_mark_sync(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si)))
		return ZSTDCB_ERROR(compression_library);
	total = (size_t)off[s->nrec];
	if (total > s->stream.cap)
		return ZSTDCB_ERROR(frame_compress);
	return 0;
}

static size_t cp_drain(void *a, int si)
{
	ZSTDCB_CCtx *ctx = (ZSTDCB_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const uint32_t *len = (const uint32_t *)s->meta.h;
	const uint64_t *off = (const uint64_t *)((const uint8_t *)s->meta.h + ((s->nrec * 4 + 15) & ~(size_t)15));
	for (size_t i = 0; i < s->nrec; i++) { /* pt_write: strictly in frame order */
		ZSTDCB_Buffer b;
		int rv;
		b.buf = (uint8_t *)s->stream.h + off[i];
		b.size = len[i];
		b.allocated = len[i];
		rv = ctx->io->fn_write(ctx->io->arg_write, &b);
		if (rv != 0)
			return mt_error(rv);
		ctx->outsize += len[i];
		ctx->curframe++;
	}
	return 0;
}

size_t ZSTDCB_compressCCtx(ZSTDCB_CCtx *ctx, ZSTDCB_RdWr_t *rdwr)
{
	static const mt_pipe_ops ops = {cp_fill, cp_launch, cp_complete, cp_drain, cp_role_start};
	size_t err;

	if (!ctx)
		return ZSTDCB_ERROR(init_missing); /* zstd-mt_compress.c:327-328 */
	/* counters restart with every call (zstd-mt_compress.c:337-341) */
	ctx->insize = ctx->outsize = ctx->frames = ctx->curframe = 0;
	zstdmt_errcode = 0;
	ctx->io = rdwr;
	ctx->maxrec = BATCH_MIN / (size_t)ctx->inputsize;
	if (ctx->maxrec < 1)
		ctx->maxrec = 1;
	err = mt_pipe_run_n(&ops, ctx, mt_nslot_for(ctx->gpus.n));
	mt_gpus_sync(&ctx->gpus);
	return err;
}

/* =================================================================== decompression */
struct dslot {
	dbuf in;     /* record bytes (headers included), H2D                                  */
	dbuf meta;   /* rec_off u64[n] | out_off u64[n+1] | rec_len u32[n] | out_len u32[n], H2D */
	dbuf status; /* u32[n], D2H                                                           */
	dbuf out;    /* decoded chunks, D2H                                                   */
	size_t nrec, in_bytes, out_bytes;
	int unsized; /* a frame of the batch states no content size: out_len holds capacities, read back */
};

struct ZSTDCB_DCtx_s {
	int threads, inputsize;
	size_t budget; /* output bytes per device batch, grows from BATCH_MIN to zmt_batch_bytes_for(the largest record seen) */
	size_t big_out;
	size_t insize, outsize, curframe, frames;
	mt_gpus gpus; /* the devices the batch slots are dealt out to (mt_host.h) */
	struct dslot s[MT_NSLOT];
	ZSTDCB_RdWr_t *io;
	int have_hdr; /* a record header read ahead of its batch */
	uint32_t hdr_csize;
	uint8_t first4[4]; /* first record: the 4 frame bytes that came with the sniff */
	int have_first4;
};

ZSTDCB_DCtx *ZSTDCB_createDCtx(int threads, int inputsize)
{
	ZSTDCB_DCtx *ctx;
	if (threads < 1 || threads > ZSTDCB_THREAD_MAX)
		return NULL;
	ctx = (ZSTDCB_DCtx *)calloc(1, sizeof *ctx);
	if (!ctx)
		return NULL;
	ctx->threads = threads;
	ctx->inputsize = inputsize ? inputsize : 1024 * 512; /* zstd-mt_decompress.c:125-128 */
	if (mt_gpus_open(&ctx->gpus)) {
		free(ctx);
		return NULL;
	}
	return ctx;
}

void ZSTDCB_freeDCtx(ZSTDCB_DCtx *ctx)
{
	if (!ctx)
		return;
	for (int i = 0; i < MT_NSLOT; i++) {
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].in);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].meta);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].status);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].out);
	}
	mt_gpus_close(&ctx->gpus);
	free(ctx);
}

size_t ZSTDCB_GetFramesDCtx(ZSTDCB_DCtx *ctx) { return ctx ? ctx->curframe : 0; }
size_t ZSTDCB_GetInsizeDCtx(ZSTDCB_DCtx *ctx) { return ctx ? ctx->insize : 0; }
size_t ZSTDCB_GetOutsizeDCtx(ZSTDCB_DCtx *ctx) { return ctx ? ctx->outsize : 0; }

#define D_META_BYTES(n) ((n) * 8 + ((n) + 1) * 8 + (n) * 4 + (n) * 4 + 64)
static uint64_t *m_rec_off(struct dslot *s, int dev) { return (uint64_t *)(dev ? s->meta.d : s->meta.h); }
static uint64_t *m_out_off(struct dslot *s, int dev) { return m_rec_off(s, dev) + BATCH_MAXREC; }
static uint32_t *m_rec_len(struct dslot *s, int dev) { return (uint32_t *)(m_out_off(s, dev) + BATCH_MAXREC + 1); }
static uint32_t *m_out_len(struct dslot *s, int dev) { return m_rec_len(s, dev) + BATCH_MAXREC; }

/* next 12-byte record header (pt_read, zstd-mt_decompress.c:299-327) */
static size_t d_read_header(ZSTDCB_DCtx *ctx, ZSTDCB_RdWr_t *io, uint32_t *csize, int *eof)
{
	uint8_t hb[12];
	ZSTDCB_Buffer b;
	int rv;
	b.buf = hb;
	b.size = 12;
	b.allocated = 12;
	rv = io->fn_read(io->arg_read, &b);
	if (rv != 0)
		return mt_error(rv);
	if (b.size == 0) {
		*eof = 1;
		return 0;
	}
	if (b.size != 12)
		return ZSTDCB_ERROR(read_fail);
	if (rd32(hb) != ZSTDCB_MAGIC_SKIPPABLE)
		return ZSTDCB_ERROR(data_error);
	ctx->insize += 12;
	*csize = rd32(hb + 8);
	return 0;
}

/* 0 = not complete yet, EXTENT_INVALID = cannot become a frame (reserved bit, reserved block type, block larger
 * than the format allows): see lz4mt_engine.c */
#define EXTENT_INVALID ((size_t)-1)
static size_t zstd_frame_extent(const uint8_t *p, size_t n, uint64_t *bound, int *sized);

static size_t d_read_batch(ZSTDCB_DCtx *ctx, ZSTDCB_RdWr_t *io, struct dslot *s, int *eof)
{
	s->nrec = 0;
	s->unsized = 0;
	s->in_bytes = 0;
	s->out_bytes = 0;
	while (s->nrec < BATCH_MAXREC) {
		uint32_t csize = 0;
		uint64_t osz;
		uint8_t *rec;
		ZSTDCB_Buffer b;
		size_t err, skip = 0;
		int rv;
		if (ctx->have_hdr) {
			csize = ctx->hdr_csize;
		} else {
			err = d_read_header(ctx, io, &csize, eof);
			if (err)
				return err;
			if (*eof)
				break;
		}
		if (s->nrec && (s->in_bytes + 12 + (size_t)csize > s->in.cap - 512 || s->out_bytes >= ctx->budget)) {
			ctx->have_hdr = 1;
			ctx->hdr_csize = csize;
			break;
		}
		ctx->have_hdr = 0;
		if (s->in_bytes + 12 + (size_t)csize + 512 > s->in.cap) {
			dbuf old = s->in;
			memset(&s->in, 0, sizeof s->in);
			if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, s->in_bytes + 12 + (size_t)csize + 512, 1, 1)) {
				dbuf_free(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in);
				s->in = old; /* keep the slot as it was: freeCtx releases it */
				return ZSTDCB_ERROR(memory_allocation);
			}
			memcpy(s->in.h, old.h, s->in_bytes);
			dbuf_free(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &old);
		}
		rec = (uint8_t *)s->in.h + s->in_bytes;
		rec[0] = 0x50; rec[1] = 0x2A; rec[2] = 0x4D; rec[3] = 0x18;
		rec[4] = 4; rec[5] = rec[6] = rec[7] = 0;
		rec[8] = (uint8_t)csize; rec[9] = (uint8_t)(csize >> 8);
		rec[10] = (uint8_t)(csize >> 16); rec[11] = (uint8_t)(csize >> 24);
		if (ctx->have_first4) {
			/* first record: 4 payload bytes arrived with the 16-byte sniff (:262-270) */
			if (csize < 4)
				return ZSTDCB_ERROR(data_error);
			memcpy(rec + 12, ctx->first4, 4);
			skip = 4;
			ctx->have_first4 = 0;
		}
		b.buf = rec + 12 + skip;
		b.size = csize - skip;
		b.allocated = b.size;
		rv = io->fn_read(io->arg_read, &b);
		if (rv != 0)
			return mt_error(rv);
		if (b.size != csize - skip)
			return ZSTDCB_ERROR(data_error);
		ctx->insize += b.size;
		ctx->frames++;
		osz = zstd_content_size(rec + 12, csize);
		if (osz == ~(uint64_t)0) {
			/* No content size: never written by zstd-mt, but pzstd-style writers that stream
			 * their frames do it, and the reference just grows its buffer (:499-522).  The block
			 * headers bound the content; the decoder replaces the capacity by the size. */
			uint64_t bound = 0;
			int sized = 0;
			const size_t fl = zstd_frame_extent(rec + 12, csize, &bound, &sized);
			if (!fl || fl == EXTENT_INVALID || sized) {
				zstdmt_errcode = GPUMT_ST_BAD_FRAME;
				return ZSTDCB_ERROR(compression_library);
			}
			osz = bound;
			s->unsized = 1;
		}
		if (osz > 0x7FFFFFFFull) {
			zstdmt_errcode = GPUMT_ST_UNSUPPORTED;
			return ZSTDCB_ERROR(compression_library);
		}
		if ((size_t)osz > ctx->big_out)
			ctx->big_out = (size_t)osz;
		m_rec_off(s, 0)[s->nrec] = s->in_bytes;
		m_rec_len(s, 0)[s->nrec] = 12 + csize;
		m_out_off(s, 0)[s->nrec] = s->out_bytes;
		m_out_len(s, 0)[s->nrec] = (uint32_t)osz;
		s->in_bytes += 12 + (size_t)csize;
		s->out_bytes += (size_t)osz;
		s->nrec++;
	}
	m_out_off(s, 0)[s->nrec] = s->out_bytes;
	return 0;
}

static size_t d_launch(ZSTDCB_DCtx *ctx, struct dslot *s)
{
	gpumt_ctx *g = mt_gpu_of(&ctx->gpus, (int)(s - ctx->s));
	/* each batch slot launches on its own kernel stream (4 + slot): the decoders are bound by the
	 * latency of a record, so the batches of the pipeline must overlap on the device */
	const int ks = mt_stream_of(&ctx->gpus, (int)(s - ctx->s));
	int rc = 0;
	if (dbuf_want(g, &s->out, s->out_bytes + 64, 1, 1) || dbuf_want(g, &s->status, s->nrec * 4 + 64, 1, 1))
		return ZSTDCB_ERROR(memory_allocation);
	memset(s->status.h, 0, s->nrec * 4); /* GPUMT_ST_OK: the decode kernel only visits those */
	rc |= gpumt_memcpy_h2d(g, s->in.d, s->in.h, s->in_bytes, 1);
	rc |= gpumt_memcpy_h2d(g, s->meta.d, s->meta.h, D_META_BYTES(BATCH_MAXREC), 1);
	rc |= gpumt_memcpy_h2d(g, s->status.d, s->status.h, s->nrec * 4, 1);
	rc |= gpumt_stream_wait(g, ks, 1);
	rc |= gpumt_zstd_decompress_batch(g, s->in.d, s->in_bytes, m_rec_off(s, 1), m_rec_len(s, 1), s->nrec,
					  s->out.d, s->out_bytes, m_out_off(s, 1), m_out_len(s, 1),
					  (uint32_t *)s->status.d, ks);
	rc |= gpumt_stream_wait(g, 2, ks);
	rc |= gpumt_memcpy_d2h(g, s->status.h, s->status.d, s->nrec * 4, 2);
	if (s->unsized) /* capacities -> decoded sizes */
		rc |= gpumt_memcpy_d2h(g, m_out_len(s, 0), m_out_len(s, 1), s->nrec * 4, 2);
	if (s->out_bytes)
		rc |= gpumt_memcpy_d2h(g, s->out.h, s->out.d, s->out_bytes, 2);
	return rc ? ZSTDCB_ERROR(compression_library) : 0;
}

static void dp_role_start(void *a) { mt_bind_near(&((ZSTDCB_DCtx *)a)->gpus); }
static size_t dp_fill(void *a, int si, int *has_data, int *eof)
{
	ZSTDCB_DCtx *ctx = (ZSTDCB_DCtx *)a;
	struct dslot *s = &ctx->s[si];
	size_t err;
	if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, ctx->budget + (ctx->budget >> 3) + 4096, 1, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->meta, D_META_BYTES(BATCH_MAXREC), 1, 1))
		return ZSTDCB_ERROR(memory_allocation);
	err = d_read_batch(ctx, ctx->io, s, eof);
	*has_data = s->nrec > 0;
	if (ctx->budget < zmt_batch_bytes_for(ctx->big_out))
		ctx->budget *= 4;
	return err;
}

static size_t dp_launch(void *a, int si)
{
	ZSTDCB_DCtx *ctx = (ZSTDCB_DCtx *)a;
	size_t err = d_launch(ctx, &ctx->s[si]);
	if (!err && gpumt_mark(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si), 2))
		err = ZSTDCB_ERROR(compression_library);
	return err;
}

static size_t dp_complete(void *a, int si)
{
	ZSTDCB_DCtx *ctx = (ZSTDCB_DCtx *)a;
	return gpumt_mark_sync(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si)) ? ZSTDCB_ERROR(compression_library) : 0;
}

static size_t dp_drain(void *a, int si)
{
	ZSTDCB_DCtx *ctx = (ZSTDCB_DCtx *)a;
	struct dslot *s = &ctx->s[si];
	const uint32_t *st = (const uint32_t *)s->status.h;
	for (size_t i = 0; i < s->nrec; i++) {
		ZSTDCB_Buffer b;
		int rv;
		if (st[i] != GPUMT_ST_OK) {
			/* pt_decompress: any ZSTD error -> compression_library, code in the global (:534-536) */
			zstdmt_errcode = st[i];
			return ZSTDCB_ERROR(compression_library);
		}
		b.buf = (uint8_t *)s->out.h + m_out_off(s, 0)[i];
		b.size = m_out_len(s, 0)[i];
		b.allocated = b.size;
		rv = ctx->io->fn_write(ctx->io->arg_write, &b);
		if (rv != 0)
			return mt_error(rv);
		ctx->outsize += b.size;
		ctx->curframe++;
	}
	return 0;
}

/* =================================================================== plain .zst streams
 * The reference hands anything that starts with a zstd frame (and is not the old zstdmt layout) to
 * its single-threaded ZSTD_decompressStream loop (st_decompress, zstd-mt_decompress.c:552-687):
 * frames of the zstd CLI / library, any number of them, possibly without a content size and with
 * skippable frames in between.  Here the input is read about one batch ahead (same request sizes as the
 * reference: ZSTD_DStreamInSize() = 128 KiB + 3), split into frames on the host by walking the
 * block headers (RFC 8878 3.1.1.2), and decoded by the same device kernels, one wave per frame;
 * frames that do not state their content size get the sum of their block bounds as capacity and the
 * decoder reports the size.  Output leaves in pieces of at most ZSTD_DStreamOutSize() = 128 KiB like
 * the reference's; GetFrames stays 0 as in the reference (st_decompress counts no frames). */
#define ZSTD_IN_CHUNK 131075u
#define ZSTD_OUT_CHUNK 131072u

/* one frame at p[0..n): total length and output bound; 0 = malformed / truncated */
static size_t zstd_frame_extent(const uint8_t *p, size_t n, uint64_t *bound, int *sized)
{
	if (n < 6)
		return 0;
	const unsigned fhd = p[4], fcs = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3, chk = (fhd >> 2) & 1;
	const unsigned did_len = did == 3 ? 4 : did, fcs_len = fcs == 0 ? single : 1u << fcs;
	size_t hp = 5;
	uint64_t window = 0, content = 0, sum = 0;
	if (fhd & 8)
		return EXTENT_INVALID; /* reserved bit */
	if (n < 5 + (1 - single) + did_len + fcs_len)
		return 0;
	if (!single) {
		const unsigned wd = p[hp++];
		const uint64_t base = 1ull << (10 + (wd >> 3));
		window = base + (base >> 3) * (wd & 7);
	}
	hp += did_len;
	for (unsigned k = 0; k < fcs_len; k++)
		content |= (uint64_t)p[hp + k] << (8 * k);
	if (fcs == 1)
		content += 256;
	hp += fcs_len;
	if (single)
		window = content;
	const uint64_t block_max = window < 131072 ? window : 131072;
	for (;;) {
		if (n - hp < 3)
			return 0;
		const uint32_t bh = (uint32_t)p[hp] | (uint32_t)p[hp + 1] << 8 | (uint32_t)p[hp + 2] << 16;
		const uint32_t last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
		hp += 3;
		if (type == 3 || (type != 1 && bsize > 131072u))
			return EXTENT_INVALID; /* reserved block type; Block_Maximum_Size is 128 KiB at most (RFC 8878 3.1.1.2.4) */
		if (type == 1) { /* RLE: one byte regenerates bsize */
			if (n - hp < 1)
				return 0;
			hp += 1;
			sum += bsize;
		} else {
			if (n - hp < bsize)
				return 0;
			hp += bsize;
			sum += type == 0 ? bsize : block_max;
		}
		if (last)
			break;
	}
	if (chk) {
		if (n - hp < 4)
			return 0;
		hp += 4;
	}
	*sized = fcs_len != 0;
	*bound = fcs_len ? content : sum;
	return hp;
}

static size_t plain_write(ZSTDCB_DCtx *ctx, ZSTDCB_RdWr_t *io, const uint8_t *p, size_t n)
{
	while (n) {
		ZSTDCB_Buffer b;
		const size_t k = n < ZSTD_OUT_CHUNK ? n : ZSTD_OUT_CHUNK;
		int rv;
		b.buf = (void *)p;
		b.size = k;
		b.allocated = k;
		rv = io->fn_write(io->arg_write, &b);
		if (rv != 0)
			return mt_error(rv);
		ctx->outsize += k;
		p += k;
		n -= k;
	}
	return 0;
}

static size_t plain_decompress(ZSTDCB_DCtx *ctx, ZSTDCB_RdWr_t *io, const uint8_t *first, size_t nfirst, int at_eof)
{
	uint8_t *raw = (uint8_t *)malloc(ZSTD_IN_CHUNK);
	size_t cap = ZSTD_IN_CHUNK, n = nfirst, err = 0, ip = 0;
	size_t want_ahead = BATCH_BYTES; /* input buffered before a round of frames is split off */
	int eof = at_eof, first_read = 1;
	struct dslot *s = &ctx->s[0];
	gpumt_ctx *g = mt_gpu_of(&ctx->gpus, (int)(s - ctx->s));
	if (!raw)
		return ZSTDCB_ERROR(memory_allocation);
	memcpy(raw, first, nfirst);
	ctx->insize += nfirst;
	/* The input is consumed incrementally: read until about one batch of input is buffered or the
	 * stream ends (first request fills the first buffer behind the sniffed bytes, :590-609), decode
	 * the complete frames of what is there, keep the incomplete tail, repeat -- the host holds about
	 * two batches of input plus the largest frame, not the whole stream. */
	for (;;) {
	int need_more = 0;
	while (!eof && n - ip < want_ahead) {
		ZSTDCB_Buffer b;
		int rv;
		const size_t want = first_read ? ZSTD_IN_CHUNK - nfirst : ZSTD_IN_CHUNK;
		if (ip && ip == n) {
			n = 0;
			ip = 0;
		}
		if (n + want > cap) {
			if (ip >= want) { /* drop what is decoded instead of growing */
				memmove(raw, raw + ip, n - ip);
				n -= ip;
				ip = 0;
			} else {
				uint8_t *nr;
				cap = cap * 2 + want;
				nr = (uint8_t *)realloc(raw, cap);
				if (!nr) {
					free(raw);
					return ZSTDCB_ERROR(memory_allocation);
				}
				raw = nr;
			}
		}
		b.buf = raw + n;
		b.size = want;
		b.allocated = want;
		rv = io->fn_read(io->arg_read, &b);
		if (rv != 0) {
			free(raw);
			return mt_error(rv);
		}
		first_read = 0;
		if (b.size == 0) {
			eof = 1;
			break;
		}
		n += b.size;
		ctx->insize += b.size;
	}
	/* ---- frames, in batches the device buffers can hold ---- */
	while (ip < n && !err) {
		size_t in_bytes = 0, out_bytes = 0, nrec = 0;
		if (dbuf_want(g, &s->meta, D_META_BYTES(BATCH_MAXREC), 1, 1)) {
			err = ZSTDCB_ERROR(memory_allocation);
			break;
		}
		/* pass 1: extents of the frames of this batch */
		size_t jp = ip;
		while (jp < n && nrec < BATCH_MAXREC) {
			uint64_t bound;
			int sized;
			size_t flen;
			if (n - jp >= 8 && (rd32(raw + jp) & 0xFFFFFFF0u) == ZSTDCB_MAGIC_SKIPPABLE) {
				const size_t sk = 8 + (size_t)rd32(raw + jp + 4);
				if (sk > n - jp) {
					if (!eof)
						need_more = 1; /* the rest of it has not been read yet */
					else
						err = ZSTDCB_ERROR(compression_library);
					break;
				}
				jp += sk;
				continue;
			}
			if (!eof && n - jp <= 0xFFFFFFF0u &&
			    (n - jp < 8 || (rd32(raw + jp) == ZSTDCB_MAGICNUMBER_MAX &&
					    !zstd_frame_extent(raw + jp, n - jp, &bound, &sized)))) {
				need_more = 1; /* an incomplete frame: wait for the rest (a damaged one is EXTENT_INVALID, below) */
				break;
			}
			if (n - jp < 4 || rd32(raw + jp) != ZSTDCB_MAGICNUMBER_MAX ||
			    !(flen = zstd_frame_extent(raw + jp, n - jp, &bound, &sized)) || flen == EXTENT_INVALID ||
			    flen > 0xFFFFFFF0u ||
			    bound > 0x7FFFFFFFull) {
				zstdmt_errcode = GPUMT_ST_BAD_FRAME;
				err = ZSTDCB_ERROR(compression_library);
				break;
			}
			if (nrec && (in_bytes + 12 + flen > BATCH_BYTES || out_bytes + bound > 4 * BATCH_BYTES))
				break;
			m_rec_off(s, 0)[nrec] = in_bytes;
			m_rec_len(s, 0)[nrec] = (uint32_t)(12 + flen);
			m_out_off(s, 0)[nrec] = out_bytes;
			m_out_len(s, 0)[nrec] = (uint32_t)bound;
			in_bytes += 12 + flen;
			out_bytes += (size_t)bound;
			nrec++;
			jp += flen;
		}
		if (err)
			break;
		if (!nrec) { /* only skippable frames were left */
			ip = jp;
			if (need_more)
				break;
			continue;
		}
		m_out_off(s, 0)[nrec] = out_bytes;
		if (dbuf_want(g, &s->in, in_bytes + 512, 1, 1) || dbuf_want(g, &s->out, out_bytes + 64, 1, 1) ||
		    dbuf_want(g, &s->status, nrec * 4 + 64, 1, 1)) {
			err = ZSTDCB_ERROR(memory_allocation);
			break;
		}
		/* pass 2: records = 12-byte skippable header + frame, the layout the kernels take */
		{
			size_t k = 0, q = ip;
			while (k < nrec) {
				if ((rd32(raw + q) & 0xFFFFFFF0u) == ZSTDCB_MAGIC_SKIPPABLE) {
					q += 8 + (size_t)rd32(raw + q + 4);
					continue;
				}
				const uint32_t flen = m_rec_len(s, 0)[k] - 12;
				uint8_t *rec = (uint8_t *)s->in.h + m_rec_off(s, 0)[k];
				rec[0] = 0x50; rec[1] = 0x2A; rec[2] = 0x4D; rec[3] = 0x18;
				rec[4] = 4; rec[5] = rec[6] = rec[7] = 0;
				rec[8] = (uint8_t)flen; rec[9] = (uint8_t)(flen >> 8);
				rec[10] = (uint8_t)(flen >> 16); rec[11] = (uint8_t)(flen >> 24);
				memcpy(rec + 12, raw + q, flen);
				q += flen;
				k++;
			}
		}
		s->nrec = nrec;
		s->in_bytes = in_bytes;
		s->out_bytes = out_bytes;
		{
			int rc = 0;
			memset(s->status.h, 0, nrec * 4);
			rc |= gpumt_memcpy_h2d(g, s->in.d, s->in.h, in_bytes, 0);
			rc |= gpumt_memcpy_h2d(g, s->meta.d, s->meta.h, D_META_BYTES(BATCH_MAXREC), 0);
			rc |= gpumt_memcpy_h2d(g, s->status.d, s->status.h, nrec * 4, 0);
			rc |= gpumt_zstd_decompress_batch(g, s->in.d, in_bytes, m_rec_off(s, 1), m_rec_len(s, 1), nrec, s->out.d,
							  out_bytes, m_out_off(s, 1), m_out_len(s, 1), (uint32_t *)s->status.d, 0);
			rc |= gpumt_memcpy_d2h(g, s->status.h, s->status.d, nrec * 4, 0);
			rc |= gpumt_memcpy_d2h(g, m_out_len(s, 0), m_out_len(s, 1), nrec * 4, 0); /* sizes of unsized frames */
			if (out_bytes)
				rc |= gpumt_memcpy_d2h(g, s->out.h, s->out.d, out_bytes, 0);
			rc |= gpumt_stream_sync(g, 0);
			if (rc) {
				err = ZSTDCB_ERROR(compression_library);
				break;
			}
		}
		for (size_t i = 0; i < nrec && !err; i++) {
			const uint32_t st = ((const uint32_t *)s->status.h)[i];
			if (st != GPUMT_ST_OK) {
				zstdmt_errcode = st;
				err = ZSTDCB_ERROR(compression_library);
				break;
			}
			err = plain_write(ctx, io, (const uint8_t *)s->out.h + m_out_off(s, 0)[i], m_out_len(s, 0)[i]);
		}
		ip = jp;
		if (need_more)
			break;
	}
	if (err || (eof && ip >= n))
		break;
	if (need_more && n - ip >= want_ahead)
		want_ahead = (n - ip) * 2; /* a frame larger than what is buffered: read on */
	}
	free(raw);
	return err;
}

size_t ZSTDCB_decompressDCtx(ZSTDCB_DCtx *ctx, ZSTDCB_RdWr_t *rdwr)
{
	uint8_t sniff[16];
	ZSTDCB_Buffer b;
	static const mt_pipe_ops ops = {dp_fill, dp_launch, dp_complete, dp_drain, dp_role_start};
	size_t err;
	int rv;

	if (!ctx)
		return ZSTDCB_ERROR(compressionParameter_unsupported); /* zstd-mt_decompress.c:703-704 */
	/* 16-byte sniff (:711-760) */
	b.buf = sniff;
	b.size = 16;
	b.allocated = 16;
	rv = rdwr->fn_read(rdwr->arg_read, &b);
	if (rv != 0)
		return mt_error(rv);
	if (b.size < 16) {
		if (b.size < 4 || !is_zstd_magic(sniff))
			return ZSTDCB_ERROR(data_error);
		if (b.size == 9)
			return 0; /* the 9-byte empty zstd frame: "create empty file" (:731-736) */
		return plain_decompress(ctx, rdwr, sniff, b.size, 1); /* a plain .zst shorter than the sniff */
	}
	ctx->insize += 16;
	ctx->have_hdr = 1;
	if (rd32(sniff) == ZSTDCB_MAGIC_SKIPPABLE && is_zstd_magic(sniff + 12)) {
		/* "pzstd style", what the compressor writes: the sniff is the first record's header + 4
		 * payload bytes (:251-284) */
		ctx->hdr_csize = rd32(sniff + 8);
		memcpy(ctx->first4, sniff + 12, 4);
		ctx->have_first4 = 1;
	} else if (is_zstd_magic(sniff) && rd32(sniff + 9) == ZSTDCB_MAGIC_SKIPPABLE) {
		/* old "zstdmt style": a 9-byte empty zstd frame, then ordinary records (:225-249).  The
		 * first header straddles the sniff: 5 more bytes complete it.  (insize: the reference
		 * counts the sniff twice here, :221 and :239; kept.) */
		uint8_t tail[5];
		b.buf = tail;
		b.size = 5;
		b.allocated = 5;
		rv = rdwr->fn_read(rdwr->arg_read, &b);
		if (rv != 0)
			return mt_error(rv);
		if (b.size != 5)
			return ZSTDCB_ERROR(data_error);
		ctx->insize += 16 + 5;
		ctx->hdr_csize = rd32(tail + 1);
		ctx->have_first4 = 0;
	} else {
		if (is_zstd_magic(sniff)) {
			ctx->insize -= 16; /* counted with the rest of the input below */
			return plain_decompress(ctx, rdwr, sniff, 16, 0); /* "some std zstd stream" (:755-759) */
		}
		return ZSTDCB_ERROR(data_error);
	}
	ctx->budget = BATCH_MIN;
	ctx->big_out = 0;
	ctx->io = rdwr;
	/* threads == 1: every callback on the calling thread, as the reference (its single-thread path) */
	err = ctx->threads == 1 ? mt_pipe_run_inline(&ops, ctx) : mt_pipe_run_n(&ops, ctx, mt_nslot_for(ctx->gpus.n));
	mt_gpus_sync(&ctx->gpus);
	return err;
}
