/*
 * brotlimt_engine.c -- the host side of brotli-mt on MI355X: BROTLIMT_* (include/brotli-mt.h) over
 * gpumt_*.
 *
 * Decompression follows the callback-visible behaviour of the reference
 * (lib/brotli-mt_decompress.c:187-454): a 4-byte sniff that must be the skippable magic (:397-407),
 * 12 more header bytes for the first record and 16 for every later one (:197-224), the payload in
 * one read (:255-262), one fn_write per record in order; the output capacity of a record is
 * hint << 16 and nothing else (:236-239), a stream that does not decode into it fails with
 * frame_decompress (:348-351).  Records move through the same three-role batch pipeline as the other
 * codecs (mt_pipe.h): H2D / decode kernel / D2H on three streams.
 * Compression: pt_compress behaviour over the device encoder, see below.
 * Plain C, no HIP header.
 */
#include "mt_host.h"
#include "mt_pipe.h"
#include "brotli-mt.h"

/* ------------------------------------------------------------------ errors (brotli-mt_common.c) */
unsigned BROTLIMT_isError(size_t code)
{
	return code > BROTLIMT_ERROR(maxCode);
}

const char *BROTLIMT_getErrorString(size_t code)
{
	/* strings of lib/brotli-mt_common.c:37-57 */
	switch ((BROTLIMT_ErrorCode)((size_t)0 - code)) {
	case BROTLIMT_error_no_error:
		return "No error detected";
	case BROTLIMT_error_memory_allocation:
		return "Allocation error : not enough memory";
	case BROTLIMT_error_read_fail:
		return "Read failure";
	case BROTLIMT_error_write_fail:
		return "Write failure";
	case BROTLIMT_error_data_error:
		return "Malformed input";
	case BROTLIMT_error_frame_compress:
		return "Could not compress frame at once";
	case BROTLIMT_error_frame_decompress:
		return "Could not decompress frame at once";
	case BROTLIMT_error_compressionParameter_unsupported:
		return "Compression parameter is out of bound";
	default:
		return "Unspecified brotli error code";
	}
}

/* callback return value -> library error (mt_error, brotli-mt_decompress.c:142-155) */
static size_t mt_error(int rv)
{
	switch (rv) {
	case -1:
		return BROTLIMT_ERROR(read_fail);
	case -2:
		return BROTLIMT_ERROR(canceled);
	case -3:
		return BROTLIMT_ERROR(memory_allocation);
	}
	return BROTLIMT_ERROR(read_fail);
}

/* =================================================================== compression
 * pt_compress of the reference (lib/brotli-mt_compress.c:194-318): one fn_read of exactly
 * `inputsize` per chunk, EOF = a zero-length read once a frame exists (an empty input still yields
 * one record), a short read becomes a short record and the loop goes on; one fn_write per record
 * in order: 16-byte header (hint = 64 KiB units the decoder must provide, :294-304) + one brotli
 * stream.  The streams come from the device encoder (gpumt_brotli_compress_batch) and are
 * decompress-identical to the input; `level` is validated and sets the default chunk size. */
struct cslot {
	dbuf in;      /* chunk data, H2D                       */
	dbuf slots;   /* device only: per-chunk records        */
	dbuf stream;  /* packed records, D2H                   */
	dbuf meta;    /* rec_len[n] u32 | pad | rec_off[n+1] u64, D2H */
	size_t n;     /* bytes in the batch                    */
	size_t nrec;
};

struct BROTLIMT_CCtx_s {
	int level, threads, inputsize;
	size_t insize, outsize, curframe, frames; /* insize / frames: reader; outsize / curframe: writer */
	mt_gpus gpus; /* the devices the batch slots are dealt out to (mt_host.h) */
	struct cslot s[MT_NSLOT];
	BROTLIMT_RdWr_t *io;
	size_t maxrec;
};

BROTLIMT_CCtx *BROTLIMT_createCCtx(int threads, int level, int inputsize)
{
	BROTLIMT_CCtx *ctx;
	if (threads < 1 || threads > BROTLIMT_THREAD_MAX)
		return NULL;
	if (level < BROTLIMT_LEVEL_MIN || level > BROTLIMT_LEVEL_MAX)
		return NULL;
	if (inputsize < 0)
		return NULL;
	ctx = (BROTLIMT_CCtx *)calloc(1, sizeof *ctx);
	if (!ctx)
		return NULL;
	ctx->threads = threads;
	ctx->level = level;
	ctx->inputsize = inputsize ? inputsize : 1024 * 1024 * (level ? level : 1); /* :105-109 */
	if (mt_gpus_open(&ctx->gpus)) {
		free(ctx); /* no device: fail loudly, there is no CPU path */
		return NULL;
	}
	return ctx;
}

void BROTLIMT_freeCCtx(BROTLIMT_CCtx *ctx)
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

size_t BROTLIMT_GetFramesCCtx(BROTLIMT_CCtx *ctx) { return ctx ? ctx->curframe : 0; }
size_t BROTLIMT_GetInsizeCCtx(BROTLIMT_CCtx *ctx) { return ctx ? ctx->insize : 0; }
size_t BROTLIMT_GetOutsizeCCtx(BROTLIMT_CCtx *ctx) { return ctx ? ctx->outsize : 0; }

static size_t c_read_batch(BROTLIMT_CCtx *ctx, BROTLIMT_RdWr_t *io, struct cslot *s, size_t maxrec, int *eof)
{
	const size_t chunk = (size_t)ctx->inputsize;
	s->n = 0;
	s->nrec = 0;
	while (s->nrec < maxrec) {
		BROTLIMT_Buffer b;
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
			return BROTLIMT_ERROR(read_fail);
		ctx->insize += b.size;
		ctx->frames++;
		s->n += b.size;
		s->nrec++;
		if (b.size < chunk)
			break; /* ragged chunk: last one of this device batch */
	}
	return 0;
}

static size_t c_launch(BROTLIMT_CCtx *ctx, struct cslot *s)
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
	rc |= gpumt_brotli_compress_batch(g, s->in.d, s->n, chunk, s->slots.d, stride, d_len, ks);
	rc |= gpumt_lz4_compact(g, s->slots.d, stride, d_len, s->nrec, s->stream.d, d_off, ks);
	/* sizes, offsets and the packed records go to the pinned mirrors from the slot's own stream, the
	 * byte count of the records read on the device (d_off[nrec]): no host round trip in between, and
	 * the batches of the pipeline overlap (gpumt_push_host) */
	rc |= gpumt_push_host(g, s->meta.h, s->meta.d, ((s->nrec * 4 + 15) & ~(size_t)15) + (s->nrec + 1) * 8, NULL, ks);
	rc |= gpumt_push_host(g, s->stream.h, s->stream.d, s->stream.cap & ~(size_t)15, d_off + s->nrec, ks);
	return rc ? BROTLIMT_ERROR(frame_compress) : 0;
}

static size_t cp_fill(void *a, int si, int *has_data, int *eof)
{
	BROTLIMT_CCtx *ctx = (BROTLIMT_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const size_t chunk = (size_t)ctx->inputsize, stride = gpumt_zstd_slot_stride(chunk);
	size_t lim = BATCH_BYTES / chunk, err;
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
		return BROTLIMT_ERROR(memory_allocation);
	err = c_read_batch(ctx, ctx->io, s, ctx->maxrec, eof);
	*has_data = s->nrec > 0;
	ctx->maxrec *= 4;
	return err;
}

static size_t cp_launch(void *a, int si)
{
	BROTLIMT_CCtx *ctx = (BROTLIMT_CCtx *)a;
	size_t err = c_launch(ctx, &ctx->s[si]);
	if (!err && gpumt_mark(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si), mt_stream_of(&ctx->gpus, si)))
		err = BROTLIMT_ERROR(frame_compress);
	return err;
}

static size_t cp_complete(void *a, int si)
{
	BROTLIMT_CCtx *ctx = (BROTLIMT_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const uint64_t *off = (const uint64_t *)((const uint8_t *)s->meta.h + ((s->nrec * 4 + 15) & ~(size_t)15));
	size_t total;
	if (gpumt_mark_sync(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si)))
		return BROTLIMT_ERROR(frame_compress);
	total = (size_t)off[s->nrec];
	if (total > s->stream.cap)
		return BROTLIMT_ERROR(frame_compress);
	return 0;
}

static size_t cp_drain(void *a, int si)
{
	BROTLIMT_CCtx *ctx = (BROTLIMT_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const uint32_t *len = (const uint32_t *)s->meta.h;
	const uint64_t *off = (const uint64_t *)((const uint8_t *)s->meta.h + ((s->nrec * 4 + 15) & ~(size_t)15));
	for (size_t i = 0; i < s->nrec; i++) { /* pt_write: strictly in frame order */
		BROTLIMT_Buffer b;
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

size_t BROTLIMT_compressCCtx(BROTLIMT_CCtx *ctx, BROTLIMT_RdWr_t *rdwr)
{
	static const mt_pipe_ops ops = {cp_fill, cp_launch, cp_complete, cp_drain};
	size_t err;

	if (!ctx)
		return BROTLIMT_ERROR(compressionParameter_unsupported); /* brotli-mt_compress.c:325-326 */
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
	dbuf in;   /* record payloads back to back, H2D                                              */
	dbuf meta; /* rec_off u64[n] | out_off u64[n+1] | rec_len u32[n] | out_cap u32[n], H2D       */
	dbuf res;  /* out_len u32[n] | status u32[n], D2H                                            */
	dbuf out;  /* one slot of hint << 16 bytes per record, D2H                                   */
	size_t nrec, in_bytes, out_bytes;
};

struct BROTLIMT_DCtx_s {
	int threads, inputsize;
	size_t budget;
	size_t insize, outsize, curframe, frames;
	mt_gpus gpus; /* the devices the batch slots are dealt out to (mt_host.h) */
	struct dslot s[MT_NSLOT];
	BROTLIMT_RdWr_t *io;
	int have_hdr; /* a record header read ahead of its batch */
	uint32_t hdr_csize, hdr_hint;
	int first; /* the next header is the first one: its magic came with the sniff */
};

BROTLIMT_DCtx *BROTLIMT_createDCtx(int threads, int inputsize)
{
	BROTLIMT_DCtx *ctx;
	if (threads < 1 || threads > BROTLIMT_THREAD_MAX)
		return NULL;
	ctx = (BROTLIMT_DCtx *)calloc(1, sizeof *ctx);
	if (!ctx)
		return NULL;
	ctx->threads = threads;
	ctx->inputsize = inputsize ? inputsize : 1024 * 64; /* brotli-mt_decompress.c:110-113 */
	if (mt_gpus_open(&ctx->gpus)) {
		free(ctx);
		return NULL;
	}
	return ctx;
}

void BROTLIMT_freeDCtx(BROTLIMT_DCtx *ctx)
{
	if (!ctx)
		return;
	for (int i = 0; i < MT_NSLOT; i++) {
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].in);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].meta);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].res);
		dbuf_free(mt_gpu_of(&ctx->gpus, i), &ctx->s[i].out);
	}
	mt_gpus_close(&ctx->gpus);
	free(ctx);
}

size_t BROTLIMT_GetFramesDCtx(BROTLIMT_DCtx *ctx) { return ctx ? ctx->curframe : 0; }
size_t BROTLIMT_GetInsizeDCtx(BROTLIMT_DCtx *ctx) { return ctx ? ctx->insize : 0; }
size_t BROTLIMT_GetOutsizeDCtx(BROTLIMT_DCtx *ctx) { return ctx ? ctx->outsize : 0; }

#define D_META_BYTES(n) ((n) * 8 + ((n) + 1) * 8 + (n) * 4 + (n) * 4 + 64)
static uint64_t *m_rec_off(struct dslot *s, int dev) { return (uint64_t *)(dev ? s->meta.d : s->meta.h); }
static uint64_t *m_out_off(struct dslot *s, int dev) { return m_rec_off(s, dev) + BATCH_MAXREC; }
static uint32_t *m_rec_len(struct dslot *s, int dev) { return (uint32_t *)(m_out_off(s, dev) + BATCH_MAXREC + 1); }
static uint32_t *m_out_cap(struct dslot *s, int dev) { return m_rec_len(s, dev) + BATCH_MAXREC; }
static uint32_t *r_out_len(struct dslot *s, int dev) { return (uint32_t *)(dev ? s->res.d : s->res.h); }
static uint32_t *r_status(struct dslot *s, int dev) { return r_out_len(s, dev) + BATCH_MAXREC; }

/* next record header (pt_read, brotli-mt_decompress.c:193-240): 12 bytes after the sniff, else 16 */
static size_t d_read_header(BROTLIMT_DCtx *ctx, BROTLIMT_RdWr_t *io, uint32_t *csize, uint32_t *hint, int *eof)
{
	uint8_t hb[16];
	BROTLIMT_Buffer b;
	int rv;
	const size_t want = ctx->first ? 12 : 16;
	b.buf = ctx->first ? hb + 4 : hb;
	b.size = want;
	b.allocated = want;
	rv = io->fn_read(io->arg_read, &b);
	if (rv != 0)
		return mt_error(rv);
	if (!ctx->first && b.size == 0) {
		*eof = 1;
		return 0;
	}
	if (b.size != want)
		return BROTLIMT_ERROR(read_fail);
	if (!ctx->first && rd32(hb) != BROTLIMT_MAGIC_SKIPPABLE)
		return BROTLIMT_ERROR(data_error);
	ctx->first = 0;
	if (rd32(hb + 4) != 8)
		return BROTLIMT_ERROR(data_error);
	if (((uint32_t)hb[12] | (uint32_t)hb[13] << 8) != BROTLIMT_MAGICNUMBER)
		return BROTLIMT_ERROR(data_error);
	ctx->insize += 16;
	*csize = rd32(hb + 8);
	*hint = (uint32_t)hb[14] | (uint32_t)hb[15] << 8;
	return 0;
}

/*
 * One wave decodes one record, and a 1 MiB record keeps it busy for ~165 ms whatever else runs, so
 * the device is only full with thousands of records in flight: the output budget of a batch grows
 * to four times the common batch size (1 GiB by default = 1 024 records of the level-1 chunk size).
 * Compressed bytes a batch may hold: the budget plus an eighth (records that did not shrink).
 */
#define D_BATCH_BYTES (4 * BATCH_BYTES)
#define D_IN_LIMIT(budget) ((budget) + ((budget) >> 3))

static size_t d_read_batch(BROTLIMT_DCtx *ctx, BROTLIMT_RdWr_t *io, struct dslot *s, int *eof)
{
	s->nrec = 0;
	s->in_bytes = 0;
	s->out_bytes = 0;
	while (s->nrec < BATCH_MAXREC) {
		uint32_t csize = 0, hint = 0;
		BROTLIMT_Buffer b;
		size_t err, cap;
		int rv;
		if (ctx->have_hdr) {
			csize = ctx->hdr_csize;
			hint = ctx->hdr_hint;
		} else {
			err = d_read_header(ctx, io, &csize, &hint, eof);
			if (err)
				return err;
			if (*eof)
				break;
		}
		cap = (size_t)hint << 16;
		if (s->nrec && (s->in_bytes + (size_t)csize > D_IN_LIMIT(ctx->budget) || s->out_bytes + cap > ctx->budget)) {
			ctx->have_hdr = 1;
			ctx->hdr_csize = csize;
			ctx->hdr_hint = hint;
			break;
		}
		ctx->have_hdr = 0;
		if (s->in_bytes + (size_t)csize + 512 > s->in.cap) {
			/* the record buffer starts at half the output budget (text shrinks more than 2:1) and
			 * doubles when a batch needs more */
			dbuf old = s->in;
			size_t want = s->in_bytes + (size_t)csize + 512;
			if (want < 2 * old.cap)
				want = 2 * old.cap;
			memset(&s->in, 0, sizeof s->in);
			if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, want, 1, 1)) {
				dbuf_free(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in);
				s->in = old; /* keep the slot as it was: freeCtx releases it */
				return BROTLIMT_ERROR(memory_allocation);
			}
			memcpy(s->in.h, old.h, s->in_bytes);
			dbuf_free(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &old);
		}
		b.buf = (uint8_t *)s->in.h + s->in_bytes;
		b.size = csize;
		b.allocated = csize;
		rv = io->fn_read(io->arg_read, &b);
		if (rv != 0)
			return mt_error(rv);
		if (b.size != csize)
			return BROTLIMT_ERROR(data_error); /* "needed more bytes!" (:259-260) */
		ctx->insize += csize;
		ctx->frames++;
		m_rec_off(s, 0)[s->nrec] = s->in_bytes;
		m_rec_len(s, 0)[s->nrec] = csize;
		m_out_off(s, 0)[s->nrec] = s->out_bytes;
		m_out_cap(s, 0)[s->nrec] = (uint32_t)cap;
		s->in_bytes += csize;
		s->out_bytes += cap;
		s->nrec++;
	}
	m_out_off(s, 0)[s->nrec] = s->out_bytes;
	return 0;
}

static size_t d_launch(BROTLIMT_DCtx *ctx, struct dslot *s)
{
	gpumt_ctx *g = mt_gpu_of(&ctx->gpus, (int)(s - ctx->s));
	/* each batch slot launches on its own kernel stream (4 + slot): the decoders are bound by the
	 * latency of a record, so the batches of the pipeline must overlap on the device */
	const int ks = mt_stream_of(&ctx->gpus, (int)(s - ctx->s));
	int rc = 0;
	if (dbuf_want(g, &s->out, s->out_bytes + 64, 1, 1) || dbuf_want(g, &s->res, BATCH_MAXREC * 8 + 64, 1, 1))
		return BROTLIMT_ERROR(memory_allocation);
	rc |= gpumt_memcpy_h2d(g, s->in.d, s->in.h, s->in_bytes + 256, 1);
	rc |= gpumt_memcpy_h2d(g, s->meta.d, s->meta.h, D_META_BYTES(BATCH_MAXREC), 1);
	rc |= gpumt_stream_wait(g, ks, 1);
	rc |= gpumt_brotli_decompress_batch(g, s->in.d, m_rec_off(s, 1), m_rec_len(s, 1), s->nrec, s->out.d,
					    m_out_off(s, 1), m_out_cap(s, 1), r_out_len(s, 1), r_status(s, 1), ks);
	rc |= gpumt_stream_wait(g, 2, ks);
	rc |= gpumt_memcpy_d2h(g, s->res.h, s->res.d, BATCH_MAXREC * 8, 2);
	if (s->out_bytes)
		rc |= gpumt_memcpy_d2h(g, s->out.h, s->out.d, s->out_bytes, 2);
	return rc ? BROTLIMT_ERROR(frame_decompress) : 0;
}

static size_t dp_fill(void *a, int si, int *has_data, int *eof)
{
	BROTLIMT_DCtx *ctx = (BROTLIMT_DCtx *)a;
	struct dslot *s = &ctx->s[si];
	size_t err;
	if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, (ctx->budget >> 1) + 4096, 1, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->meta, D_META_BYTES(BATCH_MAXREC), 1, 1))
		return BROTLIMT_ERROR(memory_allocation);
	err = d_read_batch(ctx, ctx->io, s, eof);
	*has_data = s->nrec > 0;
	if (ctx->budget < D_BATCH_BYTES)
		ctx->budget *= 4;
	return err;
}

static size_t dp_launch(void *a, int si)
{
	BROTLIMT_DCtx *ctx = (BROTLIMT_DCtx *)a;
	size_t err = d_launch(ctx, &ctx->s[si]);
	if (!err && gpumt_mark(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si), 2))
		err = BROTLIMT_ERROR(frame_decompress);
	return err;
}

static size_t dp_complete(void *a, int si)
{
	BROTLIMT_DCtx *ctx = (BROTLIMT_DCtx *)a;
	return gpumt_mark_sync(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si)) ? BROTLIMT_ERROR(frame_decompress) : 0;
}

static size_t dp_drain(void *a, int si)
{
	BROTLIMT_DCtx *ctx = (BROTLIMT_DCtx *)a;
	struct dslot *s = &ctx->s[si];
	const uint32_t *st = r_status(s, 0), *ol = r_out_len(s, 0);
	for (size_t i = 0; i < s->nrec; i++) {
		BROTLIMT_Buffer b;
		int rv;
		if (st[i] != GPUMT_ST_OK)
			return BROTLIMT_ERROR(frame_decompress); /* pt_decompress :348-351 */
		b.buf = (uint8_t *)s->out.h + m_out_off(s, 0)[i];
		b.size = ol[i];
		b.allocated = m_out_cap(s, 0)[i];
		rv = ctx->io->fn_write(ctx->io->arg_write, &b);
		if (rv != 0)
			return mt_error(rv);
		ctx->outsize += b.size;
		ctx->curframe++;
	}
	return 0;
}

size_t BROTLIMT_decompressDCtx(BROTLIMT_DCtx *ctx, BROTLIMT_RdWr_t *rdwr)
{
	uint8_t sniff[4];
	BROTLIMT_Buffer b;
	static const mt_pipe_ops ops = {dp_fill, dp_launch, dp_complete, dp_drain};
	size_t err;
	int rv;

	if (!ctx)
		return BROTLIMT_ERROR(compressionParameter_unsupported); /* brotli-mt_decompress.c:387-388 */
	/* 4-byte sniff: only the skippable-frame layout exists for brotli (:397-407) */
	b.buf = sniff;
	b.size = 4;
	b.allocated = 4;
	rv = rdwr->fn_read(rdwr->arg_read, &b);
	if (rv != 0)
		return mt_error(rv);
	if (b.size != 4)
		return BROTLIMT_ERROR(data_error);
	if (rd32(sniff) != BROTLIMT_MAGIC_SKIPPABLE)
		return BROTLIMT_ERROR(data_error);
	ctx->first = 1;
	ctx->have_hdr = 0;
	ctx->budget = BATCH_MIN;
	ctx->io = rdwr;
	/* threads == 1: every callback on the calling thread, as the reference (its single-thread path) */
	err = ctx->threads == 1 ? mt_pipe_run_inline(&ops, ctx) : mt_pipe_run_n(&ops, ctx, mt_nslot_for(ctx->gpus.n));
	mt_gpus_sync(&ctx->gpus);
	return err;
}
