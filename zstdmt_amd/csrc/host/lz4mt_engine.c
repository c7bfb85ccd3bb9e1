/*
 * lz4mt_engine.c -- the host side of lz4-mt on MI355X: LZ4MT_* (include/lz4-mt.h) over gpumt_*.
 *
 * Replaces the pthread worker pool of the reference (lib/lz4-mt_compress.c:207-353,
 * lib/lz4-mt_decompress.c:165-567): instead of T threads each pulling one chunk through the codec,
 * one host thread moves *batches* of chunks through a three-stage device pipeline
 *
 *        fn_read -> pinned in[s]  --H2D (stream 1)-->  kernels (stream 0)  --D2H (stream 2)-->
 *        pinned out[s] -> fn_write
 *
 * over the four slots of mt_pipe.h (up to three batches on the device, decompress on per-slot
 * kernel streams, while another is read into or written out).  Callback-visible behaviour follows the reference: compress issues one fn_read of
 * exactly `inputsize` bytes per chunk and one fn_write per record, in input order; decompress
 * reads 4, then 8|12 header bytes and the payload per record, and writes one chunk per call.
 * This file is plain C and never includes a HIP header.
 */
#include "mt_host.h"
#include "mt_pipe.h"
#include "lz4-mt.h"

size_t lz4mt_errcode;

/* ------------------------------------------------------------------ errors (lz4-mt_common.c) */
unsigned LZ4MT_isError(size_t code)
{
	return code > ERROR(maxCode);
}

const char *LZ4MT_getErrorString(size_t code)
{
	static const char *const names[] = {
		"No error detected",
		"Allocation error : not enough memory",
		"Read failure",
		"Write failure",
		"Malformed input",
		"Could not compress frame at once",
		"Could not decompress frame at once",
		"Compression parameter is out of bound",
		"Compression library reports failure",
		"Unspecified lz4mt error code", /* canceled has no text in the reference either */
	};
	static const char *const codec[] = {
		"", "device: malformed record header", "device: bad LZ4 frame header",
		"device: malformed LZ4 block", "device: content size mismatch",
		"device: content checksum mismatch", "device: trailing bytes after frame",
		"device: unsupported LZ4 frame feature",
	};
	size_t idx = (size_t)0 - code;
	/* like the reference, a pending codec-level error takes precedence */
	if (lz4mt_errcode >= 1 && lz4mt_errcode <= 7 && idx == LZ4MT_error_compression_library)
		return codec[lz4mt_errcode];
	if (idx < LZ4MT_error_canceled)
		return names[idx];
	return names[9];
}

/* callback return value -> library error (reference mt_error, lz4-mt_compress.c:161-173; note
 * that write failures go through the same mapping and so surface as read_fail) */
static size_t mt_error(int rv)
{
	switch (rv) {
	case -1:
		return ERROR(read_fail);
	case -2:
		return ERROR(canceled);
	case -3:
		return ERROR(memory_allocation);
	}
	return ERROR(read_fail);
}

struct cslot {
	dbuf in;      /* chunk data, H2D                       */
	dbuf slots;   /* device only: per-chunk records        */
	dbuf stream;  /* packed records, D2H                   */
	dbuf meta;    /* rec_len[n] u32 | pad | rec_off[n+1] u64, D2H */
	size_t n;     /* bytes in the batch                    */
	size_t nrec;
};

struct LZ4MT_CCtx_s {
	int level, threads, inputsize;
	size_t insize, outsize, curframe, frames; /* insize / frames: reader; outsize / curframe: writer */
	mt_gpus gpus; /* the devices the batch slots are dealt out to (mt_host.h) */
	struct cslot s[MT_NSLOT];
	LZ4MT_RdWr_t *io; /* callbacks of the running call */
	size_t maxrec;    /* records per device batch, grows (reader) */
};

LZ4MT_CCtx *LZ4MT_createCCtx(int threads, int level, int inputsize)
{
	LZ4MT_CCtx *ctx;
	if (threads < 1 || threads > LZ4MT_THREAD_MAX)
		return NULL;
	if (level < LZ4MT_LEVEL_MIN || level > LZ4MT_LEVEL_MAX)
		return NULL;
	if (inputsize < 0)
		return NULL;
	ctx = (LZ4MT_CCtx *)calloc(1, sizeof *ctx);
	if (!ctx)
		return NULL;
	ctx->level = level;
	ctx->threads = threads;
	ctx->inputsize = inputsize ? inputsize : 1024 * 1024 * 4; /* lz4-mt_compress.c:111-114 */
	if (mt_gpus_open(&ctx->gpus)) {
		free(ctx); /* no device: fail loudly, there is no CPU path */
		return NULL;
	}
	return ctx;
}

void LZ4MT_freeCCtx(LZ4MT_CCtx *ctx)
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

size_t LZ4MT_GetFramesCCtx(LZ4MT_CCtx *ctx) { return ctx ? ctx->curframe : 0; }
size_t LZ4MT_GetInsizeCCtx(LZ4MT_CCtx *ctx) { return ctx ? ctx->insize : 0; }
size_t LZ4MT_GetOutsizeCCtx(LZ4MT_CCtx *ctx) { return ctx ? ctx->outsize : 0; }

/*
 * Fill slot s with up to `maxrec` chunks.  Returns 0, or an error code; *eof is set when the
 * input is exhausted.  A short (non-zero) read is a short chunk and closes the batch, exactly
 * one fn_read per chunk as in pt_compress (lz4-mt_compress.c:256-277).
 */
static size_t c_read_batch(LZ4MT_CCtx *ctx, LZ4MT_RdWr_t *io, struct cslot *s, size_t maxrec, int *eof)
{
	const size_t chunk = (size_t)ctx->inputsize;
	s->n = 0;
	s->nrec = 0;
	while (s->nrec < maxrec) {
		LZ4MT_Buffer b;
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
			return ERROR(read_fail);
		ctx->insize += b.size;
		ctx->frames++;
		s->n += b.size;
		s->nrec++;
		if (b.size < chunk)
			break; /* ragged chunk (or the empty first read, which still yields one empty
				* frame): it must be the last one of this device batch; reading goes on
				* with the next batch, as the reference's loop does */
	}
	return 0;
}

static size_t c_launch(LZ4MT_CCtx *ctx, struct cslot *s)
{
	gpumt_ctx *g = mt_gpu_of(&ctx->gpus, (int)(s - ctx->s));
	const int ks = mt_stream_of(&ctx->gpus, (int)(s - ctx->s)); /* the slot's own kernel stream: batches overlap on the device */
	const size_t chunk = (size_t)ctx->inputsize;
	const size_t stride = gpumt_lz4_slot_stride(chunk);
	uint32_t *d_len = (uint32_t *)s->meta.d;
	uint64_t *d_off = (uint64_t *)((uint8_t *)s->meta.d + ((s->nrec * 4 + 15) & ~(size_t)15));
	int rc = 0;
	if (s->n)
		rc |= gpumt_memcpy_h2d(g, s->in.d, s->in.h, s->n, 1);
	rc |= gpumt_stream_wait(g, ks, 1);
	rc |= gpumt_lz4_compress_batch_level(g, s->in.d, s->n, chunk, s->slots.d, stride, d_len, ctx->level, ks);
	rc |= gpumt_lz4_compact(g, s->slots.d, stride, d_len, s->nrec, s->stream.d, d_off, ks);
	/* sizes, offsets and the packed records go to the pinned mirrors from the slot's own stream, the
	 * byte count of the records read on the device (d_off[nrec]): no host round trip in between, and
	 * the batches of the pipeline overlap (gpumt_push_host) */
	rc |= gpumt_push_host(g, s->meta.h, s->meta.d, ((s->nrec * 4 + 15) & ~(size_t)15) + (s->nrec + 1) * 8, NULL, ks);
	rc |= gpumt_push_host(g, s->stream.h, s->stream.d, s->stream.cap & ~(size_t)15, d_off + s->nrec, ks);
	return rc ? ERROR(compression_library) : 0;
}

/* ---- the three roles (mt_pipe.h) ---- */
static void cp_role_start(void *a) { mt_bind_near(&((LZ4MT_CCtx *)a)->gpus); }
static size_t cp_fill(void *a, int si, int *has_data, int *eof)
{
	LZ4MT_CCtx *ctx = (LZ4MT_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const size_t chunk = (size_t)ctx->inputsize, stride = gpumt_lz4_slot_stride(chunk);
	/* (twice the chunk: the LZ4 encoder is one wave per chunk at ~20 MB/s -- 512 chunks per batch, 3 batches in flight, is
	 * where the device comes near the reader: 12.3 -> 16.8 GB/s at 1 MiB chunks; the other codecs' encoders put several waves
	 * on a chunk, and the decompress leg lost 2 GB/s to the coarser pipeline with the same rule) */
	size_t lim = zmt_batch_bytes_for(2 * chunk) / chunk, err;
	if (lim < 1)
		lim = 1;
	if (lim > BATCH_MAXREC)
		lim = BATCH_MAXREC;
	if (ctx->maxrec > lim)
		ctx->maxrec = lim;
	/* (re)size this slot for the current batch size; it is free: nothing of it is in flight */
	if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, ctx->maxrec * chunk + 512, 1, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->slots, ctx->maxrec * stride, 0, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->stream, ctx->maxrec * stride + 512, 1, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->meta, ctx->maxrec * 12 + 64, 1, 1))
		return ERROR(memory_allocation);
	err = c_read_batch(ctx, ctx->io, s, ctx->maxrec, eof);
	*has_data = s->nrec > 0;
	ctx->maxrec *= 4;
	return err;
}

static size_t cp_launch(void *a, int si)
{
	LZ4MT_CCtx *ctx = (LZ4MT_CCtx *)a;
	mt_trace_launch(&ctx->gpus, "lz4mt compress", si, ctx->s[si].nrec);
	size_t err = c_launch(ctx, &ctx->s[si]);
	if (!err && gpumt_mark(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si), mt_stream_of(&ctx->gpus, si)))
		err = ERROR(compression_library);
	return err;
}

static size_t cp_complete(void *a, int si)
{
	LZ4MT_CCtx *ctx = (LZ4MT_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const uint64_t *off = (const uint64_t *)((const uint8_t *)s->meta.h + ((s->nrec * 4 + 15) & ~(size_t)15));
	size_t total;
	if (gpumt_mark_sync(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si))) /* record sizes and offsets are in host memory */
		return ERROR(compression_library);
	total = (size_t)off[s->nrec];
	if (total > s->stream.cap)
		return ERROR(frame_compress);
	return 0;
}

static size_t cp_drain(void *a, int si)
{
	LZ4MT_CCtx *ctx = (LZ4MT_CCtx *)a;
	struct cslot *s = &ctx->s[si];
	const uint32_t *len = (const uint32_t *)s->meta.h;
	const uint64_t *off = (const uint64_t *)((const uint8_t *)s->meta.h + ((s->nrec * 4 + 15) & ~(size_t)15));
	for (size_t i = 0; i < s->nrec; i++) { /* pt_write: strictly in frame order */
		LZ4MT_Buffer b;
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

size_t LZ4MT_compressCCtx(LZ4MT_CCtx *ctx, LZ4MT_RdWr_t *rdwr)
{
	static const mt_pipe_ops ops = {cp_fill, cp_launch, cp_complete, cp_drain, cp_role_start};
	size_t err;

	if (!ctx)
		return ERROR(compressionParameter_unsupported); /* lz4-mt_compress.c:317-318 */
	if (!gpumt_lz4_level_supported(ctx->level)) /* cannot happen for a context createCCtx accepted */
		return ERROR(compressionParameter_unsupported);
	ctx->io = rdwr;
	ctx->maxrec = BATCH_MIN / (size_t)ctx->inputsize;
	if (ctx->maxrec < 1)
		ctx->maxrec = 1;
	/* the reference keeps its counters across calls; so do we */
	err = mt_pipe_run_n(&ops, ctx, mt_nslot_for(ctx->gpus.n));
	mt_gpus_sync(&ctx->gpus);
	return err;
}

/* ================================================================= decompression ============ */
struct dslot {
	dbuf in;     /* record bytes (headers included), H2D                                  */
	dbuf meta;   /* rec_off u64[n] | out_off u64[n+1] | rec_len u32[n] | out_len u32[n], H2D */
	dbuf status; /* u32[n], D2H                                                           */
	dbuf out;    /* decoded chunks, D2H                                                   */
	size_t nrec, in_bytes, out_bytes;
};

struct LZ4MT_DCtx_s {
	int threads, inputsize;
	size_t budget; /* output bytes per device batch, grows from BATCH_MIN to zmt_batch_bytes_for(the largest record seen) */
	size_t big_out;
	size_t insize, outsize, curframe, frames;
	mt_gpus gpus; /* the devices the batch slots are dealt out to (mt_host.h) */
	struct dslot s[MT_NSLOT];
	LZ4MT_RdWr_t *io;
	/* a record header read ahead of its batch */
	int have_hdr;
	uint32_t hdr_csize;
	/* the next header is the stream's first: its magic went with the sniff.  (The reference tests
	 * frames == 0 for this, lz4-mt_decompress.c:200, so its DCtx decodes one stream only -- the
	 * counters carry over and a second call fails with data_error; here a DCtx can be used again.) */
	int first_hdr;
};

LZ4MT_DCtx *LZ4MT_createDCtx(int threads, int inputsize)
{
	LZ4MT_DCtx *ctx;
	if (threads < 1 || threads > LZ4MT_THREAD_MAX)
		return NULL;
	ctx = (LZ4MT_DCtx *)calloc(1, sizeof *ctx);
	if (!ctx)
		return NULL;
	ctx->threads = threads;
	ctx->inputsize = inputsize ? inputsize : 1024 + 1024 * 4; /* sic, lz4-mt_decompress.c:115 */
	if (mt_gpus_open(&ctx->gpus)) {
		free(ctx);
		return NULL;
	}
	return ctx;
}

void LZ4MT_freeDCtx(LZ4MT_DCtx *ctx)
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

size_t LZ4MT_GetFramesDCtx(LZ4MT_DCtx *ctx) { return ctx ? ctx->curframe : 0; }
size_t LZ4MT_GetInsizeDCtx(LZ4MT_DCtx *ctx) { return ctx ? ctx->insize : 0; }
size_t LZ4MT_GetOutsizeDCtx(LZ4MT_DCtx *ctx) { return ctx ? ctx->outsize : 0; }

#define D_META_BYTES(n) ((n) * 8 + ((n) + 1) * 8 + (n) * 4 + (n) * 4 + 64)

/* host view of a slot's meta arrays (capacity BATCH_MAXREC) */
static uint64_t *m_rec_off(struct dslot *s, int dev) { return (uint64_t *)(dev ? s->meta.d : s->meta.h); }
static uint64_t *m_out_off(struct dslot *s, int dev) { return m_rec_off(s, dev) + BATCH_MAXREC; }
static uint32_t *m_rec_len(struct dslot *s, int dev) { return (uint32_t *)(m_out_off(s, dev) + BATCH_MAXREC + 1); }
static uint32_t *m_out_len(struct dslot *s, int dev) { return m_rec_len(s, dev) + BATCH_MAXREC; }

/* read one record header (pt_read, lz4-mt_decompress.c:192-236): 0 = ok, *csize set; eof flagged */
static size_t d_read_header(LZ4MT_DCtx *ctx, LZ4MT_RdWr_t *io, uint32_t *csize, int *eof)
{
	uint8_t hb[12];
	LZ4MT_Buffer b;
	int rv;
	if (ctx->first_hdr) { /* magic already consumed by the sniff */
		ctx->first_hdr = 0;
		b.buf = hb + 4;
		b.size = 8;
		b.allocated = 8;
		rv = io->fn_read(io->arg_read, &b);
		if (rv != 0)
			return mt_error(rv);
		if (b.size != 8)
			return ERROR(read_fail);
	} else {
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
			return ERROR(read_fail);
		if (rd32(hb) != LZ4FMT_MAGIC_SKIPPABLE)
			return ERROR(data_error);
	}
	if (rd32(hb + 4) != 4)
		return ERROR(data_error);
	ctx->insize += 12;
	*csize = rd32(hb + 8);
	return 0;
}

static size_t d_read_batch(LZ4MT_DCtx *ctx, LZ4MT_RdWr_t *io, struct dslot *s, int *eof)
{
	s->nrec = 0;
	s->in_bytes = 0;
	s->out_bytes = 0;
	while (s->nrec < BATCH_MAXREC) {
		uint32_t csize = 0;
		uint64_t osz = 0;
		uint8_t *rec;
		LZ4MT_Buffer b;
		size_t err;
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
		/* close the batch when it is full; the header just read waits for the next one */
		if (s->nrec && (s->in_bytes + 12 + (size_t)csize > s->in.cap - 512 || s->out_bytes >= ctx->budget)) {
			ctx->have_hdr = 1;
			ctx->hdr_csize = csize;
			break;
		}
		ctx->have_hdr = 0;
		if (s->in_bytes + 12 + (size_t)csize + 512 > s->in.cap) {
			/* a single record larger than the slot: grow (nothing is in flight in this slot) */
			dbuf old = s->in;
			memset(&s->in, 0, sizeof s->in);
			if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, s->in_bytes + 12 + (size_t)csize + 512, 1, 1)) {
				dbuf_free(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in);
				s->in = old; /* keep the slot as it was: freeCtx releases it */
				return ERROR(memory_allocation);
			}
			memcpy(s->in.h, old.h, s->in_bytes);
			dbuf_free(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &old);
		}
		rec = (uint8_t *)s->in.h + s->in_bytes;
		/* rebuild the 12-byte header in front of the payload: the device checks it too */
		rec[0] = 0x50; rec[1] = 0x2A; rec[2] = 0x4D; rec[3] = 0x18;
		rec[4] = 4; rec[5] = rec[6] = rec[7] = 0;
		rec[8] = (uint8_t)csize; rec[9] = (uint8_t)(csize >> 8);
		rec[10] = (uint8_t)(csize >> 16); rec[11] = (uint8_t)(csize >> 24);
		b.buf = rec + 12;
		b.size = csize;
		b.allocated = csize;
		rv = io->fn_read(io->arg_read, &b);
		if (rv != 0)
			return mt_error(rv);
		if (b.size != csize)
			return ERROR(data_error); /* "needed more bytes" */
		ctx->insize += csize;
		ctx->frames++;
		/* output size = LE64 at payload+6 (lz4-mt_decompress.c:333-334); frames that carry no
		 * content size (only the empty frame is written that way) decode to nothing */
		if (csize >= 15 && (rec[12 + 4] & 0x08))
			osz = rd64(rec + 12 + 6);
		if (osz > 0xFFFFFFFFull)
			return ERROR(data_error);
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

static size_t d_launch(LZ4MT_DCtx *ctx, struct dslot *s)
{
	gpumt_ctx *g = mt_gpu_of(&ctx->gpus, (int)(s - ctx->s));
	/* each batch slot launches on its own kernel stream (4 + slot): the decoders are bound by the
	 * latency of a record, so the batches of the pipeline must overlap on the device */
	const int ks = mt_stream_of(&ctx->gpus, (int)(s - ctx->s));
	int rc = 0;
	if (dbuf_want(g, &s->out, s->out_bytes + 64, 1, 1) || dbuf_want(g, &s->status, s->nrec * 4 + 64, 1, 1))
		return ERROR(memory_allocation);
	rc |= gpumt_memcpy_h2d(g, s->in.d, s->in.h, s->in_bytes, 1);
	rc |= gpumt_memcpy_h2d(g, s->meta.d, s->meta.h, D_META_BYTES(BATCH_MAXREC), 1);
	rc |= gpumt_stream_wait(g, ks, 1);
	rc |= gpumt_lz4_decompress_batch(g, s->in.d, s->in_bytes, m_rec_off(s, 1), m_rec_len(s, 1), s->nrec,
					 s->out.d, s->out_bytes, m_out_off(s, 1), m_out_len(s, 1),
					 (uint32_t *)s->status.d, ks);
	rc |= gpumt_stream_wait(g, 2, ks);
	rc |= gpumt_memcpy_d2h(g, s->status.h, s->status.d, s->nrec * 4, 2);
	if (s->out_bytes)
		rc |= gpumt_memcpy_d2h(g, s->out.h, s->out.d, s->out_bytes, 2);
	return rc ? ERROR(compression_library) : 0;
}

static void dp_role_start(void *a) { mt_bind_near(&((LZ4MT_DCtx *)a)->gpus); }
static size_t dp_fill(void *a, int si, int *has_data, int *eof)
{
	LZ4MT_DCtx *ctx = (LZ4MT_DCtx *)a;
	struct dslot *s = &ctx->s[si];
	size_t err;
	/* input slot sized for the batch budget (compressed data is never larger than that plus
	 * per-record overhead); the slot is free here */
	if (dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->in, ctx->budget + (ctx->budget >> 3) + 4096, 1, 1) ||
	    dbuf_want(mt_gpu_of(&ctx->gpus, (int)(s - ctx->s)), &s->meta, D_META_BYTES(BATCH_MAXREC), 1, 1))
		return ERROR(memory_allocation);
	err = d_read_batch(ctx, ctx->io, s, eof);
	*has_data = s->nrec > 0;
	if (ctx->budget < zmt_batch_bytes_for(ctx->big_out))
		ctx->budget *= 4;
	return err;
}

static size_t dp_launch(void *a, int si)
{
	LZ4MT_DCtx *ctx = (LZ4MT_DCtx *)a;
	mt_trace_launch(&ctx->gpus, "lz4mt decompress", si, ctx->s[si].nrec);
	size_t err = d_launch(ctx, &ctx->s[si]);
	if (!err && gpumt_mark(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si), 2))
		err = ERROR(compression_library);
	return err;
}

static size_t dp_complete(void *a, int si)
{
	LZ4MT_DCtx *ctx = (LZ4MT_DCtx *)a;
	return gpumt_mark_sync(mt_gpu_of(&ctx->gpus, si), mt_mark_of(&ctx->gpus, si)) ? ERROR(compression_library) : 0;
}

static size_t dp_drain(void *a, int si)
{
	LZ4MT_DCtx *ctx = (LZ4MT_DCtx *)a;
	struct dslot *s = &ctx->s[si];
	const uint32_t *st = (const uint32_t *)s->status.h;
	for (size_t i = 0; i < s->nrec; i++) {
		LZ4MT_Buffer b;
		int rv;
		if (st[i] != GPUMT_ST_OK) {
			/* pt_decompress: LZ4F error -> compression_library (code kept in the global),
			 * frame not consumed exactly -> frame_decompress (lz4-mt_decompress.c:353-362) */
			if (st[i] == GPUMT_ST_BAD_RECORD)
				return ERROR(data_error);
			if (st[i] == GPUMT_ST_TRAILING)
				return ERROR(frame_decompress);
			lz4mt_errcode = st[i];
			return ERROR(compression_library);
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

/* =================================================================== plain .lz4 streams
 * A stream that starts with an LZ4 frame instead of a skippable record is decoded by the reference
 * on one thread with streaming LZ4F_decompress (st_decompress, lz4-mt_decompress.c:391-483): files of
 * the lz4 tool, any number of frames, typically 4 MiB linked blocks and no content size.  Here the
 * input is read about one batch ahead, split into frames on the host by walking the block headers (LZ4 frame
 * format: FLG / BD, optional content size and dictionary id, 4-byte block sizes, end mark, optional
 * content checksum) and decoded by the frame-serial kernel, one wave per frame; a frame that does not
 * state its content size gets blocks x block-maximum as capacity and the decoder reports the size.
 * GetFrames stays 0 as in the reference (st_decompress counts no frames). */
/* Length of the LZ4 frame at p (n bytes are there), or 0 = the frame is not complete yet (more input may
 * complete it), or EXTENT_INVALID = these bytes cannot become a frame however much follows (descriptor or
 * block size out of the format): the incremental reader stops at once instead of buffering the rest of a
 * damaged stream until its end. */
#define EXTENT_INVALID ((size_t)-1)
static size_t lz4_frame_extent(const uint8_t *p, size_t n, uint64_t *bound, int *supported)
{
	if (n < 7)
		return 0;
	const unsigned flg = p[4], bd = p[5];
	const unsigned bsid = (bd >> 4) & 7;
	const int has_csize = (flg >> 3) & 1, has_dict = flg & 1, bchk = (flg >> 4) & 1, cchk = (flg >> 2) & 1;
	size_t hp = 6 + (has_csize ? 8 : 0) + (has_dict ? 4 : 0) + 1;
	uint64_t sum = 0, blkmax;
	if ((flg >> 6) != 1 || (flg & 2) || (bd & 0x8F) || bsid < 4)
		return EXTENT_INVALID; /* version, reserved bits, block size id (what LZ4F_decompress rejects first) */
	if (n < hp)
		return 0;
	blkmax = 1ull << (8 + 2 * bsid); /* 4 -> 64 KiB ... 7 -> 4 MiB */
	*supported = 1; /* block checksums are verified and a dictionary id skipped by the frame-serial kernel */
	for (;;) {
		uint32_t bh, bsz;
		if (n - hp < 4)
			return 0;
		bh = rd32(p + hp);
		hp += 4;
		if (bh == 0)
			break;
		bsz = bh & 0x7FFFFFFFu;
		if (bsz > blkmax)
			return EXTENT_INVALID;
		if (n - hp < bsz + (bchk ? 4u : 0u))
			return 0;
		hp += bsz + (bchk ? 4u : 0u);
		sum += (bh & 0x80000000u) ? bsz : blkmax;
	}
	if (cchk) {
		if (n - hp < 4)
			return 0;
		hp += 4;
	}
	*bound = has_csize ? rd64(p + 6) : sum;
	return hp;
}

static size_t plain_decompress(LZ4MT_DCtx *ctx, LZ4MT_RdWr_t *io, const uint8_t *first)
{
	const size_t chunk = (size_t)ctx->inputsize;
	const size_t piece = chunk < 65536 ? 65536 : chunk;
	uint8_t *raw = (uint8_t *)malloc(chunk + 4);
	size_t cap = chunk + 4, n = 4, err = 0, ip = 0;
	size_t want_ahead = BATCH_BYTES; /* input buffered before a round of frames is split off */
	int eof = 0;
	struct dslot *s = &ctx->s[0];
	gpumt_ctx *g = mt_gpu_of(&ctx->gpus, (int)(s - ctx->s));
	if (!raw)
		return ERROR(memory_allocation);
	memcpy(raw, first, 4);
	ctx->insize = 4;
	ctx->outsize = 0;
	/* The input is consumed incrementally: read (in inputsize requests, :462-476) until about one
	 * batch of input is buffered or the stream ends, decode the complete frames of what is there, keep
	 * the incomplete tail, repeat -- the host holds about two batches of input plus the largest
	 * frame, not the whole stream, and output starts before the input ends. */
	for (;;) {
	int need_more = 0;
	while (!eof && n - ip < want_ahead) {
		LZ4MT_Buffer b;
		int rv;
		if (ip && ip == n) {
			n = 0;
			ip = 0;
		}
		if (n + chunk > cap) {
			if (ip >= chunk) { /* drop what is decoded instead of growing */
				memmove(raw, raw + ip, n - ip);
				n -= ip;
				ip = 0;
			} else {
				uint8_t *nr;
				cap = cap * 2 + chunk;
				nr = (uint8_t *)realloc(raw, cap);
				if (!nr) {
					free(raw);
					return ERROR(memory_allocation);
				}
				raw = nr;
			}
		}
		b.buf = raw + n;
		b.size = chunk;
		b.allocated = chunk;
		rv = io->fn_read(io->arg_read, &b);
		if (rv != 0) {
			free(raw);
			return mt_error(rv);
		}
		if (b.size == 0) {
			eof = 1;
			break;
		}
		n += b.size;
		ctx->insize += b.size;
	}
	while (ip < n && !err) {
		size_t in_bytes = 0, out_bytes = 0, nrec = 0, jp = ip;
		if (dbuf_want(g, &s->meta, D_META_BYTES(BATCH_MAXREC), 1, 1)) {
			err = ERROR(memory_allocation);
			break;
		}
		while (jp < n && nrec < BATCH_MAXREC) {
			uint64_t bound = 0;
			int supported = 0;
			size_t flen;
			if (n - jp >= 8 && (rd32(raw + jp) & 0xFFFFFFF0u) == LZ4FMT_MAGIC_SKIPPABLE) {
				const size_t sk = 8 + (size_t)rd32(raw + jp + 4);
				if (sk > n - jp) {
					if (!eof)
						need_more = 1; /* the rest of it has not been read yet */
					else
						err = ERROR(compression_library);
					break;
				}
				jp += sk;
				continue;
			}
			if (!eof && n - jp <= 0xFFFFFFF0u &&
			    (n - jp < 8 || (rd32(raw + jp) == LZ4FMT_MAGICNUMBER &&
					    !lz4_frame_extent(raw + jp, n - jp, &bound, &supported)))) {
				need_more = 1; /* an incomplete frame: wait for the rest (a damaged one is EXTENT_INVALID, below) */
				break;
			}
			if (n - jp < 4 || rd32(raw + jp) != LZ4FMT_MAGICNUMBER ||
			    !(flen = lz4_frame_extent(raw + jp, n - jp, &bound, &supported)) || flen == EXTENT_INVALID ||
			    !supported || flen > 0xFFFFFFF0u || bound > 0x7FFFFFFFull) {
				err = ERROR(compression_library);
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
		if (!nrec) {
			ip = jp;
			if (need_more)
				break;
			continue;
		}
		m_out_off(s, 0)[nrec] = out_bytes;
		if (dbuf_want(g, &s->in, in_bytes + 512, 1, 1) || dbuf_want(g, &s->out, out_bytes + 64, 1, 1) ||
		    dbuf_want(g, &s->status, nrec * 4 + 64, 1, 1)) {
			err = ERROR(memory_allocation);
			break;
		}
		{
			size_t k = 0, q = ip;
			while (k < nrec) {
				if ((rd32(raw + q) & 0xFFFFFFF0u) == LZ4FMT_MAGIC_SKIPPABLE) {
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
		{
			/* frame-serial kernel for every record: block sizes and counts are arbitrary here */
			const int prev = gpumt_set_variant(g, "lz4_dec", 1);
			int rc = 0;
			rc |= gpumt_memcpy_h2d(g, s->in.d, s->in.h, in_bytes, 0);
			rc |= gpumt_memcpy_h2d(g, s->meta.d, s->meta.h, D_META_BYTES(BATCH_MAXREC), 0);
			rc |= gpumt_lz4_decompress_batch(g, s->in.d, in_bytes, m_rec_off(s, 1), m_rec_len(s, 1), nrec, s->out.d,
							 out_bytes, m_out_off(s, 1), m_out_len(s, 1), (uint32_t *)s->status.d, 0);
			rc |= gpumt_memcpy_d2h(g, s->status.h, s->status.d, nrec * 4, 0);
			rc |= gpumt_memcpy_d2h(g, m_out_len(s, 0), m_out_len(s, 1), nrec * 4, 0);
			if (out_bytes)
				rc |= gpumt_memcpy_d2h(g, s->out.h, s->out.d, out_bytes, 0);
			rc |= gpumt_stream_sync(g, 0);
			gpumt_set_variant(g, "lz4_dec", prev);
			if (rc) {
				err = ERROR(compression_library);
				break;
			}
		}
		for (size_t i = 0; i < nrec && !err; i++) {
			const uint8_t *o = (const uint8_t *)s->out.h + m_out_off(s, 0)[i];
			size_t left = m_out_len(s, 0)[i];
			if (((const uint32_t *)s->status.h)[i] != GPUMT_ST_OK) {
				lz4mt_errcode = ((const uint32_t *)s->status.h)[i];
				err = ERROR(compression_library);
				break;
			}
			while (left && !err) {
				LZ4MT_Buffer b;
				const size_t k = left < piece ? left : piece;
				int rv;
				b.buf = (void *)o;
				b.size = k;
				b.allocated = k;
				rv = io->fn_write(io->arg_write, &b);
				if (rv != 0)
					err = mt_error(rv);
				ctx->outsize += k;
				o += k;
				left -= k;
			}
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

size_t LZ4MT_decompressDCtx(LZ4MT_DCtx *ctx, LZ4MT_RdWr_t *rdwr)
{
	static const mt_pipe_ops ops = {dp_fill, dp_launch, dp_complete, dp_drain, dp_role_start};
	uint8_t magic[4];
	LZ4MT_Buffer b;
	size_t err;
	int rv;

	if (!ctx)
		return ERROR(compressionParameter_unsupported); /* lz4-mt_decompress.c:493-494 */
	/* sniff: 4 bytes (lz4-mt_decompress.c:503-520), on the calling thread */
	b.buf = magic;
	b.size = 4;
	b.allocated = 4;
	rv = rdwr->fn_read(rdwr->arg_read, &b);
	if (rv != 0)
		return mt_error(rv);
	if (b.size != 4)
		return ERROR(data_error);
	if (rd32(magic) != LZ4FMT_MAGIC_SKIPPABLE) {
		if (rd32(magic) != LZ4FMT_MAGICNUMBER)
			return ERROR(data_error);
		/* plain .lz4 stream: the reference decodes it single-threaded (st_decompress, :391-483) */
		return plain_decompress(ctx, rdwr, magic);
	}
	ctx->io = rdwr;
	ctx->have_hdr = 0;
	ctx->first_hdr = 1;
	ctx->budget = BATCH_MIN;
	ctx->big_out = 0;
	/* threads == 1: every callback on the calling thread, as the reference (its single-thread path) */
	err = ctx->threads == 1 ? mt_pipe_run_inline(&ops, ctx) : mt_pipe_run_n(&ops, ctx, mt_nslot_for(ctx->gpus.n));
	mt_gpus_sync(&ctx->gpus);
	return err;
}
