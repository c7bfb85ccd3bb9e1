/*
 * tsan_harness.c -- TEST HARNESS ONLY: the 16-byte-header host engine (mt16_engine.inc as snappymt_engine.c)
 * and the batch pipeline (mt_pipe.c) under ThreadSanitizer.  The device boundary is a plain-C stand-in
 * (stored-literal snappy streams out, the oracle's snappy decoder in), synchronous, so what is checked is
 * the engine's own threading: reader, caller and writer threads over the slots, counters and callbacks.
 * (The fiber emulator cannot run under TSan: it switches stacks behind its back.)
 * The lz4-mt engine (lz4mt_engine.c) runs the same way over the oracle's LZ4 frame coder.
 *   tsan_harness <bytes> <chunk> <threads> <slots> [snappy|lz4]   -> "rv_c rv_d frames ok"
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpumt.h"
#include "snappy-mt.h"
#include "lz4-mt.h"
#include "../../oracle/zmt_oracle.h"

struct gpumt_ctx { int device; };

int gpumt_device_count(void) { return 2; }
int gpumt_host_node(gpumt_ctx *h) { (void)h; return -1; }
int gpumt_open(int device, gpumt_ctx **out)
{
	gpumt_ctx *h = calloc(1, sizeof *h);
	if (!h || device >= 2)
		return GPUMT_E_HIP;
	h->device = device < 0 ? 0 : device;
	*out = h;
	return GPUMT_OK;
}
void gpumt_close(gpumt_ctx *h) { free(h); }
void *gpumt_malloc(gpumt_ctx *h, size_t n) { (void)h; return malloc(n + 512); }
void gpumt_free(gpumt_ctx *h, void *p) { (void)h; free(p); }
void *gpumt_host_alloc(gpumt_ctx *h, size_t n) { (void)h; return malloc(n + 512); }
void gpumt_host_free(gpumt_ctx *h, void *p) { (void)h; free(p); }
int gpumt_memcpy_h2d(gpumt_ctx *h, void *d, const void *s, size_t n, int st) { (void)h; (void)st; memmove(d, s, n); return 0; }
int gpumt_memcpy_d2h(gpumt_ctx *h, void *d, const void *s, size_t n, int st) { (void)h; (void)st; memmove(d, s, n); return 0; }
int gpumt_push_host(gpumt_ctx *h, void *d, const void *s, size_t n, const uint64_t *dn, int st)
{
	(void)h; (void)st;
	if (dn && *dn < n)
		n = (size_t)*dn;
	memcpy(d, s, n);
	return 0;
}
int gpumt_stream_wait(gpumt_ctx *h, int a, int b) { (void)h; (void)a; (void)b; return 0; }
int gpumt_stream_sync(gpumt_ctx *h, int s) { (void)h; (void)s; return 0; }
int gpumt_device_sync(gpumt_ctx *h) { (void)h; return 0; }
int gpumt_mark(gpumt_ctx *h, int id, int s) { (void)h; (void)id; (void)s; return 0; }
int gpumt_mark_sync(gpumt_ctx *h, int id) { (void)h; (void)id; return 0; }
size_t gpumt_lz4_record_count(size_t n, size_t chunk) { return n ? (n + chunk - 1) / chunk : 1; }
size_t gpumt_snappy_slot_stride(size_t chunk) { return (16 + 32 + chunk + chunk / 6 + 255) & ~(size_t)255; }

/* stored-literal snappy: preamble + literals of at most 65536 bytes, behind the 16-byte record header */
int gpumt_snappy_compress_batch(gpumt_ctx *h, const void *in, size_t n, size_t chunk, void *slots, size_t stride,
				uint32_t *rec_len, int st)
{
	(void)h; (void)st;
	const size_t nrec = gpumt_lz4_record_count(n, chunk);
	for (size_t r = 0; r < nrec; r++) {
		const uint8_t *src = (const uint8_t *)in + r * chunk;
		const size_t clen = n - r * chunk < chunk ? n - r * chunk : chunk;
		uint8_t *o = (uint8_t *)slots + r * stride;
		size_t op = 16;
		uint32_t v = (uint32_t)clen;
		while (v >= 128) {
			o[op++] = (uint8_t)(v | 128);
			v >>= 7;
		}
		o[op++] = (uint8_t)v;
		for (size_t at = 0; at < clen; at += 65536) {
			const size_t m = clen - at < 65536 ? clen - at : 65536;
			o[op++] = 61u << 2;
			o[op++] = (uint8_t)(m - 1);
			o[op++] = (uint8_t)((m - 1) >> 8);
			memcpy(o + op, src + at, m);
			op += m;
		}
		const uint32_t hint = clen < chunk ? (uint32_t)(clen >> 16) + 1 : (uint32_t)(chunk >> 16);
		const uint32_t hdr[3] = {0x184D2A50u, 8u, (uint32_t)(op - 16)};
		memcpy(o, hdr, 12);
		o[12] = 0x53; o[13] = 0x50; o[14] = (uint8_t)hint; o[15] = (uint8_t)(hint >> 8);
		rec_len[r] = (uint32_t)op;
	}
	return 0;
}

int gpumt_lz4_compact(gpumt_ctx *h, const void *slots, size_t stride, const uint32_t *rec_len, size_t nrec,
		      void *stream, uint64_t *rec_off, int st)
{
	(void)h; (void)st;
	uint64_t at = 0;
	for (size_t r = 0; r < nrec; r++) {
		rec_off[r] = at;
		memcpy((uint8_t *)stream + at, (const uint8_t *)slots + r * stride, rec_len[r]);
		at += rec_len[r];
	}
	rec_off[nrec] = at;
	return 0;
}

int gpumt_snappy_decompress_batch(gpumt_ctx *h, const void *stream, const uint64_t *rec_off, const uint32_t *rec_len,
				  size_t nrec, void *out, const uint64_t *out_off, const uint32_t *out_cap,
				  uint32_t *out_len, uint32_t *status, int st)
{
	(void)h; (void)st;
	for (size_t r = 0; r < nrec; r++) {
		const size_t got = zo_snappy_decompress((const uint8_t *)stream + rec_off[r], rec_len[r],
							(uint8_t *)out + out_off[r], out_cap[r]);
		status[r] = got == (size_t)-1 ? GPUMT_ST_BAD_BLOCK : GPUMT_ST_OK;
		out_len[r] = got == (size_t)-1 ? 0 : (uint32_t)got;
	}
	return 0;
}

/* lz4-mt: records = 12-byte skippable header + one LZ4 frame, coded by the oracle's frame functions */
size_t gpumt_lz4_slot_stride(size_t chunk) { return (12 + zo_lz4f_bound(chunk) + 255) & ~(size_t)255; }
int gpumt_lz4_level_supported(int level) { return level >= 1 && level <= 12; }
int gpumt_set_variant(gpumt_ctx *h, const char *what, int v) { (void)h; (void)what; (void)v; return 0; }
int gpumt_lz4_compress_batch_level(gpumt_ctx *h, const void *in, size_t n, size_t chunk, void *slots, size_t stride,
				   uint32_t *rec_len, int level, int st)
{
	(void)h; (void)st; (void)level;
	const size_t nrec = gpumt_lz4_record_count(n, chunk);
	for (size_t r = 0; r < nrec; r++) {
		const size_t clen = n - r * chunk < chunk ? n - r * chunk : chunk;
		uint8_t *o = (uint8_t *)slots + r * stride;
		const size_t f = zo_lz4f_compress((const uint8_t *)in + r * chunk, clen, o + 12, stride - 12);
		const uint32_t hdr[3] = {0x184D2A50u, 4u, (uint32_t)f};
		memcpy(o, hdr, 12);
		rec_len[r] = (uint32_t)(12 + f);
	}
	return 0;
}
int gpumt_lz4_decompress_batch(gpumt_ctx *h, const void *stream, size_t stream_bytes, const uint64_t *rec_off,
			       const uint32_t *rec_len, size_t nrec, void *out, size_t out_bytes, const uint64_t *out_off,
			       uint32_t *out_len, uint32_t *status, int st)
{
	(void)h; (void)st; (void)stream_bytes; (void)out_bytes;
	for (size_t r = 0; r < nrec; r++) {
		const size_t got = zo_lz4f_decompress((const uint8_t *)stream + rec_off[r] + 12, rec_len[r] - 12,
						      (uint8_t *)out + out_off[r], out_len[r]);
		status[r] = got == out_len[r] ? GPUMT_ST_OK : GPUMT_ST_BAD_BLOCK;
	}
	return 0;
}

struct mem { uint8_t *p; size_t n, pos; };
static int rd(void *a, SNAPPYMT_Buffer *b)
{
	struct mem *m = a;
	const size_t k = m->n - m->pos < b->size ? m->n - m->pos : b->size;
	memcpy(b->buf, m->p + m->pos, k);
	m->pos += k;
	b->size = k;
	return 0;
}
static int wr(void *a, SNAPPYMT_Buffer *b)
{
	struct mem *m = a;
	if (m->n - m->pos < b->size)
		return -1;
	memcpy(m->p + m->pos, b->buf, b->size);
	m->pos += b->size;
	return 0;
}

int main(int argc, char **argv)
{
	const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1 << 20;
	const int chunk = argc > 2 ? atoi(argv[2]) : 4096, threads = argc > 3 ? atoi(argv[3]) : 4;
	if (argc > 4)
		setenv("GPUMT_SLOTS", argv[4], 1);
	uint8_t *src = malloc(n + 1), *cmp = malloc(n + n / 8 + 65536), *back = malloc(n + 1);
	for (size_t i = 0; i < n; i++)
		src[i] = (uint8_t)(i * 2654435761u >> 13);
	struct mem in = {src, n, 0}, out = {cmp, n + n / 8 + 65536, 0};
	struct mem in2 = {cmp, 0, 0}, out2 = {back, n, 0};
	size_t rv_c, rv_d, frames;
	if (argc > 5 && !strcmp(argv[5], "lz4")) { /* LZ4MT_Buffer has the layout of SNAPPYMT_Buffer */
		LZ4MT_RdWr_t io = {(int (*)(void *, LZ4MT_Buffer *))rd, &in, (int (*)(void *, LZ4MT_Buffer *))wr, &out};
		LZ4MT_CCtx *c = LZ4MT_createCCtx(threads, 1, chunk);
		rv_c = LZ4MT_compressCCtx(c, &io);
		frames = LZ4MT_GetFramesCCtx(c);
		LZ4MT_freeCCtx(c);
		in2.n = out.pos;
		LZ4MT_RdWr_t io2 = {(int (*)(void *, LZ4MT_Buffer *))rd, &in2, (int (*)(void *, LZ4MT_Buffer *))wr, &out2};
		LZ4MT_DCtx *d = LZ4MT_createDCtx(threads, 0);
		rv_d = LZ4MT_decompressDCtx(d, &io2);
		LZ4MT_freeDCtx(d);
	} else {
		SNAPPYMT_RdWr_t io = {rd, &in, wr, &out};
		SNAPPYMT_CCtx *c = SNAPPYMT_createCCtx(threads, 0, chunk);
		rv_c = SNAPPYMT_compressCCtx(c, &io);
		frames = SNAPPYMT_GetFramesCCtx(c);
		SNAPPYMT_freeCCtx(c);
		in2.n = out.pos;
		SNAPPYMT_RdWr_t io2 = {rd, &in2, wr, &out2};
		SNAPPYMT_DCtx *d = SNAPPYMT_createDCtx(threads, 0);
		rv_d = SNAPPYMT_decompressDCtx(d, &io2);
		SNAPPYMT_freeDCtx(d);
	}
	printf("%zd %zd %zu %d\n", (ssize_t)rv_c, (ssize_t)rv_d, frames, out2.pos == n && !memcmp(src, back, n));
	return 0;
}
