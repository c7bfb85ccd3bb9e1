/*
 * api_bench.c -- end-to-end (PCIe-inclusive) throughput of the drop-in API: LZ4MT_compressCCtx /
 * LZ4MT_decompressDCtx of libzstdmt_amd.so with in-memory callbacks, i.e. the same measurement
 * oracle/cpu_bench.c makes for the reference library.  Developer tool (numbers quoted in DESIGN.md).
 *   api_bench <bytes> <chunk>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lz4-mt.h"

int zmt_gen_text(uint8_t *dst, size_t n, uint64_t seed, uint64_t offset, int threads);

struct mem { uint8_t *p; size_t n, pos; };
static int rd(void *a, LZ4MT_Buffer *b)
{
	struct mem *m = a;
	size_t k = m->n - m->pos < b->size ? m->n - m->pos : b->size;
	memcpy(b->buf, m->p + m->pos, k);
	m->pos += k;
	b->size = k;
	return 0;
}
static int wr(void *a, LZ4MT_Buffer *b)
{
	struct mem *m = a;
	if (m->n - m->pos < b->size)
		return -1;
	memcpy(m->p + m->pos, b->buf, b->size);
	m->pos += b->size;
	return 0;
}
static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
int main(int argc, char **argv)
{
	size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : (size_t)1 << 30;
	int chunk = argc > 2 ? atoi(argv[2]) : 131072;
	size_t cap = n + n / 64 + (n / chunk + 2) * 64 + 4096;
	uint8_t *src = malloc(n), *cmp = malloc(cap), *back = malloc(n);
	zmt_gen_text(src, n, 20260926, 0, 32);
	memset(cmp, 0, cap);
	memset(back, 0, n);
	for (int rep = 0; rep < 2; rep++) { /* rep 0 warms up (allocations, first touch) */
		struct mem in = { src, n, 0 }, out = { cmp, cap, 0 };
		LZ4MT_RdWr_t io = { rd, &in, wr, &out };
		double t0 = now();
		LZ4MT_CCtx *c = LZ4MT_createCCtx(4, 1, chunk);
		if (!c) { fprintf(stderr, "no device\n"); return 2; }
		double t_create = now() - t0;
		t0 = now();
		size_t rv = LZ4MT_compressCCtx(c, &io);
		double tc = now() - t0;
		t0 = now();
		LZ4MT_freeCCtx(c);
		double t_free = now() - t0;
		if (rep)
			fprintf(stderr, "compress: create %.3f s, run %.3f s, free %.3f s\n", t_create, tc, t_free);
		if (LZ4MT_isError(rv)) { fprintf(stderr, "compress: %s\n", LZ4MT_getErrorString(rv)); return 3; }
		struct mem in2 = { cmp, out.pos, 0 }, out2 = { back, n, 0 };
		LZ4MT_RdWr_t io2 = { rd, &in2, wr, &out2 };
		LZ4MT_DCtx *d = LZ4MT_createDCtx(4, 0);
		t0 = now();
		rv = LZ4MT_decompressDCtx(d, &io2);
		double td = now() - t0;
		LZ4MT_freeDCtx(d);
		if (LZ4MT_isError(rv)) { fprintf(stderr, "decompress: %s\n", LZ4MT_getErrorString(rv)); return 4; }
		if (out2.pos != n || memcmp(src, back, n)) { fprintf(stderr, "round trip mismatch\n"); return 5; }
		if (rep)
			printf("{\"api\": \"LZ4MT_*\", \"bytes\": %zu, \"chunk\": %d, \"compressed\": %zu, \"compress_MBps\": %.1f, "
			       "\"decompress_MBps\": %.1f}\n", n, chunk, out.pos, n / 1e6 / tc, n / 1e6 / td);
	}
	return 0;
}
