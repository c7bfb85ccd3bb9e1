/*
 * cpu_bench.c -- CPU baseline timer for bench.py's "cpu_baseline" leg.  TEST/BENCH INFRASTRUCTURE.
 *
 * Times the lz4-mt path on the host cores over an in-memory synthetic buffer, through the very
 * API the reference CLI uses (programs/main.c:228-245 / :276-293): LZ4MT_createCCtx ->
 * LZ4MT_compressCCtx with fn_read/fn_write callbacks (here memcpy, so disk is excluded) and the
 * DCtx mirror.  Two kinds:
 *   reference-zstd / reference-brotli : the same for oracle/_ref/lib{zstd,brotli}mt_ref.so
 *   reference : oracle/_ref/liblz4mt_ref.so = the reference's own lib/lz4-mt_*.c + liblz4 (dlopen)
 *   port      : the oracle restatement (zo_lz4mt_*_mt), T threads
 * Prints one JSON object.
 *
 *   cpu_bench <kind> <ref.so|-> <bytes> <chunk> <threads> <seed>
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zmt_oracle.h"

int zmt_gen_text(uint8_t *dst, size_t n, uint64_t seed, uint64_t offset, int threads);

typedef struct {
	void *buf;
	size_t size, allocated;
} Buf; /* LZ4MT_Buffer, lib/lz4-mt.h:67-71 */
typedef int (*rw_fn)(void *, Buf *);
typedef struct {
	rw_fn fn_read;
	void *arg_read;
	rw_fn fn_write;
	void *arg_write;
} RdWr; /* LZ4MT_RdWr_t, lib/lz4-mt.h:84-89 */

struct mem {
	uint8_t *p;
	size_t n, pos;
};
static int rd(void *a, Buf *b)
{
	struct mem *m = (struct mem *)a;
	size_t k = m->n - m->pos < b->size ? m->n - m->pos : b->size;
	memcpy(b->buf, m->p + m->pos, k);
	m->pos += k;
	b->size = k;
	return 0;
}
static int wr(void *a, Buf *b)
{
	struct mem *m = (struct mem *)a;
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
	if (argc < 7) {
		fprintf(stderr, "usage: cpu_bench reference|reference-zstd|reference-brotli|port ref.so bytes chunk threads seed [level]\n");
		return 2;
	}
	const char *kind = argv[1];
	size_t n = strtoull(argv[3], 0, 10), chunk = strtoull(argv[4], 0, 10);
	int T = atoi(argv[5]);
	uint64_t seed = strtoull(argv[6], 0, 10);
	const int level = argc > 7 ? atoi(argv[7]) : 1; /* reference libraries only */
	size_t cap = zo_lz4mt_compress_bound(n, chunk) + n / 128 + 4096; /* also covers ZSTD_compressBound */
	uint8_t *src = malloc(n + 64), *cmp = malloc(cap), *back = malloc(n + 64);
	double tc, td;
	size_t csz = 0, dsz = 0;

	if (!src || !cmp || !back)
		return 3;
	zmt_gen_text(src, n, seed, 0, T);
	memset(cmp, 0, cap); /* touch pages outside the timed region */
	memset(back, 0, n);

	if (!strcmp(kind, "reference") || !strcmp(kind, "reference-zstd") || !strcmp(kind, "reference-brotli")) {
		/* the reference library itself: lz4-mt (LZ4MT_*) or zstd-mt (ZSTDCB_*, same shapes,
		 * lib/zstd-mt.h:115-205), level 1 */
		const int z = !strcmp(kind, "reference-zstd") ? 1 : !strcmp(kind, "reference-brotli") ? 2 : 0;
		void *so = dlopen(argv[2], RTLD_NOW);
		if (!so) {
			fprintf(stderr, "dlopen %s: %s\n", argv[2], dlerror());
			return 4;
		}
		void *(*createC)(int, int, int) = dlsym(so, z == 2 ? "BROTLIMT_createCCtx" : z ? "ZSTDCB_createCCtx" : "LZ4MT_createCCtx");
		size_t (*compressC)(void *, RdWr *) = dlsym(so, z == 2 ? "BROTLIMT_compressCCtx" : z ? "ZSTDCB_compressCCtx" : "LZ4MT_compressCCtx");
		void (*freeC)(void *) = dlsym(so, z == 2 ? "BROTLIMT_freeCCtx" : z ? "ZSTDCB_freeCCtx" : "LZ4MT_freeCCtx");
		void *(*createD)(int, int) = dlsym(so, z == 2 ? "BROTLIMT_createDCtx" : z ? "ZSTDCB_createDCtx" : "LZ4MT_createDCtx");
		size_t (*decompressD)(void *, RdWr *) = dlsym(so, z == 2 ? "BROTLIMT_decompressDCtx" : z ? "ZSTDCB_decompressDCtx" : "LZ4MT_decompressDCtx");
		void (*freeD)(void *) = dlsym(so, z == 2 ? "BROTLIMT_freeDCtx" : z ? "ZSTDCB_freeDCtx" : "LZ4MT_freeDCtx");
		unsigned (*isErr)(size_t) = dlsym(so, z == 2 ? "BROTLIMT_isError" : z ? "ZSTDCB_isError" : "LZ4MT_isError");
		struct mem in = { src, n, 0 }, out = { cmp, cap, 0 };
		RdWr io = { rd, &in, wr, &out };
		void *c = createC(T, level, (int)chunk);
		double t0 = now();
		size_t rv = compressC(c, &io);
		tc = now() - t0;
		freeC(c);
		if (isErr(rv))
			return 5;
		csz = out.pos;
		struct mem in2 = { cmp, csz, 0 }, out2 = { back, n, 0 };
		RdWr io2 = { rd, &in2, wr, &out2 };
		void *d = createD(T, 0);
		t0 = now();
		rv = decompressD(d, &io2);
		td = now() - t0;
		freeD(d);
		if (isErr(rv))
			return 6;
		dsz = out2.pos;
	} else {
		double t0 = now();
		csz = zo_lz4mt_compress_mt(src, n, chunk, cmp, cap, T);
		tc = now() - t0;
		t0 = now();
		dsz = zo_lz4mt_decompress_mt(cmp, csz, back, n, T);
		td = now() - t0;
	}
	if (dsz != n || memcmp(src, back, n))
		return 7;
	printf("{\"kind\": \"%s\", \"level\": %d, \"bytes\": %zu, \"chunk\": %zu, \"threads\": %d, \"compressed\": %zu, "
	       "\"compress_s\": %.6f, \"decompress_s\": %.6f, \"compress_MBps\": %.1f, "
	       "\"decompress_MBps\": %.1f, \"roundtrip_MBps\": %.1f}\n",
	       kind, level, n, chunk, T, csz, tc, td, n / 1e6 / tc, n / 1e6 / td, n / 1e6 / (tc + td));
	return 0;
}
