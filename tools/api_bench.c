/*
 * api_bench.c -- end-to-end (PCIe-inclusive) throughput of the drop-in APIs of libzstdmt_amd.so:
 * LZ4MT_*, ZSTDCB_*, BROTLIMT_* or SNAPPYMT_* compressCCtx / decompressDCtx with in-memory callbacks, i.e. the same
 * measurement oracle/cpu_bench.c makes for the reference libraries.  Developer tool (numbers quoted
 * in DESIGN.md).  The two APIs have identical shapes, so they are bound by name at run time.
 *   api_bench <lz4|zstd|brotli|snappy> <bytes> <chunk>
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int zmt_gen_text(uint8_t *dst, size_t n, uint64_t seed, uint64_t offset, int threads);

typedef struct { void *buf; size_t size, allocated; } Buf;
typedef int (cb_fn)(void *, Buf *);
typedef struct { cb_fn *fn_read; void *arg_read; cb_fn *fn_write; void *arg_write; } RdWr;

struct mem { uint8_t *p; size_t n, pos; };
static int rd(void *a, Buf *b)
{
	struct mem *m = a;
	size_t k = m->n - m->pos < b->size ? m->n - m->pos : b->size;
	memcpy(b->buf, m->p + m->pos, k);
	m->pos += k;
	b->size = k;
	return 0;
}
static int wr(void *a, Buf *b)
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
/* The callbacks alone: the calls the engine makes (one fn_read of `chunk` bytes per chunk into a 256 MiB batch buffer that is reused
 * every 256 MiB, as the engine's slots are; one fn_write per record out of such a buffer), with no engine behind them -- the rate a
 * caller with these callbacks can be served at by ANY library that keeps the contract of one fn_read / one fn_write at a time
 * (/root/reference/lib/lz4-mt_compress.c:256-277).  ZMT_API_BOUND=1 prints it next to the legs */
static double g_alone_rd, g_alone_wr; /* MB/s of the callbacks alone (ZMT_API_BOUND=1), 0 = not measured */
static void callbacks_alone(const uint8_t *src, size_t n, size_t chunk, uint8_t *sink, size_t n_out)
{
	const size_t batch = (size_t)256 << 20;
	uint8_t *slot = malloc(batch);
	memset(slot, 1, batch);
	for (int rep = 0; rep < 2; rep++) {
		struct mem in = { (uint8_t *)src, n, 0 }, out = { sink, n_out, 0 };
		double t0 = now();
		for (size_t off = 0; in.pos < n;) {
			Buf b = { slot + off, chunk, chunk };
			rd(&in, &b);
			off = off + chunk + chunk > batch ? 0 : off + chunk;
		}
		const double tr = now() - t0;
		t0 = now();
		const size_t rec = chunk / 2 + 12; /* a record of a text chunk */
		for (size_t off = 0; out.pos + rec <= n_out;) {
			Buf b = { slot + off, rec, rec };
			wr(&out, &b);
			off = off + 2 * rec > batch ? 0 : off + rec;
		}
		const double tw = now() - t0;
		if (rep) {
			g_alone_rd = n / 1e6 / tr;
			g_alone_wr = out.pos / 1e6 / tw;
		}
		if (rep)
			fprintf(stderr, "callbacks alone: fn_read of %zu bytes x %zu calls %.1f MB/s; fn_write of %zu bytes x %zu calls %.1f MB/s\n",
				chunk, n / chunk, n / 1e6 / tr, rec, n_out / rec, out.pos / 1e6 / tw);
	}
	free(slot);
}
static void *sym(void *so, const char *pfx, const char *name)
{
	char buf[64];
	snprintf(buf, sizeof buf, "%s%s", pfx, name);
	void *p = dlsym(so, buf);
	if (!p) {
		fprintf(stderr, "missing %s\n", buf);
		exit(9);
	}
	return p;
}
int main(int argc, char **argv)
{
	const char *codec = argc > 1 ? argv[1] : "lz4";
	const char *pfx = !strcmp(codec, "zstd") ? "ZSTDCB_" : !strcmp(codec, "brotli") ? "BROTLIMT_" : !strcmp(codec, "snappy") ? "SNAPPYMT_" : "LZ4MT_";
	size_t n = argc > 2 ? strtoull(argv[2], 0, 10) : (size_t)1 << 30;
	int chunk = argc > 3 ? atoi(argv[3]) : 0;
	void *so = dlopen(argc > 4 ? argv[4] : "libzstdmt_amd.so", RTLD_NOW);
	const int level = argc > 5 ? atoi(argv[5]) : 1;
	if (!so) {
		fprintf(stderr, "dlopen: %s\n", dlerror());
		return 8;
	}
	void *(*createC)(int, int, int) = sym(so, pfx, "createCCtx");
	size_t (*compressC)(void *, RdWr *) = sym(so, pfx, "compressCCtx");
	void (*freeC)(void *) = sym(so, pfx, "freeCCtx");
	void *(*createD)(int, int) = sym(so, pfx, "createDCtx");
	size_t (*decompressD)(void *, RdWr *) = sym(so, pfx, "decompressDCtx");
	void (*freeD)(void *) = sym(so, pfx, "freeDCtx");
	unsigned (*isErr)(size_t) = sym(so, pfx, "isError");
	const char *(*errStr)(size_t) = sym(so, pfx, "getErrorString");
	size_t cap = n + n / 64 + (n / (chunk ? chunk : 131072) + 2) * 64 + 4096;
	uint8_t *src = malloc(n), *cmp = malloc(cap), *back = malloc(n);
	zmt_gen_text(src, n, 20260926, 0, 32);
	memset(cmp, 0, cap);
	memset(back, 0, n);
	if (getenv("ZMT_API_BOUND"))
		callbacks_alone(src, n, chunk ? (size_t)chunk : 131072, cmp, n / 2);
	for (int rep = 0; rep < 2; rep++) { /* rep 0 warms up (allocations, first touch) */
		struct mem in = { src, n, 0 }, out = { cmp, cap, 0 };
		RdWr io = { rd, &in, wr, &out };
		double t0 = now();
		void *c = createC(4, level, chunk);
		if (!c) { fprintf(stderr, "no device\n"); return 2; }
		double t_create = now() - t0;
		t0 = now();
		size_t rv = compressC(c, &io);
		double tc = now() - t0;
		t0 = now();
		freeC(c);
		double t_free = now() - t0;
		if (rep)
			fprintf(stderr, "compress: create %.3f s, run %.3f s, free %.3f s\n", t_create, tc, t_free);
		if (isErr(rv)) { fprintf(stderr, "compress: %s\n", errStr(rv)); return 3; }
		struct mem in2 = { cmp, out.pos, 0 }, out2 = { back, n, 0 };
		RdWr io2 = { rd, &in2, wr, &out2 };
		void *d = createD(4, 0);
		t0 = now();
		rv = decompressD(d, &io2);
		double td = now() - t0;
		freeD(d);
		if (isErr(rv)) { fprintf(stderr, "decompress: %s\n", errStr(rv)); return 4; }
		if (out2.pos != n || memcmp(src, back, n)) { fprintf(stderr, "round trip mismatch\n"); return 5; }
		if (rep) {
			printf("{\"api\": \"%s*\", \"level\": %d, \"bytes\": %zu, \"chunk\": %d, \"compressed\": %zu, \"compress_MBps\": %.1f, "
			       "\"decompress_MBps\": %.1f", pfx, level, n, chunk, out.pos, n / 1e6 / tc, n / 1e6 / td);
			if (g_alone_rd > 0)
				printf(", \"callbacks_alone_MBps\": {\"fn_read\": %.1f, \"fn_write\": %.1f, \"what\": \"the same callbacks driven with "
				       "the engine's call pattern and no engine behind them: the bound of any library under the one-fn_read-at-a-time contract\"}",
				       g_alone_rd, g_alone_wr);
			printf("}\n");
		}
	}
	return 0;
}
