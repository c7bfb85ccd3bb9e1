/*
 * mt_host.h -- small helpers shared by the host engines (lz4mt_engine.c, zstdmt_engine.c):
 * little-endian reads and a device buffer + pinned mirror pair that only ever grows.
 * Plain C over include/gpumt.h; no HIP header.
 */
#ifndef ZMT_MT_HOST_H
#define ZMT_MT_HOST_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpumt.h"

/*
 * Device batches.  The kernels are latency-bound per chunk (a wave per chunk / per record), so a
 * batch should hold thousands of chunks to fill 256 CUs; pinned host memory on the other hand costs
 * ~0.3 s per GiB to allocate and free (measured, tools/ubench/hostalloc.hip).  Batches therefore
 * start at 64 MiB (small inputs stay cheap) and grow to 256 MiB; the buffers live as long as the
 * context, so repeated calls on one context do not pay for them again.
 */
#define BATCH_MIN (zmt_batch_bytes() < ((size_t)64 << 20) ? zmt_batch_bytes() : (size_t)64 << 20)
#define BATCH_BYTES (zmt_batch_bytes())
static inline size_t zmt_batch_bytes(void)
{
	static size_t v;
	if (!v) {
		const char *e = getenv("GPUMT_BATCH_MB"); /* developer knob: device batch size */
		const char *k = getenv("GPUMT_BATCH_KB"); /* test knob: batches of a few records (tests/emu) */
		size_t mb = e && *e ? (size_t)strtoull(e, 0, 10) : 256;
		v = (mb < 16 ? 16 : mb > 2048 ? 2048 : mb) << 20;
		if (k && *k) {
			size_t kb = (size_t)strtoull(k, 0, 10);
			v = (kb < 16 ? 16 : kb > (2048u << 10) ? (2048u << 10) : kb) << 10;
		}
	}
	return v;
}
#define BATCH_MAXREC 8192
/*
 * The kernels run one wave per chunk / record, and a wave moves ~20 MB/s whatever else runs: the device needs about a thousand
 * chunks in flight to keep up with the reader (measured, profiles/r06_sweeps/api_chunk_sizes.txt: LZ4MT at the reference's
 * default 4 MiB chunk -- lib/lz4-mt_compress.c:114 -- 3.9 GB/s with 3 x 64 chunks in flight, 9.8 with 3 x 256).  So unless the
 * batch size is set by hand a batch grows to hold 256 chunks of the size in use, up to 1 GiB.
 */
static inline size_t zmt_batch_bytes_for(size_t unit)
{
	const char *e = getenv("GPUMT_BATCH_MB"), *k = getenv("GPUMT_BATCH_KB");
	const size_t base = zmt_batch_bytes();
	size_t want = unit > ((size_t)1 << 22) ? (size_t)1 << 30 : unit * 256;
	if ((e && *e) || (k && *k))
		return base;
	if (want > ((size_t)1 << 30))
		want = (size_t)1 << 30;
	return want > base ? want : base;
}

static inline uint32_t rd32(const uint8_t *p)
{
	return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
static inline uint64_t rd64(const uint8_t *p)
{
	return (uint64_t)rd32(p) | (uint64_t)rd32(p + 4) << 32;
}

/*
 * The devices of a context.  By default one: the GPU GPUMT_DEVICE names (0 if unset).  With
 * GPUMT_DEVICES = "all" or a list ("0,1,2,3") a context opens several and deals its batch slots out
 * round-robin (slot s -> device s % n): every batch lives on one device from its H2D copy to its
 * result, batches of different devices run side by side, order is kept by the pipeline as before.
 * The callback contract bounds what this buys to the reader's / writer's memcpy rate (~15-20 GB/s),
 * which is why it matters for the device-bound legs: LZ4HC levels, brotli decode.
 */
#define MT_NGPU_MAX 16
typedef struct {
	gpumt_ctx *g[MT_NGPU_MAX];
	int n;
	int trace; /* GPUMT_TRACE, read once in mt_gpus_open (the pipeline threads only test the field) */
} mt_gpus;

int mt_bind_to_node(int node); /* mt_pipe.c */
/* the reader / writer thread of a context next to its pinned buffers: the host node of its devices when they share one */
static inline void mt_bind_near(const mt_gpus *m)
{
	int node = m->n > 0 ? gpumt_host_node(m->g[0]) : -1;
	for (int i = 1; i < m->n; i++)
		if (gpumt_host_node(m->g[i]) != node)
			node = -1;
	if (mt_bind_to_node(node) && m->trace)
		fprintf(stderr, "[mt_pipe] reader / writer thread bound to host node %d (the device's)\n", node);
}

static inline void mt_gpus_close(mt_gpus *m)
{
	for (int i = 0; i < m->n; i++)
		gpumt_close(m->g[i]);
	m->n = 0;
}
/* 0 on success; nothing is left open on failure */
static inline int mt_gpus_open(mt_gpus *m)
{
	const char *e = getenv("GPUMT_TRACE");
	m->trace = e && *e ? atoi(e) : 0;
	e = getenv("GPUMT_DEVICES");
	m->n = 0;
	if (!e || !*e) {
		if (gpumt_open(GPUMT_DEVICE_DEFAULT, &m->g[0]) != GPUMT_OK)
			return -1;
		m->n = 1;
		return 0;
	}
	if (!strcmp(e, "all")) {
		const int nd = gpumt_device_count();
		for (int d = 0; d < nd && m->n < MT_NGPU_MAX; d++) {
			if (gpumt_open(d, &m->g[m->n]) != GPUMT_OK) {
				mt_gpus_close(m);
				return -1;
			}
			m->n++;
		}
		return m->n ? 0 : -1;
	}
	while (*e && m->n < MT_NGPU_MAX) {
		char *end;
		const long d = strtol(e, &end, 10);
		if (end == e || d < 0 || gpumt_open((int)d, &m->g[m->n]) != GPUMT_OK) {
			mt_gpus_close(m);
			return -1;
		}
		m->n++;
		e = *end == ',' ? end + 1 : end;
		if (*end && *end != ',')
			break;
	}
	return m->n ? 0 : -1;
}
static inline gpumt_ctx *mt_gpu_of(const mt_gpus *m, int slot) { return m->g[slot % m->n]; }
/* GPUMT_TRACE >= 2: which device context a batch went to (the order tests/test_gpu_cli.py asserts: batch b uses slot
 * b % nslot, slot s the device context s % n -- the in-order writer of lib/lz4-mt_compress.c:178-205 re-expressed) */
static inline void mt_trace_launch(const mt_gpus *m, const char *who, int slot, size_t nrec)
{
	if (m->trace > 1)
		fprintf(stderr, "[%s] launch slot %d -> device context %d of %d, stream %d, %zu records\n", who, slot,
			slot % m->n, m->n, 4 + (slot / m->n) % 12, nrec);
}
/* the slot's own kernel stream (4..15) and completion mark on its device */
static inline int mt_stream_of(const mt_gpus *m, int slot) { return 4 + (slot / m->n) % 12; }
static inline int mt_mark_of(const mt_gpus *m, int slot) { return (slot / m->n) % GPUMT_NMARKS; }
static inline void mt_gpus_sync(const mt_gpus *m)
{
	for (int i = 0; i < m->n; i++)
		gpumt_device_sync(m->g[i]);
}

/* a device buffer + pinned mirror that only ever grows */
typedef struct {
	void *d;
	void *h;
	size_t cap;
} dbuf;

static inline int dbuf_want(gpumt_ctx *g, dbuf *b, size_t bytes, int pinned, int device)
{
	if (bytes <= b->cap)
		return 0;
	/* the caller owns an idle slot (its last batch is drained), so nothing on the device uses the old
	 * buffers: no device-wide wait here -- it would stall the batches of the other slots in flight */
	if (b->d)
		gpumt_free(g, b->d);
	if (b->h)
		gpumt_host_free(g, b->h);
	b->d = b->h = NULL;
	b->cap = 0;
	bytes += bytes / 8 + 4096;
	if (device && !(b->d = gpumt_malloc(g, bytes)))
		return -1;
	if (pinned && !(b->h = gpumt_host_alloc(g, bytes)))
		return -1;
	b->cap = bytes;
	return 0;
}
static inline void dbuf_free(gpumt_ctx *g, dbuf *b)
{
	if (b->d)
		gpumt_free(g, b->d);
	if (b->h)
		gpumt_host_free(g, b->h);
	memset(b, 0, sizeof *b);
}

/* =================================================================== compression ============ */

#endif
