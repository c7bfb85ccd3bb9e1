/*
 * gpumt.hip -- C-ABI shim between the plain-C host engine and the gfx950 kernels (include/gpumt.h).
 * Owns the HIP device, four streams, timing events and per-handle scratch; launches the kernels.
 * There is deliberately no CPU fallback anywhere in this file.
 */
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/gpumt.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

extern "C" {
__global__ void zmt_xxh32_kernel(const u8 *, const u64 *, const u32 *, u32, u32 *, const u32 *,
				 const u32 *, u32 *);
__global__ void zmt_lz4_enc3_u16_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *,
					unsigned long long *);
__global__ void zmt_lz4_enc3_p17_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *,
					unsigned long long *);
__global__ void zmt_lz4_enc3_p17_prof_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *,
					     unsigned long long *);
__global__ void zmt_push_host_kernel(const u8 *, u8 *, u64, const u64 *);
__global__ void zmt_lz4hc_enc_kernel(const u8 *, u64, u32, u32, u8 *, u64, u32 *, const u32 *, u8 *, int);
__global__ void zmt_lz4_enc5_u16_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *,
					unsigned long long *);
__global__ void zmt_lz4_enc5_p17_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *,
					unsigned long long *);
__global__ void zmt_lz4_enc5_u32_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *,
					unsigned long long *);
__global__ void zmt_lz4_enc3_u32_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *,
					unsigned long long *);
__global__ void zmt_lz4_dec_serial(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *,
				   u32 *, u32 *, u32 *, u32 *, u32);
__global__ void zmt_dec_nblk_kernel(const u32 *, u32, u32 *);
__global__ void zmt_dec_frames_kernel(const u8 *, const u64 *, const u32 *, u32, const u32 *,
				      const u64 *, u64 *, u32 *, u32 *, u32 *, u32 *, u32 *, u32 *);
__global__ void zmt_dec_parse4_kernel(const u8 *, u64, const u64 *, const u32 *, const u64 *, u16 *, u32 *, u32 *);
#define C3_DECL(NAME)                                                                                              \
	__global__ void NAME(const u8 *, u64, u32, u32, u8 *, const u64 *, const u32 *, const u64 *, const u64 *,  \
			     const u32 *, const u32 *, const u32 *, const u16 *, const u32 *, const u32 *, u32 *);
C3_DECL(zmt_dec_copy3_w4_kernel)
C3_DECL(zmt_dec_copy3_w8_kernel)
C3_DECL(zmt_dec_copy3_w16_kernel)
#define C3_DECLP(NAME)                                                                                             \
	__global__ void NAME(const u8 *, u64, u32, u32, u8 *, const u64 *, const u32 *, const u64 *, const u64 *,  \
			     const u32 *, const u32 *, const u32 *, const u16 *, const u32 *, const u32 *, u32 *, \
			     unsigned long long *);
C3_DECLP(zmt_dec_copy3_w4_kernel_prof)
C3_DECLP(zmt_dec_copy3_w8_kernel_prof)
C3_DECLP(zmt_dec_copy3_w16_kernel_prof)
__global__ void zmt_brotli_enc_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
__global__ void zmt_brotli_enc_t2_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
__global__ void zmt_brotli_enc_t3_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
__global__ void zmt_snappy_enc_kernel(const u8 *, u64, u32, u32, u8 *, u64, u32 *);
__global__ void zmt_snappy_dec_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *, u32 *, u32 *);
__global__ void zmt_snappy_dec2_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *, u32 *, u32 *);
__global__ void zmt_brotli_assemble_kernel(u64, u32, u32, u32, u8 *, u64, const u32 *, u32 *);
__global__ void zmt_brotli_dec_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *,
				      u32 *, u32 *, u8 *, const u8 *, u32);
__global__ void zmt_brotli_dec_kernel_prof(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *,
					   const u32 *, u32 *, u32 *, u8 *, const u8 *, u32, unsigned long long *);
#ifndef B4_NG
#define B4_NG 4 /* streams per wave of zmt_brotli_dec4_kernel (brotli_dec4.hip) */
#endif
__global__ void zmt_brotli_dec4_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *,
				       u32 *, u32 *);
__global__ void zmt_zstd_dec_small_kernel(const u8 *, u64, const u64 *, const u32 *, u32, u8 *, const u64 *,
					  u32 *, u8 *, u32 *, u32 *, u32 *, u8 *, u64);
__global__ void zmt_zstd_dec_kernel(const u8 *, u64, const u64 *, const u32 *, u32, u8 *, const u64 *,
				    u32 *, u8 *, u32 *, u32 *, u32 *, u32, u8 *, u64);
__global__ void zmt_zstd_dec_kernel_prof(const u8 *, u64, const u64 *, const u32 *, u32, u8 *, const u64 *,
					 u32 *, u8 *, u32 *, u32 *, u32 *, u32, unsigned long long *, u8 *, u64);
__global__ void zmt_zstd_seq_kernel(const u8 *, u64, const u64 *, const u32 *, u32, const u64 *, const u32 *,
				    const u32 *, u8 *, u64);
__global__ void zmt_xxh64_verify_kernel(const u8 *, const u64 *, const u32 *, u32, const u32 *, const u32 *,
					u32 *);
__global__ void zmt_zstd_enc_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
__global__ void zmt_zstd_enc_t2_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
__global__ void zmt_zstd_enc_t3_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
__global__ void zmt_zstd_enc_kernel_prof(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *, unsigned long long *);
__global__ void zmt_zstd_assemble_kernel(u64, u32, u32, u32, u8 *, u64, const u32 *, u32 *);
__global__ void zmt_zstd_probe_kernel(const u8 *, const u64 *, const u32 *, u32, u32 *, u32 *);
__global__ void zmt_probe_kernel(const u8 *, const u64 *, const u32 *, u32, u32 *);
__global__ void zmt_scan_kernel(const u32 *, u32, u64 *);
__global__ void zmt_compact_kernel(const u8 *, u64, const u32 *, const u64 *, u32, u8 *);
__global__ void zmt_iota_kernel(u64 *, u32 *, u64, u64, u32);
}

#define NTIMERS 16

struct gpumt_ctx {
	int device;
	hipStream_t st[GPUMT_NSTREAMS];
	hipEvent_t t0[NTIMERS], t1[NTIMERS];
	hipEvent_t xev;
	hipEvent_t mark[GPUMT_NMARKS];
	/* scratch, grown on demand (never shrinks) */
	void *scratch[2][GPUMT_NSTREAMS]; /* [0] compress side, [1] decompress side; one per launching stream, so
					   * batches launched on different streams can overlap */
	size_t scratch_bytes[2][GPUMT_NSTREAMS];
	int dec_variant;
	unsigned long long *d_prof; /* 16 phase counters of the profiling decoder */
	int xflags;  /* experiment switches of the parse kernel (developer) */
	int lz4_ring; /* copy3: log2 of the LDS ring per wave (12, 13 or 14) */
	int enc_variant; /* LZ4 fast levels: 0 = the window encoder (lz4_enc5.hip), 3 = the probe-batch encoder (lz4_enc3.hip) */
	int dec_pad;  /* copy stage: dynamic-LDS padding per workgroup = resident-wave cap (developer A/B, GPUMT_LZ4_DEC_PAD) */
	int debug_free; /* GPUMT_DEBUG_FREE: gpumt_free / gpumt_host_free check that the streams are idle */
	int profile; /* record events in timer slots 8.. around individual kernels */
	int num_cus;
	int zenc_waves[3]; /* resident waves of the persistent zstd encoder kernels (whole device), per level tier */
	int hc_waves;     /* developer: grid of the LZ4HC encoder (0 = GPUMT_LZ4HC_WAVES) */
	int zdec_variant; /* 0 = small-table kernel, then general; 1 = general only */
	int zseq_variant; /* 0 = sequence pre-pass (zstd_dec_seq.hip) in front of the frame decoder; 1 = none */
	int sdec_variant; /* snappy decoder: 0 = element by element, 1 = 64 elements per batch (snappy.hip) */
	int bdec_variant; /* brotli decoder: 0 = by batch size, 1 = the general kernel only, 2 = dec4 (four records per wave) + general for what it hands over */
	int bdec_waves;   /* resident waves of the persistent brotli decoder kernel (whole device) */
	int benc_waves[3]; /* resident waves of the persistent brotli encoder kernels (whole device), per quality tier */
	int senc_waves, sdec_waves, sdec2_waves; /* ... of the snappy kernels */
	void *d_brotli_static; /* device copy of the RFC 7932 constant data */
	char err[256];
	char name[128];
};

static int fail(gpumt_ctx *h, hipError_t e, const char *what)
{
	if (h)
		snprintf(h->err, sizeof h->err, "%s: %s", what, hipGetErrorString(e));
	return GPUMT_E_HIP;
}
#define CK(call)                                                                                   \
	do {                                                                                       \
		hipError_t e_ = (call);                                                            \
		if (e_ != hipSuccess)                                                              \
			return fail(h, e_, #call);                                                 \
	} while (0)

#define PROF0(slot)                                                                               \
	do {                                                                                       \
		if (h->profile)                                                                    \
			(void)hipEventRecord(h->t0[slot], h->st[s]);                               \
	} while (0)
#define PROF1(slot)                                                                               \
	do {                                                                                       \
		if (h->profile)                                                                    \
			(void)hipEventRecord(h->t1[slot], h->st[s]);                               \
	} while (0)

static int use(gpumt_ctx *h)
{
	CK(hipSetDevice(h->device));
	return GPUMT_OK;
}

static void *dev_alloc(gpumt_ctx *h, size_t bytes);
static void dev_free(gpumt_ctx *h, void *p);

static int want_scratch(gpumt_ctx *h, int k, int s, size_t bytes)
{
	if (bytes <= h->scratch_bytes[k][s])
		return GPUMT_OK;
	if (h->scratch[k][s]) {
		/* only work queued on this stream can be using the old area; a device-wide wait here would
		 * stall the batches other slots have in flight (130 ms per growing brotli batch, GPUMT_TRACE) */
		CK(hipStreamSynchronize(h->st[s]));
		dev_free(h, h->scratch[k][s]);
		h->scratch[k][s] = NULL;
		h->scratch_bytes[k][s] = 0;
	}
	bytes = (bytes + 0xFFFFF) & ~(size_t)0xFFFFF;
	h->scratch[k][s] = dev_alloc(h, bytes);
	if (!h->scratch[k][s]) {
		snprintf(h->err, sizeof h->err, "device scratch of %zu bytes: out of memory", bytes);
		return GPUMT_E_HIP;
	}
	h->scratch_bytes[k][s] = bytes;
	return GPUMT_OK;
}

extern "C" {

int gpumt_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

/* The host engines keep several batches in flight on streams of their own; with the runtime's default of 4
 * hardware queues those streams share queues and wait for each other's kernels (measured:
 * tools/ubench/pipe_overlap.hip).  Set once, when the library is loaded -- before any thread of the host can be
 * inside getenv / setenv on our account and before the first HIP call of a process that reaches HIP through
 * this library (the CLI, the drop-in APIs); a host that exports its own value, or initialised HIP earlier, keeps
 * what it has (overwrite = 0). */
__attribute__((constructor)) static void gpumt_library_init(void) { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

int gpumt_open(int device, gpumt_ctx **out)
{
	int n = 0;
	gpumt_ctx *h;
	if (!out)
		return GPUMT_E_ARG;
	*out = NULL;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return GPUMT_E_NODEVICE;
	if (device == GPUMT_DEVICE_DEFAULT) {
		/* the drop-in APIs (LZ4MT_* ...) have no device argument: GPUMT_DEVICE selects the GPU of
		 * the contexts a process creates (one process per GPU, like the ranks of bench.py) */
		const char *e = getenv("GPUMT_DEVICE");
		device = (e && *e) ? atoi(e) : 0;
	}
	if (device < 0 || device >= n)
		return GPUMT_E_NODEVICE;
	h = (gpumt_ctx *)calloc(1, sizeof *h);
	if (!h)
		return GPUMT_E_NOMEM;
	h->device = device;
	{
		hipDeviceProp_t p;
		hipError_t e = hipSetDevice(device);
		if (e == hipSuccess)
			e = hipGetDeviceProperties(&p, device);
		if (e != hipSuccess) {
			free(h);
			return GPUMT_E_NODEVICE;
		}
		snprintf(h->name, sizeof h->name, "%s (%s, %d CUs)", p.name, p.gcnArchName,
			 p.multiProcessorCount);
		h->num_cus = p.multiProcessorCount;
		if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
			/* kernels are built for gfx950 only */
			free(h);
			return GPUMT_E_NODEVICE;
		}
	}
	for (int i = 0; i < GPUMT_NSTREAMS; i++)
		if (hipStreamCreateWithFlags(&h->st[i], hipStreamNonBlocking) != hipSuccess) {
			free(h);
			return GPUMT_E_HIP;
		}
	for (int i = 0; i < NTIMERS; i++) {
		(void)hipEventCreate(&h->t0[i]);
		(void)hipEventCreate(&h->t1[i]);
	}
	(void)hipEventCreateWithFlags(&h->xev, hipEventDisableTiming);
	for (int i = 0; i < GPUMT_NMARKS; i++)
		(void)hipEventCreateWithFlags(&h->mark[i], hipEventDisableTiming);
	{
		/* developer knob for A/B runs through the drop-in API / the CLI, which have no handle to call
		 * gpumt_set_variant on: GPUMT_SNAPPY_DEC=1 selects the batched snappy decoder */
		const char *e = getenv("GPUMT_SNAPPY_DEC");
		h->sdec_variant = e && *e ? atoi(e) : 0;
		/* GPUMT_LZ4_DEC: 0 = frames + parse4 + copy3 (default), 1 = frame-serial;
		 * GPUMT_LZ4_RING: log2 of copy3's LDS ring per wave (12 = 4 KiB, the default; 13; 14) */
		e = getenv("GPUMT_LZ4_DEC");
		h->dec_variant = e && *e ? atoi(e) : 0;
		e = getenv("GPUMT_LZ4_RING");
		h->lz4_ring = e && *e ? atoi(e) : 12;
		e = getenv("GPUMT_LZ4_ENC");
		h->enc_variant = e && *e ? atoi(e) : 0;
		e = getenv("GPUMT_LZ4_DEC_PAD");
		h->dec_pad = e && *e ? atoi(e) : 0;
		/* GPUMT_ZSTD_SEQ=1: no sequence pre-pass in front of the zstd frame decoder */
		e = getenv("GPUMT_ZSTD_SEQ");
		h->zseq_variant = e && *e ? atoi(e) : 0;
		/* GPUMT_BROTLI_DEC: 0 = the batch size chooses (default), 1 = the general kernel, 2 = dec4 first */
		e = getenv("GPUMT_BROTLI_DEC");
		h->bdec_variant = e && *e ? atoi(e) : 0;
		/* GPUMT_DEBUG_FREE=1: the guard of the buffer contract (include/gpumt.h) for callers without a handle */
		e = getenv("GPUMT_DEBUG_FREE");
		h->debug_free = e && *e ? atoi(e) : 0;
		if (h->debug_free)
			fprintf(stderr, "gpumt: free guard on for device %d (GPUMT_DEBUG_FREE)\n", device);
	}
	*out = h;
	return GPUMT_OK;
}

void gpumt_close(gpumt_ctx *h)
{
	if (!h)
		return;
	(void)hipSetDevice(h->device);
	(void)hipDeviceSynchronize();
	for (int k = 0; k < 2; k++)
		for (int i = 0; i < GPUMT_NSTREAMS; i++)
			if (h->scratch[k][i])
				dev_free(h, h->scratch[k][i]);
	for (int i = 0; i < NTIMERS; i++) {
		(void)hipEventDestroy(h->t0[i]);
		(void)hipEventDestroy(h->t1[i]);
	}
	(void)hipEventDestroy(h->xev);
	for (int i = 0; i < GPUMT_NMARKS; i++)
		(void)hipEventDestroy(h->mark[i]);
	for (int i = 0; i < GPUMT_NSTREAMS; i++)
		(void)hipStreamDestroy(h->st[i]);
	free(h);
}

const char *gpumt_last_error(gpumt_ctx *h) { return h ? h->err : "no handle"; }
const char *gpumt_device_name(gpumt_ctx *h) { return h ? h->name : ""; }

int gpumt_host_node(gpumt_ctx *h)
{
	char bus[64] = "", path[128];
	int node = -1;
	if (!h || hipDeviceGetPCIBusId(bus, (int)sizeof bus, h->device) != hipSuccess)
		return -1;
	for (char *c = bus; *c; c++)
		if (*c >= 'A' && *c <= 'F')
			*c = (char)(*c - 'A' + 'a'); /* sysfs spells the address in lower case */
	snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
	FILE *f = fopen(path, "r");
	if (!f)
		return -1;
	if (fscanf(f, "%d", &node) != 1)
		node = -1;
	fclose(f);
	return node;
}

/*
 * Device buffers of a context (batch slots, per-wave scratch) are GiB-sized and hipMalloc / hipFree of
 * that size cost tens of milliseconds each -- 0.86 s of a 1.3 s BROTLIMT_decompressDCtx call on 8 GiB
 * were first-use allocations (GPUMT_TRACE).  Like the pinned staging buffers below, freed device
 * buffers therefore stay in a process-wide cache (per device, bounded by GPUMT_DEVICE_CACHE_MB,
 * default 16384) for the next context of the process.
 */
#define DEV_CACHE_SLOTS 192
static struct {
	pthread_mutex_t mu;
	void *p[DEV_CACHE_SLOTS];
	size_t cap[DEV_CACHE_SLOTS];
	int dev[DEV_CACHE_SLOTS];
	int used[DEV_CACHE_SLOTS];
	size_t idle, limit;
	int init;
} g_dev = {PTHREAD_MUTEX_INITIALIZER, {0}, {0}, {0}, {0}, 0, 0, 0};

static void *dev_alloc(gpumt_ctx *h, size_t bytes)
{
	void *p = NULL;
	int slot = -1;
	if (!bytes)
		bytes = 1;
	pthread_mutex_lock(&g_dev.mu);
	if (!g_dev.init) {
		const char *e = getenv("GPUMT_DEVICE_CACHE_MB");
		g_dev.limit = (size_t)(e && *e ? strtoull(e, 0, 10) : 16384) << 20;
		g_dev.init = 1;
	}
	{
		int best = -1;
		for (int i = 0; i < DEV_CACHE_SLOTS; i++)
			if (g_dev.p[i] && !g_dev.used[i] && g_dev.dev[i] == h->device && g_dev.cap[i] >= bytes &&
			    g_dev.cap[i] <= bytes + bytes / 2 + (1u << 20) && (best < 0 || g_dev.cap[i] < g_dev.cap[best]))
				best = i;
		if (best >= 0) {
			g_dev.used[best] = 1;
			g_dev.idle -= g_dev.cap[best];
			p = g_dev.p[best];
		}
	}
	pthread_mutex_unlock(&g_dev.mu);
	if (p)
		return p;
	if (hipMalloc(&p, bytes) != hipSuccess) {
		/* out of memory with idle buffers in the cache: give them back and try once more */
		pthread_mutex_lock(&g_dev.mu);
		for (int i = 0; i < DEV_CACHE_SLOTS; i++)
			if (g_dev.p[i] && !g_dev.used[i] && g_dev.dev[i] == h->device) {
				(void)hipFree(g_dev.p[i]);
				g_dev.idle -= g_dev.cap[i];
				g_dev.p[i] = NULL;
			}
		pthread_mutex_unlock(&g_dev.mu);
		if (hipMalloc(&p, bytes) != hipSuccess)
			return NULL;
	}
	pthread_mutex_lock(&g_dev.mu);
	for (int i = 0; i < DEV_CACHE_SLOTS && slot < 0; i++)
		if (!g_dev.p[i])
			slot = i;
	if (slot >= 0) { /* tracked: its size is known when it comes back (untracked ones are simply freed) */
		g_dev.p[slot] = p;
		g_dev.cap[slot] = bytes;
		g_dev.dev[slot] = h->device;
		g_dev.used[slot] = 1;
	}
	pthread_mutex_unlock(&g_dev.mu);
	return p;
}
static void dev_free(gpumt_ctx *h, void *p)
{
	int keep = 0, slot = -1;
	pthread_mutex_lock(&g_dev.mu);
	for (int i = 0; i < DEV_CACHE_SLOTS && slot < 0; i++)
		if (g_dev.p[i] == p && g_dev.used[i])
			slot = i;
	if (slot >= 0) {
		if (g_dev.idle + g_dev.cap[slot] <= g_dev.limit) {
			keep = 1;
			g_dev.idle += g_dev.cap[slot];
			g_dev.used[slot] = 0;
		} else {
			g_dev.p[slot] = NULL;
			g_dev.used[slot] = 0;
		}
	}
	pthread_mutex_unlock(&g_dev.mu);
	(void)h;
	if (!keep)
		(void)hipFree(p);
}

void *gpumt_malloc(gpumt_ctx *h, size_t bytes)
{
	if (!h || use(h))
		return NULL;
	return dev_alloc(h, bytes);
}
/* debug guard of the "free only idle buffers" contract (include/gpumt.h): a buffer handed back while any stream of its
 * context still has work queued is counted and reported, and the streams are drained before it is recycled */
static unsigned long g_free_busy;
static int free_guard(gpumt_ctx *h, const char *who, void *p)
{
	if (!h->debug_free)
		return 0;
	int busy = 0;
	for (int i = 0; i < GPUMT_NSTREAMS; i++)
		if (hipStreamQuery(h->st[i]) == hipErrorNotReady) {
			busy = 1;
			(void)hipStreamSynchronize(h->st[i]);
		}
	if (busy) {
		__atomic_fetch_add(&g_free_busy, 1ul, __ATOMIC_RELAXED);
		fprintf(stderr, "gpumt: %s(%p) with work queued on the context's streams (drained; GPUMT_DEBUG_FREE)\n", who, p);
	}
	return busy;
}
unsigned long gpumt_debug_free_busy(void) { return __atomic_exchange_n(&g_free_busy, 0ul, __ATOMIC_RELAXED); }

void gpumt_free(gpumt_ctx *h, void *p)
{
	/* the buffer must be idle: it may go to the cache and from there to another context without the
	 * device-wide wait hipFree implies */
	if (h && p && !use(h)) {
		(void)free_guard(h, "gpumt_free", p);
		dev_free(h, p);
	}
}
/*
 * Pinned host memory costs ~0.3 s per GiB to allocate and to free (page pinning), which used to be
 * most of a drop-in API call on a few GiB: freed staging buffers are therefore kept in a
 * process-wide cache and handed to the next context that asks (a decompress context after a
 * compress context, the next file of the CLI, the next call of a long-running caller).  The cache
 * is bounded (GPUMT_PINNED_CACHE_MB, default 16384); hipHostMallocPortable makes a buffer usable by
 * the contexts of every device.
 */
#define PIN_CACHE_SLOTS 64
static struct {
	pthread_mutex_t mu;
	void *p[PIN_CACHE_SLOTS];
	size_t cap[PIN_CACHE_SLOTS];
	size_t total, limit;
	int init;
} g_pin = {PTHREAD_MUTEX_INITIALIZER, {0}, {0}, 0, 0, 0};

struct pin_hdr { /* in front of every pinned buffer: its capacity */
	size_t cap;
	size_t pad[7];
};

void *gpumt_host_alloc(gpumt_ctx *h, size_t bytes)
{
	void *p = NULL;
	if (!h || use(h))
		return NULL;
	if (!bytes)
		bytes = 1;
	pthread_mutex_lock(&g_pin.mu);
	if (!g_pin.init) {
		const char *e = getenv("GPUMT_PINNED_CACHE_MB");
		g_pin.limit = (size_t)(e && *e ? strtoull(e, 0, 10) : 16384) << 20;
		g_pin.init = 1;
	}
	{
		int best = -1;
		for (int i = 0; i < PIN_CACHE_SLOTS; i++)
			if (g_pin.p[i] && g_pin.cap[i] >= bytes && g_pin.cap[i] <= bytes + bytes / 2 + (1u << 20) &&
			    (best < 0 || g_pin.cap[i] < g_pin.cap[best]))
				best = i;
		if (best >= 0) {
			p = g_pin.p[best];
			g_pin.total -= g_pin.cap[best];
			g_pin.p[best] = NULL;
		}
	}
	pthread_mutex_unlock(&g_pin.mu);
	if (p)
		return p;
	/* non-coherent = CPU-cached pinned memory: the callbacks memcpy in and out of it at full host
	 * speed; visibility is established by the stream synchronisation the engine does anyway */
	if (hipHostMalloc(&p, bytes + sizeof(struct pin_hdr), hipHostMallocNonCoherent | hipHostMallocPortable) !=
	    hipSuccess)
		return NULL;
	((struct pin_hdr *)p)->cap = bytes;
	return (u8 *)p + sizeof(struct pin_hdr);
}
void gpumt_host_free(gpumt_ctx *h, void *p)
{
	if (!h || !p)
		return;
	if (h->debug_free && !use(h))
		(void)free_guard(h, "gpumt_host_free", p);
	struct pin_hdr *hd = (struct pin_hdr *)((u8 *)p - sizeof(struct pin_hdr));
	const size_t cap = hd->cap;
	pthread_mutex_lock(&g_pin.mu);
	if (g_pin.total + cap <= g_pin.limit) {
		for (int i = 0; i < PIN_CACHE_SLOTS; i++)
			if (!g_pin.p[i]) {
				g_pin.p[i] = p;
				g_pin.cap[i] = cap;
				g_pin.total += cap;
				p = NULL;
				break;
			}
	}
	pthread_mutex_unlock(&g_pin.mu);
	if (p && !use(h))
		(void)hipHostFree(hd);
}

/* pin memory the caller owns (a mapping shared by the ranks of a multi-GPU job, an application buffer) so that
 * gpumt_memcpy_d2h / _h2d move it at copy-engine speed without a staging pass */
int gpumt_host_register(gpumt_ctx *h, void *p, size_t bytes)
{
	if (!h || !p || !bytes)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipHostRegister(p, bytes, hipHostRegisterPortable));
	return GPUMT_OK;
}
int gpumt_host_unregister(gpumt_ctx *h, void *p)
{
	if (!h || !p)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipHostUnregister(p));
	return GPUMT_OK;
}

/* give the idle buffers of the process-wide caches (device memory of this context's device, pinned host memory)
 * back to the system: a long-running host that compressed once does not have to keep up to 2 x 16 GiB alive.
 * Returns the number of bytes released. */
size_t gpumt_trim_caches(gpumt_ctx *h)
{
	size_t freed = 0;
	if (!h || use(h))
		return 0;
	pthread_mutex_lock(&g_dev.mu);
	for (int i = 0; i < DEV_CACHE_SLOTS; i++)
		if (g_dev.p[i] && !g_dev.used[i] && g_dev.dev[i] == h->device) {
			(void)hipFree(g_dev.p[i]);
			g_dev.idle -= g_dev.cap[i];
			freed += g_dev.cap[i];
			g_dev.p[i] = NULL;
		}
	pthread_mutex_unlock(&g_dev.mu);
	pthread_mutex_lock(&g_pin.mu);
	for (int i = 0; i < PIN_CACHE_SLOTS; i++)
		if (g_pin.p[i]) {
			(void)hipHostFree((u8 *)g_pin.p[i] - sizeof(struct pin_hdr));
			g_pin.total -= g_pin.cap[i];
			freed += g_pin.cap[i];
			g_pin.p[i] = NULL;
		}
	pthread_mutex_unlock(&g_pin.mu);
	return freed;
}

#define STREAM_OK(s) ((s) >= 0 && (s) < GPUMT_NSTREAMS)

int gpumt_memcpy_h2d(gpumt_ctx *h, void *dst, const void *src, size_t n, int s)
{
	if (!h || !STREAM_OK(s))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, h->st[s]));
	return GPUMT_OK;
}
int gpumt_memcpy_d2h(gpumt_ctx *h, void *dst, const void *src, size_t n, int s)
{
	if (!h || !STREAM_OK(s))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, h->st[s]));
	return GPUMT_OK;
}
int gpumt_push_host(gpumt_ctx *h, void *dst_host, const void *src, size_t n, const uint64_t *d_n, int s)
{
	void *dp = NULL;
	if (!h || !STREAM_OK(s) || ((uintptr_t)dst_host & 15) || ((uintptr_t)src & 15))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	if (!n)
		return GPUMT_OK;
	CK(hipHostGetDevicePointer(&dp, dst_host, 0));
	const size_t nv = n >> 4;
	unsigned grid = (unsigned)((nv + 255) / 256);
	if (grid > 512)
		grid = 512; /* a few waves per CU saturate the host link and leave the rest to the codec kernels */
	if (grid < 1)
		grid = 1;
	hipLaunchKernelGGL(zmt_push_host_kernel, dim3(grid), dim3(256), 0, h->st[s], (const u8 *)src, (u8 *)dp, (u64)n,
			   (const u64 *)d_n);
	CK(hipGetLastError());
	return GPUMT_OK;
}
int gpumt_memcpy_d2d(gpumt_ctx *h, void *dst, const void *src, size_t n, int s)
{
	if (!h || !STREAM_OK(s))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, h->st[s]));
	return GPUMT_OK;
}
int gpumt_memset(gpumt_ctx *h, void *dst, int byte, size_t n, int s)
{
	if (!h || !STREAM_OK(s))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipMemsetAsync(dst, byte, n, h->st[s]));
	return GPUMT_OK;
}
int gpumt_stream_sync(gpumt_ctx *h, int s)
{
	if (!h || !STREAM_OK(s))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipStreamSynchronize(h->st[s]));
	return GPUMT_OK;
}
int gpumt_device_sync(gpumt_ctx *h)
{
	if (!h)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipDeviceSynchronize());
	return GPUMT_OK;
}
int gpumt_mark(gpumt_ctx *h, int id, int s)
{
	if (!h || !STREAM_OK(s) || id < 0 || id >= GPUMT_NMARKS)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipEventRecord(h->mark[id], h->st[s]));
	return GPUMT_OK;
}
int gpumt_mark_sync(gpumt_ctx *h, int id)
{
	if (!h || id < 0 || id >= GPUMT_NMARKS)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipEventSynchronize(h->mark[id]));
	return GPUMT_OK;
}
int gpumt_stream_wait(gpumt_ctx *h, int waiter, int signaler)
{
	hipEvent_t ev;
	if (!h || !STREAM_OK(waiter) || !STREAM_OK(signaler))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	/* a fresh event per edge: edges may be in flight concurrently */
	CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	CK(hipEventRecord(ev, h->st[signaler]));
	CK(hipStreamWaitEvent(h->st[waiter], ev, 0));
	CK(hipEventDestroy(ev)); /* destruction is deferred until the event completes */
	return GPUMT_OK;
}
void *gpumt_stream_handle(gpumt_ctx *h, int s)
{
	return (h && STREAM_OK(s)) ? (void *)h->st[s] : NULL;
}

int gpumt_timer_start(gpumt_ctx *h, int slot, int s)
{
	if (!h || slot < 0 || slot >= NTIMERS || !STREAM_OK(s))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipEventRecord(h->t0[slot], h->st[s]));
	return GPUMT_OK;
}
int gpumt_timer_stop(gpumt_ctx *h, int slot, int s)
{
	if (!h || slot < 0 || slot >= NTIMERS || !STREAM_OK(s))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipEventRecord(h->t1[slot], h->st[s]));
	return GPUMT_OK;
}
int gpumt_timer_ms(gpumt_ctx *h, int slot, float *ms)
{
	if (!h || slot < 0 || slot >= NTIMERS || !ms)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	CK(hipEventSynchronize(h->t1[slot]));
	CK(hipEventElapsedTime(ms, h->t0[slot], h->t1[slot]));
	return GPUMT_OK;
}

/* ------------------------------------------------------------------------------- LZ4 */
size_t gpumt_lz4_slot_stride(size_t chunk)
{
	size_t full = chunk / 65536, part = chunk % 65536;
	size_t b = 12 + 19 + 4 * (full + (part ? 1 : 0)) + chunk + 8;
	return (b + 255) & ~(size_t)255;
}

size_t gpumt_lz4_record_count(size_t n, size_t chunk)
{
	if (!chunk)
		return 0;
	return n ? (n + chunk - 1) / chunk : 1;
}

int gpumt_xxh32_batch(gpumt_ctx *h, const void *d_base, const uint64_t *d_off,
		      const uint32_t *d_len, size_t n, uint32_t *d_hash, int s)
{
	if (!h || !STREAM_OK(s) || n > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	if (!n)
		return GPUMT_OK;
	hipLaunchKernelGGL(zmt_xxh32_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0,
			   h->st[s], (const u8 *)d_base, d_off, d_len, (u32)n, d_hash,
			   (const u32 *)NULL, (const u32 *)NULL, (u32 *)NULL);
	CK(hipGetLastError());
	return GPUMT_OK;
}

int gpumt_lz4_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk,
			     void *d_slots, size_t slot_stride, uint32_t *d_rec_len, int s)
{
	return gpumt_lz4_compress_batch_level(h, d_in, n, chunk, d_slots, slot_stride, d_rec_len, 1, s);
}

int gpumt_lz4_level_supported(int level) { return level >= 1 && level <= 12; }

int gpumt_lz4_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk,
				   void *d_slots, size_t slot_stride, uint32_t *d_rec_len, int level, int s)
{
	size_t nrec = gpumt_lz4_record_count(n, chunk);
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || nrec > 0x3FFFFFFFu ||
	    slot_stride < gpumt_lz4_slot_stride(chunk) || !gpumt_lz4_level_supported(level))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	/* scratch: off[nrec] u64 | len[nrec] u32 | chk[nrec] u32 | (HC) one table set per wave */
	const bool hc = level >= 3;
	const size_t hc_max = h->hc_waves > 0 ? (size_t)h->hc_waves : GPUMT_LZ4HC_WAVES;
	const size_t hc_grid = nrec < hc_max ? nrec : hc_max;
	const size_t hc_base = (nrec * 16 + 255) & ~(size_t)255;
	if (want_scratch(h, 0, s, hc ? hc_base + hc_grid * GPUMT_LZ4HC_SCRATCH : nrec * 16))
		return GPUMT_E_HIP;
	u64 *off = (u64 *)h->scratch[0][s];
	u32 *len = (u32 *)(off + nrec);
	u32 *chk = len + nrec;
	hipLaunchKernelGGL(zmt_iota_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0,
			   h->st[s], off, len, (u64)n, (u64)chunk, (u32)nrec);
	PROF0(8);
	hipLaunchKernelGGL(zmt_xxh32_kernel, dim3((unsigned)((nrec * 4 + 255) / 256)), dim3(256), 0,
			   h->st[s], (const u8 *)d_in, (const u64 *)off, (const u32 *)len, (u32)nrec,
			   chk, (const u32 *)NULL, (const u32 *)NULL, (u32 *)NULL);
	PROF1(8);
	PROF0(9);
	unsigned long long *eprof = NULL;
	if (h->profile == 4) {
		if (!h->d_prof) {
			CK(hipMalloc((void **)&h->d_prof, 16 * sizeof(unsigned long long)));
			CK(hipMemset(h->d_prof, 0, 16 * sizeof(unsigned long long)));
		}
		eprof = h->d_prof;
	}
	if (hc) {
		/* levels 3..9: hash-chain parser, 10..12: optimal parser (lz4_enc_hc.hip) */
		hipLaunchKernelGGL(zmt_lz4hc_enc_kernel, dim3((unsigned)hc_grid), dim3(64), 0, h->st[s],
				   (const u8 *)d_in, (u64)n, (u32)chunk, (u32)nrec, (u8 *)d_slots,
				   (u64)slot_stride, d_rec_len, (const u32 *)chk,
				   (u8 *)h->scratch[0][s] + hc_base, level);
	} else {
		/* chunk <= 64 KiB: every record is a single independent block (byU16 table).  Else linked-block records
		 * (17-bit entries up to 128 KiB chunks, 32-bit beyond); the ragged last record may be <= 64 KiB and then
		 * belongs to the byU16 kernel (each kernel skips records of the other kind) */
		const bool v3 = h->enc_variant == 3 || eprof;
		typedef void (*enc_fn)(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
		const enc_fn k16 = v3 ? zmt_lz4_enc3_u16_kernel : zmt_lz4_enc5_u16_kernel;
		const enc_fn k17 = eprof ? zmt_lz4_enc3_p17_prof_kernel : v3 ? zmt_lz4_enc3_p17_kernel : zmt_lz4_enc5_p17_kernel;
		const enc_fn k32 = v3 ? zmt_lz4_enc3_u32_kernel : zmt_lz4_enc5_u32_kernel;
		if (chunk <= 65536) {
			hipLaunchKernelGGL(k16, dim3((unsigned)nrec), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
					   (u32)chunk, 0u, (u32)nrec, (u8 *)d_slots, (u64)slot_stride, d_rec_len,
					   (const u32 *)chk, eprof);
		} else {
			hipLaunchKernelGGL(chunk <= 131072 ? k17 : k32, dim3((unsigned)nrec), dim3(64),
					   chunk <= 131072 ? (size_t)h->xflags : 0 /* developer: dynamic-LDS padding = occupancy knob */,
					   h->st[s], (const u8 *)d_in, (u64)n, (u32)chunk, 0u, (u32)nrec, (u8 *)d_slots,
					   (u64)slot_stride, d_rec_len, (const u32 *)chk, eprof);
			hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n, (u32)chunk,
					   (u32)(nrec - 1), (u32)nrec, (u8 *)d_slots, (u64)slot_stride, d_rec_len,
					   (const u32 *)chk, (unsigned long long *)NULL);
		}
	}
	PROF1(9);
	CK(hipGetLastError());
	return GPUMT_OK;
}

int gpumt_lz4_compact(gpumt_ctx *h, const void *d_slots, size_t slot_stride,
		      const uint32_t *d_rec_len, size_t nrec, void *d_stream, uint64_t *d_rec_off,
		      int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	PROF0(10);
	hipLaunchKernelGGL(zmt_scan_kernel, dim3(1), dim3(1024), 0, h->st[s], d_rec_len, (u32)nrec,
			   d_rec_off);
	hipLaunchKernelGGL(zmt_compact_kernel, dim3((unsigned)nrec), dim3(256), 0, h->st[s],
			   (const u8 *)d_slots, (u64)slot_stride, d_rec_len,
			   (const u64 *)d_rec_off, (u32)nrec, (u8 *)d_stream);
	PROF1(10);
	CK(hipGetLastError());
	return GPUMT_OK;
}

int gpumt_lz4_probe_sizes(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
			  const uint32_t *d_rec_len, size_t nrec, uint32_t *d_out_len,
			  uint64_t *d_out_off, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	hipLaunchKernelGGL(zmt_probe_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0,
			   h->st[s], (const u8 *)d_stream, d_rec_off, d_rec_len, (u32)nrec,
			   d_out_len);
	hipLaunchKernelGGL(zmt_scan_kernel, dim3(1), dim3(1024), 0, h->st[s],
			   (const u32 *)d_out_len, (u32)nrec, d_out_off);
	CK(hipGetLastError());
	return GPUMT_OK;
}

int gpumt_lz4_decompress_batch(gpumt_ctx *h, const void *d_stream, size_t stream_bytes,
			       const uint64_t *d_rec_off, const uint32_t *d_rec_len, size_t nrec,
			       void *d_out, size_t out_bytes, const uint64_t *d_out_off,
			       uint32_t *d_out_len, uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	const u32 n = (u32)nrec;
	const unsigned g256 = (unsigned)((nrec + 255) / 256);
	/* scratch carve-up (all offsets 16-byte aligned) */
	const size_t nblk_max = out_bytes / 65536 + nrec + 1;
	const size_t ntok_max = stream_bytes / 3 + 128 * nblk_max + 256;
	size_t off = 0;
#define CARVE(var, type, count)                                                                    \
	size_t var##_o = off;                                                                      \
	off += (((size_t)(count) * sizeof(type)) + 15) & ~(size_t)15;
	CARVE(ce, u32, nrec)
	CARVE(cv, u32, nrec)
	CARVE(est, u32, nrec)
	CARVE(blk0, u64, nrec + 1)
	CARVE(rnb, u32, nrec)
	CARVE(rfl, u32, nrec)
	CARVE(bco, u64, nblk_max)
	CARVE(bcs, u32, nblk_max)
	CARVE(bnt, u32, nblk_max)
	CARVE(bol, u32, nblk_max)
	CARVE(tok, u16, ntok_max)
#undef CARVE
	const bool split = h->dec_variant == 0;
	if (h->profile >= 2 && !h->d_prof) {
		CK(hipMalloc((void **)&h->d_prof, 16 * sizeof(unsigned long long)));
		CK(hipMemset(h->d_prof, 0, 16 * sizeof(unsigned long long)));
	}
	if (want_scratch(h, 1, s, split ? off : nrec * 8 + 32))
		return GPUMT_E_HIP;
	u8 *sc = (u8 *)h->scratch[1][s];
	u32 *ce = (u32 *)(sc + ce_o), *cv = (u32 *)(sc + cv_o);
	PROF0(11);
	if (split) {
		u32 *est = (u32 *)(sc + est_o), *rnb = (u32 *)(sc + rnb_o), *rfl = (u32 *)(sc + rfl_o);
		u64 *blk0 = (u64 *)(sc + blk0_o), *bco = (u64 *)(sc + bco_o);
		u32 *bcs = (u32 *)(sc + bcs_o), *bnt = (u32 *)(sc + bnt_o), *bol = (u32 *)(sc + bol_o);
		u16 *tok = (u16 *)(sc + tok_o);
		PROF0(13);
		hipLaunchKernelGGL(zmt_dec_nblk_kernel, dim3(g256), dim3(256), 0, h->st[s], d_out_len, n, est);
		hipLaunchKernelGGL(zmt_scan_kernel, dim3(1), dim3(1024), 0, h->st[s], (const u32 *)est, n, blk0);
		hipLaunchKernelGGL(zmt_dec_frames_kernel, dim3(g256), dim3(256), 0, h->st[s],
				   (const u8 *)d_stream, d_rec_off, d_rec_len, n, d_out_len,
				   (const u64 *)blk0, bco, bcs, rnb, rfl, d_status, ce, cv);
		PROF1(13);
		const int ring = h->lz4_ring < 12 ? 12 : h->lz4_ring > 14 ? 14 : h->lz4_ring;
		PROF0(14);
		hipLaunchKernelGGL(zmt_dec_parse4_kernel, dim3((unsigned)((nblk_max + 63) / 64)), dim3(64), 0,
				   h->st[s], (const u8 *)d_stream, (u64)stream_bytes, (const u64 *)bco,
				   (const u32 *)bcs, (const u64 *)(blk0 + nrec), tok, bnt, bol);
		PROF1(14);
		PROF0(15);
#define C3_LAUNCH(NAME)                                                                                            \
	hipLaunchKernelGGL(NAME, dim3((unsigned)nrec), dim3(64), (size_t)h->dec_pad, h->st[s], (const u8 *)d_stream, \
			   (u64)stream_bytes, 0u, n, (u8 *)d_out, d_out_off, d_out_len, (const u64 *)blk0,         \
			   (const u64 *)bco, (const u32 *)bcs, (const u32 *)rnb, (const u32 *)rfl, (const u16 *)tok, \
			   (const u32 *)bnt, (const u32 *)bol, d_status)
#define C3_LAUNCHP(NAME)                                                                                           \
	hipLaunchKernelGGL(NAME, dim3((unsigned)nrec), dim3(64), 0, h->st[s], (const u8 *)d_stream,               \
			   (u64)stream_bytes, 0u, n, (u8 *)d_out, d_out_off, d_out_len, (const u64 *)blk0,         \
			   (const u64 *)bco, (const u32 *)bcs, (const u32 *)rnb, (const u32 *)rfl, (const u16 *)tok, \
			   (const u32 *)bnt, (const u32 *)bol, d_status, h->d_prof)
		/* (one stream, one launch per stage.  Slices of the records on internal streams -- a slice's parse under the copy of
		 * the one before it, its XXH32 verify under the copy of the next -- were built and measured in rounds 2 and 6: a kernel
		 * that arrives while the device is full of copy waves is starved, not interleaved, and a slice's parse lasts as long as
		 * the whole batch's; 13.4-32.9 ms against 13.55, profiles/r06_sweeps/lz4_dec_slices.txt) */
		if (h->profile == 8) {
			if (ring == 12)
				C3_LAUNCHP(zmt_dec_copy3_w4_kernel_prof);
			else if (ring == 13)
				C3_LAUNCHP(zmt_dec_copy3_w8_kernel_prof);
			else
				C3_LAUNCHP(zmt_dec_copy3_w16_kernel_prof);
		} else {
			if (ring == 12)
				C3_LAUNCH(zmt_dec_copy3_w4_kernel);
			else if (ring == 13)
				C3_LAUNCH(zmt_dec_copy3_w8_kernel);
			else
				C3_LAUNCH(zmt_dec_copy3_w16_kernel);
		}
		PROF1(15);
		/* records the fast path does not cover (block size > 64 KiB, odd block counts) */
		hipLaunchKernelGGL(zmt_lz4_dec_serial, dim3(n), dim3(64), 0, h->st[s], (const u8 *)d_stream,
				   d_rec_off, d_rec_len, n, (u8 *)d_out, d_out_off, d_out_len, d_status, ce,
				   cv, 100u);
	} else {
		hipLaunchKernelGGL(zmt_lz4_dec_serial, dim3(n), dim3(64), 0, h->st[s], (const u8 *)d_stream,
				   d_rec_off, d_rec_len, n, (u8 *)d_out, d_out_off, d_out_len, d_status, ce,
				   cv, 0xFFFFFFFFu);
	}
	PROF1(11);
	PROF0(12);
	hipLaunchKernelGGL(zmt_xxh32_kernel, dim3((unsigned)((nrec * 4 + 255) / 256)), dim3(256), 0,
			   h->st[s], (const u8 *)d_out, d_out_off, d_out_len, n, (u32 *)NULL,
			   (const u32 *)ce, (const u32 *)cv, d_status);
	PROF1(12);
	CK(hipGetLastError());
	return GPUMT_OK;
}

/* ------------------------------------------------------------------ zstd */
#define ZE_BLOCK 131072u
#define ZE_BSTRIDE (ZE_BLOCK + 16u)
#define ZE_HDR 32u
#define ZE_MAXSEQ (ZE_BLOCK / 4u)

size_t gpumt_zstd_slot_stride(size_t chunk)
{
	const size_t nb = chunk ? (chunk + ZE_BLOCK - 1) / ZE_BLOCK : 1;
	return (ZE_HDR + nb * ZE_BSTRIDE + 255) & ~(size_t)255;
}

/* level -> encoder tier (zstd_enc.hip): 1-2 the fast parse (4 Ki-entry table, 16 waves per CU), 3-9 and 10-22
 * larger tables and a 6-byte hash (8 / 4 waves per CU) */
int gpumt_zstd_level_tier(int level) { return level <= 2 ? 0 : level <= 9 ? 1 : 2; }

int gpumt_zstd_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
			      size_t slot_stride, uint32_t *d_rec_len, int s)
{
	return gpumt_zstd_compress_batch_level(h, d_in, n, chunk, d_slots, slot_stride, d_rec_len, 1, s);
}

int gpumt_zstd_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				    size_t slot_stride, uint32_t *d_rec_len, int level, int s)
{
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || slot_stride < gpumt_zstd_slot_stride(chunk) ||
	    level < 1 || level > 22)
		return GPUMT_E_ARG;
	const int tier = h->profile == 6 ? 0 : gpumt_zstd_level_tier(level);
	if (use(h))
		return GPUMT_E_HIP;
	const size_t nrec = gpumt_lz4_record_count(n, chunk);
	const u32 bpr = (u32)((chunk + ZE_BLOCK - 1) / ZE_BLOCK);
	const size_t nblk = nrec * bpr;
	if (nblk > 0x7FFFFFFFu)
		return GPUMT_E_ARG;
	/* persistent waves, exactly as many as stay resident (LDS-limited), blocks taken round-robin */
	if (!h->zenc_waves[tier]) {
		int per_cu = 0;
		if (tier == 0)
			CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zmt_zstd_enc_kernel, 64, 0));
		else if (tier == 1)
			CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zmt_zstd_enc_t2_kernel, 64, 0));
		else
			CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zmt_zstd_enc_t3_kernel, 64, 0));
		h->zenc_waves[tier] = (per_cu > 0 ? per_cu : 4) * (h->num_cus > 0 ? h->num_cus : 256);
	}
	const unsigned grid = (unsigned)(nblk < (size_t)h->zenc_waves[tier] ? nblk : (size_t)h->zenc_waves[tier]);
	const size_t seq_bytes = (size_t)grid * (3 * ZE_MAXSEQ * 4 + 16 * 20544 + ZE_BLOCK + 64);
	if (h->profile == 6)
		fprintf(stderr, "gpumt: zstd encoder grid %u waves\n", grid);
	if (want_scratch(h, 0, s, nblk * 4 + 64 + seq_bytes))
		return GPUMT_E_HIP;
	u32 *blk_len = (u32 *)h->scratch[0][s];
	u8 *seqbuf = (u8 *)h->scratch[0][s] + ((nblk * 4 + 63) & ~(size_t)63);
	if (h->profile == 6 && !h->d_prof) {
		CK(hipMalloc((void **)&h->d_prof, 16 * sizeof(unsigned long long)));
		CK(hipMemset(h->d_prof, 0, 16 * sizeof(unsigned long long)));
	}
	PROF0(9);
	if (h->profile == 6)
		hipLaunchKernelGGL(zmt_zstd_enc_kernel_prof, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in,
				   (u64)n, (u32)chunk, (u32)nblk, bpr, (u8 *)d_slots, (u64)slot_stride, blk_len,
				   seqbuf, h->d_prof);
	else if (tier == 0)
		hipLaunchKernelGGL(zmt_zstd_enc_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
				   (u32)chunk, (u32)nblk, bpr, (u8 *)d_slots, (u64)slot_stride, blk_len, seqbuf);
	else if (tier == 1)
		hipLaunchKernelGGL(zmt_zstd_enc_t2_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
				   (u32)chunk, (u32)nblk, bpr, (u8 *)d_slots, (u64)slot_stride, blk_len, seqbuf);
	else
		hipLaunchKernelGGL(zmt_zstd_enc_t3_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
				   (u32)chunk, (u32)nblk, bpr, (u8 *)d_slots, (u64)slot_stride, blk_len, seqbuf);
	hipLaunchKernelGGL(zmt_zstd_assemble_kernel, dim3((unsigned)nrec), dim3(256), 0, h->st[s], (u64)n,
			   (u32)chunk, (u32)nrec, bpr, (u8 *)d_slots, (u64)slot_stride, (const u32 *)blk_len,
			   d_rec_len);
	PROF1(9);
	CK(hipGetLastError());
	return GPUMT_OK;
}

int gpumt_zstd_probe_sizes(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
			   const uint32_t *d_rec_len, size_t nrec, uint32_t *d_out_len,
			   uint64_t *d_out_off, uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	hipLaunchKernelGGL(zmt_zstd_probe_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0,
			   h->st[s], (const u8 *)d_stream, d_rec_off, d_rec_len, (u32)nrec, d_out_len,
			   d_status);
	hipLaunchKernelGGL(zmt_scan_kernel, dim3(1), dim3(1024), 0, h->st[s],
			   (const u32 *)d_out_len, (u32)nrec, d_out_off);
	CK(hipGetLastError());
	return GPUMT_OK;
}

/* developer A/B: dynamic-LDS padding of the sequence pre-pass = a cap on its resident waves (GPUMT_ZSEQ_PAD, bytes) */
static size_t zseq_pad(void)
{
	static long v = -1;
	if (v < 0) {
		const char *e = getenv("GPUMT_ZSEQ_PAD");
		v = e && *e ? atol(e) : 0;
	}
	return (size_t)v;
}

int gpumt_zstd_decompress_batch(gpumt_ctx *h, const void *d_stream, size_t stream_bytes,
				const uint64_t *d_rec_off, const uint32_t *d_rec_len, size_t nrec,
				void *d_out, size_t out_bytes, const uint64_t *d_out_off,
				uint32_t *d_out_len, uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	/* per-record scratch (literals + sequences decoded ahead) is what bounds a launch: records go
	 * in slices of at most ZD_SLICE (5 GiB of scratch), each slice still 4x the resident waves */
	const size_t ZD_SLICE = 16384;
	const size_t slice = nrec < ZD_SLICE ? nrec : ZD_SLICE;
	const size_t lit_bytes = slice * (size_t)GPUMT_ZSTD_DEC_SCRATCH;
	/* the sequence pre-pass (zstd_dec_seq.hip) writes into a buffer as large as the output, record i at its d_out_off:
	 * only batches that can hold frames of more than one block have one */
	const size_t chk_bytes = (nrec * 8 + 64 + 15) & ~(size_t)15;
	const bool seq_on = h->zseq_variant == 0 && out_bytes / nrec > 131072;
	const size_t seq_bytes = seq_on ? out_bytes + 32 : 0;
	if (want_scratch(h, 1, s, lit_bytes + chk_bytes + seq_bytes))
		return GPUMT_E_HIP;
	u32 *chk_e = (u32 *)((u8 *)h->scratch[1][s] + lit_bytes), *chk_v = chk_e + nrec;
	/* the region's capacity goes to both kernels (zstd_dec_seq.h): records whose [out_off, out_off + out_len) does not fit
	 * out_bytes are decoded without the pre-pass instead of writing past the scratch */
	u8 *seqbuf = seq_on ? (u8 *)h->scratch[1][s] + lit_bytes + chk_bytes + 16 : NULL;
	const u64 seqcap = (u64)out_bytes;
	if (h->profile == 5 && !h->d_prof) {
		CK(hipMalloc((void **)&h->d_prof, 16 * sizeof(unsigned long long)));
		CK(hipMemset(h->d_prof, 0, 16 * sizeof(unsigned long long)));
	}
	PROF0(11);
	for (size_t b = 0; b < nrec; b += slice) {
		const size_t m = nrec - b < slice ? nrec - b : slice;
		if (seqbuf)
			hipLaunchKernelGGL(zmt_zstd_seq_kernel, dim3((unsigned)m), dim3(64), zseq_pad(), h->st[s],
					   (const u8 *)d_stream, (u64)stream_bytes, d_rec_off + b, d_rec_len + b, (u32)m,
					   d_out_off + b, (const u32 *)(d_out_len + b), (const u32 *)(d_status + b), seqbuf, seqcap);
		if (h->profile == 5) {
			hipLaunchKernelGGL(zmt_zstd_dec_kernel_prof, dim3((unsigned)m), dim3(64), 0, h->st[s],
					   (const u8 *)d_stream, (u64)stream_bytes, d_rec_off + b, d_rec_len + b, (u32)m,
					   (u8 *)d_out, d_out_off + b, d_out_len + b, (u8 *)h->scratch[1][s], d_status + b,
					   chk_e + b, chk_v + b, 0u, h->d_prof, seqbuf, seqcap);
		} else {
			/* small-table variant first (16 waves per CU); records that need the full-size tables
			 * come back with status 101 and are decoded by the general variant (12 waves per CU) */
			const u32 want = h->zdec_variant == 1 ? 0u : 101u;
			if (h->zdec_variant != 1)
				hipLaunchKernelGGL(zmt_zstd_dec_small_kernel, dim3((unsigned)m), dim3(64), 0, h->st[s],
						   (const u8 *)d_stream, (u64)stream_bytes, d_rec_off + b, d_rec_len + b, (u32)m,
						   (u8 *)d_out, d_out_off + b, d_out_len + b, (u8 *)h->scratch[1][s],
						   d_status + b, chk_e + b, chk_v + b, seqbuf, seqcap);
			hipLaunchKernelGGL(zmt_zstd_dec_kernel, dim3((unsigned)m), dim3(64), 0, h->st[s],
					   (const u8 *)d_stream, (u64)stream_bytes, d_rec_off + b, d_rec_len + b, (u32)m,
					   (u8 *)d_out, d_out_off + b, d_out_len + b, (u8 *)h->scratch[1][s], d_status + b,
					   chk_e + b, chk_v + b, want, seqbuf, seqcap);
		}
	}
	/* XXH64 content checksums, for the frames that carry one */
	hipLaunchKernelGGL(zmt_xxh64_verify_kernel, dim3((unsigned)((nrec * 4 + 255) / 256)), dim3(256), 0,
			   h->st[s], (const u8 *)d_out, d_out_off, (const u32 *)d_out_len, (u32)nrec, (const u32 *)chk_e,
			   (const u32 *)chk_v, d_status);
	PROF1(11);
	CK(hipGetLastError());
	return GPUMT_OK;
}

/* brotli-mt compress: same slot geometry as zstd (gpumt_zstd_slot_stride), 16-byte record headers */
/* quality -> encoder tier (brotli_enc.hip): 0-3 the fast parse (4 Ki-entry table), 4-8 and 9-11 larger tables and
 * minimum match 6 */
int gpumt_brotli_level_tier(int level) { return level <= 3 ? 0 : level <= 8 ? 1 : 2; }

int gpumt_brotli_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				size_t slot_stride, uint32_t *d_rec_len, int s)
{
	return gpumt_brotli_compress_batch_level(h, d_in, n, chunk, d_slots, slot_stride, d_rec_len, 1, s);
}

int gpumt_brotli_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				      size_t slot_stride, uint32_t *d_rec_len, int level, int s)
{
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || slot_stride < gpumt_zstd_slot_stride(chunk) ||
	    level < 0 || level > 11)
		return GPUMT_E_ARG;
	const int tier = gpumt_brotli_level_tier(level);
	if (use(h))
		return GPUMT_E_HIP;
	const size_t nrec = gpumt_lz4_record_count(n, chunk);
	const u32 bpr = (u32)((chunk + ZE_BLOCK - 1) / ZE_BLOCK);
	const size_t nblk = nrec * bpr;
	if (nblk > 0x7FFFFFFFu)
		return GPUMT_E_ARG;
	if (!h->benc_waves[tier]) {
		int per_cu = 0;
		if (tier == 0)
			CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zmt_brotli_enc_kernel, 64, 0));
		else if (tier == 1)
			CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zmt_brotli_enc_t2_kernel, 64, 0));
		else
			CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zmt_brotli_enc_t3_kernel, 64, 0));
		h->benc_waves[tier] = (per_cu > 0 ? per_cu : 4) * (h->num_cus > 0 ? h->num_cus : 256);
		if (getenv("GPUMT_VERBOSE"))
			fprintf(stderr, "gpumt: brotli encoder tier %d grid %d waves\n", tier, h->benc_waves[tier]);
	}
	const unsigned grid = (unsigned)(nblk < (size_t)h->benc_waves[tier] ? nblk : (size_t)h->benc_waves[tier]);
	const size_t seq_bytes = (size_t)grid * (3 * ZE_MAXSEQ * 4);
	if (want_scratch(h, 0, s, nblk * 4 + 64 + seq_bytes))
		return GPUMT_E_HIP;
	u32 *blk_len = (u32 *)h->scratch[0][s];
	u8 *seqbuf = (u8 *)h->scratch[0][s] + ((nblk * 4 + 63) & ~(size_t)63);
	PROF0(9);
	if (tier == 0)
		hipLaunchKernelGGL(zmt_brotli_enc_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
				   (u32)chunk, (u32)nblk, bpr, (u8 *)d_slots, (u64)slot_stride, blk_len, seqbuf);
	else if (tier == 1)
		hipLaunchKernelGGL(zmt_brotli_enc_t2_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
				   (u32)chunk, (u32)nblk, bpr, (u8 *)d_slots, (u64)slot_stride, blk_len, seqbuf);
	else
		hipLaunchKernelGGL(zmt_brotli_enc_t3_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
				   (u32)chunk, (u32)nblk, bpr, (u8 *)d_slots, (u64)slot_stride, blk_len, seqbuf);
	hipLaunchKernelGGL(zmt_brotli_assemble_kernel, dim3((unsigned)nrec), dim3(256), 0, h->st[s], (u64)n,
			   (u32)chunk, (u32)nrec, bpr, (u8 *)d_slots, (u64)slot_stride, (const u32 *)blk_len,
			   d_rec_len);
	PROF1(9);
	CK(hipGetLastError());
	return GPUMT_OK;
}

extern "C" const unsigned char zmt_brotli_static[], zmt_brotli_static_end[];

int gpumt_brotli_decompress_batch(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
				  const uint32_t *d_rec_len, size_t nrec, void *d_out,
				  const uint64_t *d_out_off, const uint32_t *d_out_cap,
				  uint32_t *d_out_len, uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	if (!h->d_brotli_static) {
		const size_t n = (size_t)(zmt_brotli_static_end - zmt_brotli_static);
		CK(hipMalloc(&h->d_brotli_static, n + 64));
		CK(hipMemcpy(h->d_brotli_static, zmt_brotli_static, n, hipMemcpyHostToDevice));
	}
	if (!h->bdec_waves) {
		int per_cu = 0;
		CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zmt_brotli_dec_kernel, 64, 0));
		h->bdec_waves = (per_cu > 0 ? per_cu : 4) * (h->num_cus > 0 ? h->num_cus : 256);
		if (getenv("GPUMT_VERBOSE"))
			fprintf(stderr, "gpumt: brotli decoder grid %d waves\n", h->bdec_waves);
	}
	const unsigned grid = (unsigned)(nrec < (size_t)h->bdec_waves ? nrec : (size_t)h->bdec_waves);
	if (want_scratch(h, 1, s, (size_t)grid * GPUMT_BROTLI_SCRATCH))
		return GPUMT_E_HIP;
	if (h->profile == 7 && !h->d_prof) {
		CK(hipMalloc((void **)&h->d_prof, 16 * sizeof(unsigned long long)));
		CK(hipMemset(h->d_prof, 0, 16 * sizeof(unsigned long long)));
	}
	PROF0(12);
	/* Two kernels.  zmt_brotli_dec4_kernel runs four records per wave (streams without context modelling / block
	 * switching; what it hands over -- status 102 -- goes to the general kernel): a quarter of the waves per record, each
	 * twice as slow as a one-record wave.  That wins as soon as the one-record kernel would need more than one round of
	 * its resident waves (8 192 records: 283 ms against 500), and loses below (1 024 records: 308 ms against 151), so
	 * the batch size chooses.  bdec_variant: 0 = that rule, 1 = the general kernel alone, 2 = dec4 first whatever the size */
	u32 want = 0xFFFFFFFFu;
	const bool use4 = h->bdec_variant == 2 || (h->bdec_variant == 0 && nrec > (size_t)h->bdec_waves);
	if (use4 && h->profile != 7) {
		hipLaunchKernelGGL(zmt_brotli_dec4_kernel, dim3((unsigned)((nrec + B4_NG - 1) / B4_NG)), dim3(64), 0, h->st[s],
				   (const u8 *)d_stream, d_rec_off, d_rec_len, (u32)nrec, (u8 *)d_out, d_out_off, d_out_cap,
				   d_out_len, d_status);
		want = 102u;
	}
	if (h->profile == 7)
		hipLaunchKernelGGL(zmt_brotli_dec_kernel_prof, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_stream,
				   d_rec_off, d_rec_len, (u32)nrec, (u8 *)d_out, d_out_off, d_out_cap, d_out_len,
				   d_status, (u8 *)h->scratch[1][s], (const u8 *)h->d_brotli_static, want, h->d_prof);
	else
		hipLaunchKernelGGL(zmt_brotli_dec_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_stream,
				   d_rec_off, d_rec_len, (u32)nrec, (u8 *)d_out, d_out_off, d_out_cap, d_out_len,
				   d_status, (u8 *)h->scratch[1][s], (const u8 *)h->d_brotli_static, want);
	PROF1(12);
	CK(hipGetLastError());
	return GPUMT_OK;
}

/* ---- snappy-mt (snappy.hip): persistent waves, one record at a time ---- */
size_t gpumt_snappy_slot_stride(size_t chunk)
{
	return (16 + 32 + chunk + chunk / 6 + 255) & ~(size_t)255; /* header + snappy_max_compressed_length */
}

static int snappy_waves(gpumt_ctx *h, int *cache, const void *kernel, const char *what)
{
	if (!*cache) {
		int per_cu = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess)
			per_cu = 0;
		*cache = (per_cu > 0 ? per_cu : 8) * (h->num_cus > 0 ? h->num_cus : 256);
		if (getenv("GPUMT_VERBOSE"))
			fprintf(stderr, "gpumt: snappy %s grid %d waves\n", what, *cache);
	}
	return *cache;
}

int gpumt_snappy_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				size_t slot_stride, uint32_t *d_rec_len, int s)
{
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || slot_stride < gpumt_snappy_slot_stride(chunk))
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	const size_t nrec = gpumt_lz4_record_count(n, chunk);
	if (nrec > 0x7FFFFFFFu)
		return GPUMT_E_ARG;
	const int waves = snappy_waves(h, &h->senc_waves, (const void *)zmt_snappy_enc_kernel, "encoder");
	const unsigned grid = (unsigned)(nrec < (size_t)waves ? nrec : (size_t)waves);
	hipLaunchKernelGGL(zmt_snappy_enc_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_in, (u64)n,
			   (u32)chunk, (u32)nrec, (u8 *)d_slots, (u64)slot_stride, d_rec_len);
	CK(hipGetLastError());
	return GPUMT_OK;
}

int gpumt_snappy_decompress_batch(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
				  const uint32_t *d_rec_len, size_t nrec, void *d_out,
				  const uint64_t *d_out_off, const uint32_t *d_out_cap,
				  uint32_t *d_out_len, uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	if (h->sdec_variant == 1) {
		const int waves = snappy_waves(h, &h->sdec2_waves, (const void *)zmt_snappy_dec2_kernel, "decoder (batched)");
		const unsigned grid = (unsigned)(nrec < (size_t)waves ? nrec : (size_t)waves);
		hipLaunchKernelGGL(zmt_snappy_dec2_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_stream, d_rec_off,
				   d_rec_len, (u32)nrec, (u8 *)d_out, d_out_off, d_out_cap, d_out_len, d_status);
		CK(hipGetLastError());
		return GPUMT_OK;
	}
	const int waves = snappy_waves(h, &h->sdec_waves, (const void *)zmt_snappy_dec_kernel, "decoder");
	const unsigned grid = (unsigned)(nrec < (size_t)waves ? nrec : (size_t)waves);
	hipLaunchKernelGGL(zmt_snappy_dec_kernel, dim3(grid), dim3(64), 0, h->st[s], (const u8 *)d_stream, d_rec_off,
			   d_rec_len, (u32)nrec, (u8 *)d_out, d_out_off, d_out_cap, d_out_len, d_status);
	CK(hipGetLastError());
	return GPUMT_OK;
}

/* read (and clear) the phase counters of the profiling decoder; returns count written */
int gpumt_debug_counters(gpumt_ctx *h, unsigned long long *dst, int n)
{
	if (!h || !dst || n < 16)
		return GPUMT_E_ARG;
	if (use(h))
		return GPUMT_E_HIP;
	if (!h->d_prof) {
		CK(hipMalloc((void **)&h->d_prof, 16 * sizeof(unsigned long long)));
		CK(hipMemset(h->d_prof, 0, 16 * sizeof(unsigned long long)));
	}
	CK(hipDeviceSynchronize());
	CK(hipMemcpy(dst, h->d_prof, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	CK(hipMemset(h->d_prof, 0, 16 * sizeof(unsigned long long)));
	return 16;
}

int gpumt_set_variant(gpumt_ctx *h, const char *what, int variant)
{
	int prev = -1;
	if (!h || !what)
		return -1;
	if (!strcmp(what, "lz4_dec")) {
		prev = h->dec_variant;
		h->dec_variant = variant;
	} else if (!strcmp(what, "lz4_ring")) {
		prev = h->lz4_ring;
		h->lz4_ring = variant;
	} else if (!strcmp(what, "lz4_enc")) {
		prev = h->enc_variant;
		h->enc_variant = variant;
	} else if (!strcmp(what, "lz4_dec_pad")) {
		prev = h->dec_pad;
		h->dec_pad = variant;
	} else if (!strcmp(what, "brotli_dec")) {
		prev = h->bdec_variant;
		h->bdec_variant = variant;
	} else if (!strcmp(what, "debug_free")) {
		prev = h->debug_free;
		h->debug_free = variant;
	} else if (!strcmp(what, "k2x")) {
		prev = h->xflags;
		h->xflags = variant;
	} else if (!strcmp(what, "hc_waves")) {
		prev = h->hc_waves;
		h->hc_waves = variant;
	} else if (!strcmp(what, "zstd_dec")) {
		prev = h->zdec_variant;
		h->zdec_variant = variant;
	} else if (!strcmp(what, "zstd_seq")) {
		prev = h->zseq_variant;
		h->zseq_variant = variant;
	} else if (!strcmp(what, "snappy_dec")) {
		prev = h->sdec_variant;
		h->sdec_variant = variant;
	} else if (!strcmp(what, "profile")) {
		prev = h->profile;
		h->profile = variant;
	}
	return prev;
}

} /* extern "C" */
