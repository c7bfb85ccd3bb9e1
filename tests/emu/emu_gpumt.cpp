/*
 * emu_gpumt.cpp -- TEST HARNESS ONLY: the device boundary (include/gpumt.h) over the fiber emulator,
 * so that the host engines (the .c files of zstdmt_amd/csrc/host: record framing, batching, the three-role
 * pipeline, the reference's callback protocol and error codes) can be exercised by `-m "not gpu"`
 * tests on small inputs.  libzstdmt_emu_host.so = the engines, unchanged, + this file + the kernels
 * compiled as host C++ (tests/emu/Makefile).  It is never loaded by the product: the shipped
 * library binds the same symbols to gpumt.hip, and that one fails without a GPU.
 *
 * Everything is synchronous: "device" memory is host memory filled with a garbage pattern (device
 * allocations are not zeroed either), a kernel launch has finished when the call returns, streams
 * and markers only check their index ranges.  Launches are serialised by one mutex (the emulator
 * keeps per-launch state in globals); the engines launch from one thread anyway.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/gpumt.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

extern "C" {
size_t emu_lz4_slot_stride(size_t chunk);
size_t emu_zstd_slot_stride(size_t chunk);
void emu_lz4_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len);
void emu_lz4hc_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, int level);
void emu_lz4_compact(const u8 *slots, u64 stride, const u32 *rec_len, u32 nrec, u8 *stream, u64 *rec_off);
void emu_lz4_decompress_batch(int variant, const u8 *stream, u64 stream_bytes, const u64 *rec_off,
			      const u32 *rec_len, u32 nrec, u8 *out, u64 out_bytes, const u64 *out_off,
			      u32 *out_len, u32 *status);
void emu_zstd_decompress_batch(const u8 *stream, u64 stream_bytes, const u64 *rec_off, const u32 *rec_len,
			       u32 nrec, u8 *out, const u64 *out_off, u32 *out_len, u32 *status);
void emu_zstd_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid);
void emu_zstd_compress_batch_level(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid, int level);
void emu_brotli_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid);
void emu_brotli_compress_batch_level(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid, int level);
void emu_brotli_decompress_batch(const u8 *stream, const u64 *rec_off, const u32 *rec_len, u32 nrec, u8 *out,
				 const u64 *out_off, const u32 *out_cap, u32 *out_len, u32 *status,
				 const u8 *blob, u32 grid);
size_t emu_snappy_slot_stride(size_t chunk);
void emu_snappy_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid);
void emu_snappy_decompress_batch(const u8 *stream, const u64 *rec_off, const u32 *rec_len, u32 nrec, u8 *out,
				 const u64 *out_off, const u32 *out_cap, u32 *out_len, u32 *status, u32 grid);
extern const unsigned char zmt_brotli_static[], zmt_brotli_static_end[];
}

struct gpumt_ctx {
	int device;
	int lz4_dec_variant;
	unsigned long long launches; /* kernels-batches run on this "device" (tests read it) */
};

static std::mutex g_launch;
static unsigned long long g_opened, g_launches[16];

#define NSTREAM 16
#define STREAM_OK(s) ((s) >= 0 && (s) < NSTREAM)
#define GRID 3 /* persistent-grid waves for the block coders: fewer than blocks, so the grid loops run */

static int emu_devices(void)
{
	const char *e = getenv("EMU_GPUMT_DEVICES");
	int n = e && *e ? atoi(e) : 2;
	return n < 1 ? 1 : n > 16 ? 16 : n;
}

extern "C" {

int gpumt_device_count(void) { return emu_devices(); }

int gpumt_open(int device, gpumt_ctx **out)
{
	if (!out)
		return GPUMT_E_ARG;
	if (device == GPUMT_DEVICE_DEFAULT) {
		const char *e = getenv("GPUMT_DEVICE");
		device = e && *e ? atoi(e) : 0;
	}
	if (device < 0 || device >= emu_devices())
		return GPUMT_E_HIP;
	gpumt_ctx *h = (gpumt_ctx *)calloc(1, sizeof *h);
	if (!h)
		return GPUMT_E_HIP;
	h->device = device;
	g_opened++;
	*out = h;
	return GPUMT_OK;
}

void gpumt_close(gpumt_ctx *h) { free(h); }
const char *gpumt_last_error(gpumt_ctx *) { return "emulated device"; }
const char *gpumt_device_name(gpumt_ctx *) { return "fiber emulator"; }
int gpumt_host_node(gpumt_ctx *) { return -1; }

/* test hooks: contexts opened so far, launches per emulated device */
unsigned long long emu_gpumt_opened(void) { return g_opened; }
unsigned long long emu_gpumt_launches(int device) { return device >= 0 && device < 16 ? g_launches[device] : 0; }

static void *garbage_alloc(size_t bytes)
{
	void *p = malloc(bytes + 512);
	if (p)
		memset(p, 0xA5, bytes + 512);
	return p;
}
void *gpumt_malloc(gpumt_ctx *h, size_t bytes) { return h ? garbage_alloc(bytes) : nullptr; }
void gpumt_free(gpumt_ctx *, void *p) { free(p); }
void *gpumt_host_alloc(gpumt_ctx *h, size_t bytes)
{
	void *p = nullptr;
	if (!h || posix_memalign(&p, 4096, bytes + 512))
		return nullptr;
	memset(p, 0x5A, bytes + 512);
	return p;
}
void gpumt_host_free(gpumt_ctx *, void *p) { free(p); }
int gpumt_host_register(gpumt_ctx *h, void *p, size_t bytes) { return (h && p && bytes) ? GPUMT_OK : GPUMT_E_ARG; }
int gpumt_host_unregister(gpumt_ctx *h, void *p) { return (h && p) ? GPUMT_OK : GPUMT_E_ARG; }
size_t gpumt_trim_caches(gpumt_ctx *) { return 0; }
unsigned long gpumt_debug_free_busy(void) { return 0; }

static int copy(gpumt_ctx *h, void *dst, const void *src, size_t n, int s)
{
	if (!h || !STREAM_OK(s) || (n && (!dst || !src)))
		return GPUMT_E_ARG;
	memmove(dst, src, n);
	return GPUMT_OK;
}
int gpumt_memcpy_h2d(gpumt_ctx *h, void *dst, const void *src, size_t n, int s) { return copy(h, dst, src, n, s); }
int gpumt_memcpy_d2h(gpumt_ctx *h, void *dst, const void *src, size_t n, int s) { return copy(h, dst, src, n, s); }
int gpumt_memcpy_d2d(gpumt_ctx *h, void *dst, const void *src, size_t n, int s) { return copy(h, dst, src, n, s); }

int gpumt_push_host(gpumt_ctx *h, void *dst, const void *src, size_t n, const uint64_t *d_n, int s)
{
	if (!h || !STREAM_OK(s) || !dst || !src || (((uintptr_t)dst | (uintptr_t)src) & 15))
		return GPUMT_E_ARG;
	if (d_n && *d_n < n)
		n = (size_t)*d_n;
	memcpy(dst, src, n);
	return GPUMT_OK;
}

int gpumt_memset(gpumt_ctx *h, void *dst, int byte, size_t n, int s)
{
	if (!h || !STREAM_OK(s) || !dst)
		return GPUMT_E_ARG;
	memset(dst, byte, n);
	return GPUMT_OK;
}
int gpumt_stream_sync(gpumt_ctx *h, int s) { return h && STREAM_OK(s) ? GPUMT_OK : GPUMT_E_ARG; }
int gpumt_device_sync(gpumt_ctx *h) { return h ? GPUMT_OK : GPUMT_E_ARG; }
int gpumt_stream_wait(gpumt_ctx *h, int w, int s) { return h && STREAM_OK(w) && STREAM_OK(s) ? GPUMT_OK : GPUMT_E_ARG; }
int gpumt_mark(gpumt_ctx *h, int id, int s)
{
	return h && id >= 0 && id < GPUMT_NMARKS && STREAM_OK(s) ? GPUMT_OK : GPUMT_E_ARG;
}
int gpumt_mark_sync(gpumt_ctx *h, int id) { return h && id >= 0 && id < GPUMT_NMARKS ? GPUMT_OK : GPUMT_E_ARG; }

size_t gpumt_lz4_slot_stride(size_t chunk) { return emu_lz4_slot_stride(chunk); }
size_t gpumt_zstd_slot_stride(size_t chunk) { return emu_zstd_slot_stride(chunk); }
size_t gpumt_lz4_record_count(size_t n, size_t chunk) { return n ? (n + chunk - 1) / chunk : 1; }
int gpumt_lz4_level_supported(int level) { return level >= 1 && level <= 12; }

static void count(gpumt_ctx *h)
{
	h->launches++;
	g_launches[h->device & 15]++;
}

int gpumt_lz4_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				   size_t slot_stride, uint32_t *d_rec_len, int level, int s)
{
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || slot_stride < gpumt_lz4_slot_stride(chunk) ||
	    !gpumt_lz4_level_supported(level))
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	if (level <= 2)
		emu_lz4_compress_batch((const u8 *)d_in, n, (u32)chunk, (u8 *)d_slots, slot_stride, d_rec_len);
	else
		emu_lz4hc_compress_batch((const u8 *)d_in, n, (u32)chunk, (u8 *)d_slots, slot_stride, d_rec_len, level);
	return GPUMT_OK;
}

int gpumt_lz4_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
			     size_t slot_stride, uint32_t *d_rec_len, int s)
{
	return gpumt_lz4_compress_batch_level(h, d_in, n, chunk, d_slots, slot_stride, d_rec_len, 1, s);
}

int gpumt_lz4_compact(gpumt_ctx *h, const void *d_slots, size_t slot_stride, const uint32_t *d_rec_len,
		      size_t nrec, void *d_stream, uint64_t *d_rec_off, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_lz4_compact((const u8 *)d_slots, slot_stride, d_rec_len, (u32)nrec, (u8 *)d_stream, d_rec_off);
	return GPUMT_OK;
}

int gpumt_lz4_decompress_batch(gpumt_ctx *h, const void *d_stream, size_t stream_bytes,
			       const uint64_t *d_rec_off, const uint32_t *d_rec_len, size_t nrec, void *d_out,
			       size_t out_bytes, const uint64_t *d_out_off, uint32_t *d_out_len,
			       uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_lz4_decompress_batch(h->lz4_dec_variant == 1 ? 1 : 0, (const u8 *)d_stream, stream_bytes, d_rec_off,
				 d_rec_len, (u32)nrec, (u8 *)d_out, out_bytes, d_out_off, d_out_len, d_status);
	return GPUMT_OK;
}

int gpumt_zstd_level_tier(int level) { return level <= 2 ? 0 : level <= 9 ? 1 : 2; }
int gpumt_zstd_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				    size_t slot_stride, uint32_t *d_rec_len, int level, int s)
{
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || slot_stride < gpumt_zstd_slot_stride(chunk) ||
	    level < 1 || level > 22)
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_zstd_compress_batch_level((const u8 *)d_in, n, (u32)chunk, (u8 *)d_slots, slot_stride, d_rec_len, GRID, level);
	return GPUMT_OK;
}
int gpumt_zstd_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
			      size_t slot_stride, uint32_t *d_rec_len, int s)
{
	return gpumt_zstd_compress_batch_level(h, d_in, n, chunk, d_slots, slot_stride, d_rec_len, 1, s);
}

int gpumt_zstd_decompress_batch(gpumt_ctx *h, const void *d_stream, size_t stream_bytes,
				const uint64_t *d_rec_off, const uint32_t *d_rec_len, size_t nrec, void *d_out,
				size_t out_bytes, const uint64_t *d_out_off, uint32_t *d_out_len,
				uint32_t *d_status, int s)
{
	(void)out_bytes;
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_zstd_decompress_batch((const u8 *)d_stream, stream_bytes, d_rec_off, d_rec_len, (u32)nrec, (u8 *)d_out,
				  d_out_off, d_out_len, d_status);
	return GPUMT_OK;
}

int gpumt_brotli_level_tier(int level) { return level <= 3 ? 0 : level <= 8 ? 1 : 2; }
int gpumt_brotli_compress_batch_level(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				      size_t slot_stride, uint32_t *d_rec_len, int level, int s)
{
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || slot_stride < gpumt_zstd_slot_stride(chunk) ||
	    level < 0 || level > 11)
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_brotli_compress_batch_level((const u8 *)d_in, n, (u32)chunk, (u8 *)d_slots, slot_stride, d_rec_len, GRID, level);
	return GPUMT_OK;
}
int gpumt_brotli_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				size_t slot_stride, uint32_t *d_rec_len, int s)
{
	return gpumt_brotli_compress_batch_level(h, d_in, n, chunk, d_slots, slot_stride, d_rec_len, 1, s);
}

int gpumt_brotli_decompress_batch(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
				  const uint32_t *d_rec_len, size_t nrec, void *d_out, const uint64_t *d_out_off,
				  const uint32_t *d_out_cap, uint32_t *d_out_len, uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_brotli_decompress_batch((const u8 *)d_stream, d_rec_off, d_rec_len, (u32)nrec, (u8 *)d_out, d_out_off,
				    d_out_cap, d_out_len, d_status, zmt_brotli_static, GRID);
	return GPUMT_OK;
}

size_t gpumt_snappy_slot_stride(size_t chunk) { return emu_snappy_slot_stride(chunk); }

int gpumt_snappy_compress_batch(gpumt_ctx *h, const void *d_in, size_t n, size_t chunk, void *d_slots,
				size_t slot_stride, uint32_t *d_rec_len, int s)
{
	if (!h || !STREAM_OK(s) || chunk == 0 || chunk > 0x40000000u || slot_stride < gpumt_snappy_slot_stride(chunk))
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_snappy_compress_batch((const u8 *)d_in, n, (u32)chunk, (u8 *)d_slots, slot_stride, d_rec_len, GRID);
	return GPUMT_OK;
}

int gpumt_snappy_decompress_batch(gpumt_ctx *h, const void *d_stream, const uint64_t *d_rec_off,
				  const uint32_t *d_rec_len, size_t nrec, void *d_out, const uint64_t *d_out_off,
				  const uint32_t *d_out_cap, uint32_t *d_out_len, uint32_t *d_status, int s)
{
	if (!h || !STREAM_OK(s) || nrec == 0 || nrec > 0x3FFFFFFFu)
		return GPUMT_E_ARG;
	std::lock_guard<std::mutex> lk(g_launch);
	count(h);
	emu_snappy_decompress_batch((const u8 *)d_stream, d_rec_off, d_rec_len, (u32)nrec, (u8 *)d_out, d_out_off,
				    d_out_cap, d_out_len, d_status, GRID);
	return GPUMT_OK;
}

int gpumt_set_variant(gpumt_ctx *h, const char *what, int variant)
{
	if (!h || !what)
		return -1;
	if (!strcmp(what, "lz4_dec")) {
		const int prev = h->lz4_dec_variant;
		h->lz4_dec_variant = variant;
		return prev;
	}
	/* ("snappy_dec": the emulated launch reads EMU_SNAPPY_DEC from the environment, see emu_api.cpp) */
	return 0;
}

} /* extern "C" */
