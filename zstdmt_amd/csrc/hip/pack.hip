/*
 * pack.hip -- the ordered-concatenation side of the MT stream.
 *
 * The reference keeps output order with a done-list flushed under write_mutex
 * (/root/reference/lib/lz4-mt_compress.c:178-205, pt_write).  On the device, order is positional:
 * record i lives in slot i, an exclusive scan of the record lengths gives its byte offset in the
 * stream, and one workgroup per record moves it there (zmt_compact_kernel).  The decompress side
 * needs the mirror image: per-record content sizes (the LE64 the reference reads at payload+6,
 * lib/lz4-mt_decompress.c:333-334) and their scan (zmt_probe_kernel + zmt_scan_kernel).
 */
#include "lz4_common.h"

typedef u32 u32x4 __attribute__((vector_size(16)));

/* out[i] = content size of record i (0 when the frame carries none, e.g. the empty frame) */
extern "C" __global__ void __launch_bounds__(256)
zmt_probe_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
		 const u32 *__restrict__ rec_len, u32 nrec, u32 *__restrict__ out_len)
{
	u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= nrec)
		return;
	const u8 *r = stream + rec_off[i];
	u32 rl = rec_len[i], v = 0;
	if (rl >= 12 + 15 && (r[12 + 4] & 0x08)) {
		u64 cs = ld64u(r + 12 + 6);
		v = cs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)cs;
	}
	out_len[i] = v;
}

/* exclusive scan u32 -> u64, single workgroup of 1024; off[n] = total */
extern "C" __global__ void __launch_bounds__(1024)
zmt_scan_kernel(const u32 *__restrict__ len, u32 n, u64 *__restrict__ off)
{
	__shared__ u64 part[1024];
	const u32 t = threadIdx.x;
	const u32 per = (n + 1023) / 1024;
	const u32 lo = t * per < n ? t * per : n;
	const u32 hi = lo + per < n ? lo + per : n;
	u64 s = 0;
	for (u32 i = lo; i < hi; i++)
		s += len[i];
	part[t] = s;
	__syncthreads();
	/* Hillis-Steele over 1024 partials */
	for (u32 d = 1; d < 1024; d <<= 1) {
		u64 v = (t >= d) ? part[t - d] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	u64 run = part[t] - s;
	for (u32 i = lo; i < hi; i++) {
		off[i] = run;
		run += len[i];
	}
	if (t == 1023)
		off[n] = part[1023];
}

/* record i: slots + i*stride (256-aligned)  ->  stream + off[i] (any alignment) */
extern "C" __global__ void __launch_bounds__(256)
zmt_compact_kernel(const u8 *__restrict__ slots, u64 stride, const u32 *__restrict__ rec_len,
		   const u64 *__restrict__ off, u32 nrec, u8 *__restrict__ stream)
{
	const u32 rec = blockIdx.x;
	if (rec >= nrec)
		return;
	const u8 *s = slots + (u64)rec * stride;
	u8 *d = stream + off[rec];
	const u32 n = rec_len[rec];
	const u32 t = threadIdx.x;
	u32 head = (u32)((16 - ((u64)d & 15)) & 15);
	if (head > n)
		head = n;
	if (t < head)
		d[t] = s[t];
	const u32 nb = (n - head) >> 4;
	for (u32 k = t; k < nb; k += 256) {
		u32x4 v;
		__builtin_memcpy(&v, s + head + 16 * k, 16); /* source misaligned by `head` */
		*(u32x4 *)(d + head + 16 * k) = v;
	}
	const u32 done = head + nb * 16;
	if (t < n - done)
		d[done + t] = s[done + t];
}

/* off[i] = i*chunk, len[i] = bytes of chunk i of an n-byte buffer (n == 0: one empty chunk) */
extern "C" __global__ void __launch_bounds__(256)
zmt_iota_kernel(u64 *__restrict__ off, u32 *__restrict__ len, u64 n, u64 chunk, u32 nrec)
{
	u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= nrec)
		return;
	u64 o = (u64)i * chunk;
	off[i] = o;
	len[i] = (u32)(n - o < chunk ? n - o : chunk);
}

/*
 * Results leave the device through a kernel that stores into pinned host memory (dst is the host
 * buffer's device-visible address), on the stream of the kernels that produced them.  Two things
 * this buys over hipMemcpyAsync after a host-side wait (measured, tools/ubench/pipe_overlap.hip): the
 * byte count can stay on the device (n_dev: e.g. the total of a scan), so the host does not have to
 * wake up between the kernels and the copy; and batches launched on different streams really
 * overlap -- with the runtime's copy issued from the completing thread the next batches' input
 * copies did not start before it, i.e. the pipeline ran one batch at a time.
 * src and dst 16-byte aligned; bytes = min(n, *n_dev) when n_dev is given.
 */
extern "C" __global__ void __launch_bounds__(256)
zmt_push_host_kernel(const u8 *__restrict__ src, u8 *__restrict__ dst, u64 n, const u64 *__restrict__ n_dev)
{
	if (n_dev && *n_dev < n)
		n = *n_dev;
	const u64 nv = n >> 4;
	const u32x4 *s4 = (const u32x4 *)src;
	u32x4 *d4 = (u32x4 *)dst;
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nv; i += (u64)gridDim.x * 256)
		d4[i] = s4[i];
	if (blockIdx.x == 0 && threadIdx.x < (u32)(n & 15))
		dst[(nv << 4) + threadIdx.x] = src[(nv << 4) + threadIdx.x];
}
