// The host engine's per-batch call pattern (H2D on a copy stream, kernel on the slot's stream, results
// back on a third) with batches launched two ahead: do the batches overlap on the device?
// (developer microbenchmark; variants bisect what serialises them)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long ticks)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		__builtin_amdgcn_s_sleep(32);
}
__global__ void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = src[i];
}
static hipStream_t st[16];
static void stream_wait(int waiter, int signaler, int keep_events)
{
	hipEvent_t ev;
	CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	CK(hipEventRecord(ev, st[signaler]));
	CK(hipStreamWaitEvent(st[waiter], ev, 0));
	if (!keep_events)
		CK(hipEventDestroy(ev));
}
int main(int argc, char **argv)
{
	const int variant = argc > 1 ? atoi(argv[1]) : 0;
	const size_t n = (size_t)256 << 20, nout = (size_t)144 << 20;
	const int NS = 4, NB = 12;
	void *h_in[NS], *h_out[NS], *h_meta[NS], *d_in[NS], *d_out[NS];
	for (int i = 0; i < NS; i++) {
		CK(hipHostMalloc(&h_in[i], n, hipHostMallocPortable));
		CK(hipHostMalloc(&h_out[i], n, hipHostMallocPortable));
		CK(hipHostMalloc(&h_meta[i], 65536, hipHostMallocPortable));
		memset(h_in[i], 1, n);
		CK(hipMalloc(&d_in[i], n));
		CK(hipMalloc(&d_out[i], n));
	}
	for (int i = 0; i < 16; i++)
		CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
	hipEvent_t base, mark[NS], h0[NB], h1[NB], k0[NB], k1[NB];
	CK(hipEventCreate(&base));
	for (int i = 0; i < NS; i++)
		CK(hipEventCreate(&mark[i]));
	for (int i = 0; i < NB; i++) {
		CK(hipEventCreate(&h0[i])); CK(hipEventCreate(&h1[i])); CK(hipEventCreate(&k0[i])); CK(hipEventCreate(&k1[i]));
	}
	const unsigned long long ticks = 25ull * 100000;
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(base, st[1]));
	int done = 0;
	for (int b = 0; b < NB || done < NB;) {
		if (b < NB) {
			const int s = b % NS, ks = 4 + s;
			if (variant & 32) {
				CK(hipEventRecord(h0[b], st[ks]));
				hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, st[ks], (const uint4 *)h_in[s], (uint4 *)d_in[s], n / 16);
				CK(hipEventRecord(h1[b], st[ks]));
			} else {
			CK(hipEventRecord(h0[b], st[1]));
			CK(hipMemcpyAsync(d_in[s], h_in[s], n, hipMemcpyHostToDevice, st[1]));
			CK(hipEventRecord(h1[b], st[1]));
			}
			if (variant & 32) {
			} else if (variant & 1) {
				CK(hipStreamWaitEvent(st[ks], h1[b], 0));
			} else
				stream_wait(ks, 1, variant & 2);
			CK(hipEventRecord(k0[b], st[ks]));
			hipLaunchKernelGGL(spin_kernel, dim3(2048), dim3(64), 0, st[ks], ticks);
			CK(hipEventRecord(k1[b], st[ks]));
			if (variant & 16)
				hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, st[ks], (const uint4 *)d_out[s], (uint4 *)h_out[s], nout / 16);
			if (variant & 1)
				CK(hipStreamWaitEvent(st[2], k1[b], 0));
			else
				stream_wait(2, ks, variant & 2);
			CK(hipMemcpyAsync(h_meta[s], d_out[s], 65536, hipMemcpyDeviceToHost, st[2]));
			CK(hipEventRecord(mark[s], st[2]));
			b++;
		}
		while (done < b && (b >= NB || b - done >= NS - 1)) {
			const int s = done % NS;
			CK(hipEventSynchronize(mark[s]));
			if (!(variant & (4 | 16))) {
				CK(hipMemcpyAsync(h_out[s], d_out[s], nout, hipMemcpyDeviceToHost, st[3]));
				if (variant & 8) {
					CK(hipEventRecord(mark[s], st[3]));
					CK(hipEventSynchronize(mark[s]));
				} else
					CK(hipStreamSynchronize(st[3]));
			}
			done++;
		}
	}
	CK(hipDeviceSynchronize());
	printf("variant %d (1 = wait on timing events instead of fresh ones, 2 = events not destroyed, 4 = no result copy, 8 = event sync instead of stream sync, 16 = results leave through a copy kernel on the slot's stream, 32 = input arrives through one)\n", variant);
	for (int b = 0; b < NB; b++) {
		float a, c, d, e;
		CK(hipEventElapsedTime(&a, base, h0[b])); CK(hipEventElapsedTime(&c, base, h1[b]));
		CK(hipEventElapsedTime(&d, base, k0[b])); CK(hipEventElapsedTime(&e, base, k1[b]));
		printf("  batch %2d: H2D %8.2f..%8.2f  kernel %8.2f..%8.2f\n", b, a, c, d, e);
	}
	return 0;
}
