// Does a host<->device copy on one stream overlap a long kernel on another?  (developer microbenchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned *sink)
{
	__shared__ unsigned pad[2856];
	pad[threadIdx.x] = threadIdx.x;
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		__builtin_amdgcn_s_sleep(32);
	if (sink && pad[threadIdx.x] == 0xFFFFFFFFu)
		*sink = 1;
}
__global__ void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = src[i];
}

int main(int argc, char **argv)
{
	const size_t n = (size_t)256 << 20;
	const int blocks = argc > 1 ? atoi(argv[1]) : 2048;
	void *h_in, *h_out, *d_a, *d_b;
	CK(hipHostMalloc(&h_in, n, hipHostMallocPortable));
	CK(hipHostMalloc(&h_out, n, hipHostMallocPortable));
	memset(h_in, 1, n);
	memset(h_out, 0, n);
	CK(hipMalloc(&d_a, n));
	CK(hipMalloc(&d_b, n));
	hipStream_t sk, sc, sd;
	CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
	hipEvent_t e0, k1, c0, c1, x0, x1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&k1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
	CK(hipEventCreate(&x0)); CK(hipEventCreate(&x1));
	const unsigned long long ticks = 25ull * 100000; /* 100 MHz wall clock: 25 ms */
	for (int mode = 0; mode < 5; mode++) {
		for (int rep = 0; rep < 2; rep++) {
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(e0, sk));
			hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, sk, ticks, (unsigned *)nullptr);
			CK(hipEventRecord(k1, sk));
			CK(hipEventRecord(c0, sc));
			if (mode == 0)
				CK(hipMemcpyAsync(d_a, h_in, n, hipMemcpyHostToDevice, sc));
			else if (mode == 1)
				CK(hipMemcpyAsync(h_out, d_b, n, hipMemcpyDeviceToHost, sc));
			else if (mode == 2)
				hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, sc, (const uint4 *)h_in, (uint4 *)d_a, n / 16);
			else if (mode == 3)
				hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, sc, (const uint4 *)d_b, (uint4 *)h_out, n / 16);
			else {
				CK(hipMemcpyAsync(d_a, h_in, n, hipMemcpyHostToDevice, sc));
				CK(hipEventRecord(x0, sd));
				CK(hipMemcpyAsync(h_out, d_b, n, hipMemcpyDeviceToHost, sd));
				CK(hipEventRecord(x1, sd));
			}
			CK(hipEventRecord(c1, sc));
			CK(hipDeviceSynchronize());
			float tk, t0c, t1c, tx0 = 0, tx1 = 0;
			CK(hipEventElapsedTime(&tk, e0, k1));
			CK(hipEventElapsedTime(&t0c, e0, c0));
			CK(hipEventElapsedTime(&t1c, e0, c1));
			if (mode == 4) { CK(hipEventElapsedTime(&tx0, e0, x0)); CK(hipEventElapsedTime(&tx1, e0, x1)); }
			static const char *nm[] = {"hipMemcpyAsync H2D", "hipMemcpyAsync D2H", "copy kernel H2D (reads pinned)", "copy kernel D2H (writes pinned)", "H2D + D2H on two streams"};
			if (rep)
				printf("%-34s kernel(%d blocks) 0..%.2f ms | copy %.2f..%.2f ms (%.1f GB/s)%s\n", nm[mode], blocks, tk, t0c, t1c,
				       n / 1e6 / (t1c - t0c), mode == 4 ? "" : "");
			if (rep && mode == 4)
				printf("%-34s   second copy %.2f..%.2f ms\n", "", tx0, tx1);
		}
	}
	return 0;
}
