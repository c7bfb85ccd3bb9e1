// LDS allocation granularity on this device: resident 64-thread workgroups per CU by dynamic LDS size
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_occupancy.hip -o tools/ubench/lds_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) k(unsigned *o) { extern __shared__ unsigned s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); o[threadIdx.x] = s[63 - threadIdx.x]; }
int main()
{
	int last = -1;
	for (int b = 4096; b <= 20480; b += 4) {
		int n = 0;
		hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, b);
		if (n != last) { printf("LDS %6d B per workgroup -> %2d workgroups per CU (%d x %d = %d)\n", b, n, n, b, n * b); last = n; }
	}
	return 0;
}
