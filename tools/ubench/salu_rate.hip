// What does a CU sustain in scalar vs vector ALU wave-instructions per cycle, by waves per CU?
// (developer microbenchmark: several of the codec kernels are ~50-70 % SALU and level off near
//  1 wave-instruction per cycle per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE> __global__ void __launch_bounds__(64) rate_kernel(unsigned iters, unsigned *sink)
{
	unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
	unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
	for (unsigned i = 0; i < iters; i++) {
		if (MODE == 0 || MODE == 2) /* 8 scalar ops, 4 independent chains */
			asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
				     "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
				     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
		if (MODE == 1 || MODE == 2) /* 8 vector ops */
			asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
				     "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
				     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
		if (MODE == 3) /* a dependent scalar chain with a compare + branch, like a decoder's control flow */
			asm volatile("s_add_u32 %0, %0, 1\n s_lshr_b32 %1, %0, 3\n s_and_b32 %2, %1, 7\n s_add_u32 %3, %3, %2\n"
				     "s_cmp_eq_u32 %2, 99\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, %3\n s_xor_b32 %1, %1, %0\n1:\n"
				     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
	}
	if (s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3 == 0xDEADBEEF)
		*sink = 1;
}

int main()
{
	unsigned *sink;
	CK(hipMalloc(&sink, 4));
	hipDeviceProp_t p;
	CK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	const double ghz = p.clockRate * 1e-6;
	printf("%d CUs, %.2f GHz\n", cus, ghz);
	fflush(stdout);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const unsigned iters = 20000;
	static const char *nm[] = {"8 SALU (4 chains)", "8 VALU (4 chains)", "8 SALU + 8 VALU", "dependent SALU chain + branch (8)"};
	for (int mode = 0; mode < 4; mode++)
		for (int wpc = 1; wpc <= 32; wpc *= 2) {
			float ms = 0;
			for (int rep = 0; rep < 2; rep++) {
				CK(hipEventRecord(e0));
				switch (mode) {
				case 0: hipLaunchKernelGGL(rate_kernel<0>, dim3(cus * wpc), dim3(64), 0, 0, iters, sink); break;
				case 1: hipLaunchKernelGGL(rate_kernel<1>, dim3(cus * wpc), dim3(64), 0, 0, iters, sink); break;
				case 2: hipLaunchKernelGGL(rate_kernel<2>, dim3(cus * wpc), dim3(64), 0, 0, iters, sink); break;
				default: hipLaunchKernelGGL(rate_kernel<3>, dim3(cus * wpc), dim3(64), 0, 0, iters, sink); break;
				}
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				CK(hipEventElapsedTime(&ms, e0, e1));
			}
			const double n = (mode == 2 ? 16.0 : 8.0) * iters * wpc; /* wave-instructions per CU (loop overhead not counted) */
			printf("%-36s %2d waves/CU: %8.3f ms  %.2f wave-instr/cycle/CU\n", nm[mode], wpc, ms, n / (ms * 1e-3 * ghz * 1e9));
			fflush(stdout);
		}
	return 0;
}
