// microbenchmark (round 3): what the LZ4 copy stage may spend on its stores, and whether LDS-DMA behaves.
//   A  misaligned ds_write_b64 / ds_write_b128 by number of ACTIVE lanes (is a replay paid per active lane?)
//   B  sequence-shaped stores straight to global memory (lane l writes 8 / 16 bytes at 12 l + 3): GB/s of the chip
//   C  global_load_lds_dwordx4: 1 KiB per wave instruction from per-lane global addresses, checked and timed
// hipcc --offload-arch=gfx950 -O3 tools/ubench/store_cost.hip -o tools/ubench/store_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
typedef u32 v2u __attribute__((ext_vector_type(2)));
typedef u32 v4u __attribute__((ext_vector_type(4)));

template <int W>
__global__ void __launch_bounds__(1024) k_masked(u32 *out, int iters, int active, int pat, u64 *cycles)
{
	__shared__ __attribute__((aligned(16))) u8 lds[16384 + 4096];
	const int tid = threadIdx.x, lane = tid & 63;
	for (int i = tid; i < (int)sizeof(lds); i += blockDim.x)
		lds[i] = (u8)i;
	__syncthreads();
	u32 base = (u32)(size_t)lds + (u32)(tid >> 6) * 64;
	u32 wv = 0x01020304u * (u32)(tid + 1);
	v2u w2 = {wv, ~wv};
	v4u w4 = {wv, ~wv, wv * 3, wv * 5};
	// active lanes spread over the wave (every 64/active-th lane)
	const bool on = (lane % (64 / active)) == 0;
	u64 t0 = clock64();
	for (int it = 0; it < iters; it++) {
		u32 off = pat == 0 ? (u32)lane * 13 + 3 + (it & 7) : (u32)lane * (W == 8 ? 8 : 16);
		u32 addr = base + off;
		if (on) {
			if (W == 8)
				asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:2048\n\tds_write_b64 %0, %1 offset:4096\n\t"
					     "ds_write_b64 %0, %1 offset:6144\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(w2) : "memory");
			else
				asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:2048\n\tds_write_b128 %0, %1 offset:4096\n\t"
					     "ds_write_b128 %0, %1 offset:6144\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(w4) : "memory");
		}
	}
	u64 t1 = clock64();
	__syncthreads();
	if (lane == 0)
		cycles[tid >> 6] = t1 - t0;
	out[tid] = lds[tid];
}

// B: every wave owns a 64 KiB region and writes it front to back in "sequences"
template <int W, int MIS>
__global__ void __launch_bounds__(256) k_gstore(u8 *dst, int reps)
{
	const int lane = threadIdx.x & 63;
	const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
	u8 *p = dst + wave * 65536;
	const u32 stride = MIS ? (W == 8 ? 12u : 24u) : (u32)W;
	const u32 span = 64 * stride;
	u64 v = wave * 0x9E3779B97F4A7C15ull + lane;
	for (int r = 0; r < reps; r++)
		for (u32 o = 0; o + span + 32 <= 65536; o += span) {
			u8 *d = p + o + lane * stride + (MIS ? 3 : 0);
			if (W == 8)
				__builtin_memcpy(d, &v, 8);
			else {
				__builtin_memcpy(d, &v, 8);
				__builtin_memcpy(d + 8, &v, 8);
			}
			v += 64;
		}
}

// C: LDS-DMA.  Each wave copies its 64 KiB region 1 KiB at a time into LDS and sums it.
__global__ void __launch_bounds__(256) k_dma(const u8 *src, u32 *sums, int use_dma, int reps)
{
	__shared__ __attribute__((aligned(16))) u8 lds[4][2][1024];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const u64 wave = (u64)blockIdx.x * 4 + w;
	const u8 *p = src + wave * 65536;
	u32 acc = 0;
	for (int r = 0; r < reps; r++)
		for (u32 o = 0; o < 65536; o += 1024) {
			u8 *stage = lds[w][(o >> 10) & 1];
			if (use_dma) {
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + o + 16 * lane),
								 (__attribute__((address_space(3))) void *)stage, 16, 0, 0);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			} else {
				v4u x = *(const v4u *)(p + o + 16 * lane);
				*(v4u *)(stage + 16 * lane) = x;
			}
			__builtin_amdgcn_wave_barrier();
			// read it back rotated by one lane so that the value really comes through LDS
			v4u y = *(const v4u *)(stage + 16 * ((lane + 1) & 63));
			acc += y.x ^ y.y ^ y.z ^ y.w;
			__builtin_amdgcn_wave_barrier();
		}
	sums[wave * 64 + lane] = acc;
}

int main()
{
	u32 *dout; u64 *dcy;
	hipMalloc(&dout, 4096 * 4); hipMalloc(&dcy, 8 * 64);
	printf("A: misaligned LDS stores by active lanes (cycles per wave-instruction on the CU, 16 waves/CU)\n");
	for (int W : {8, 16})
		for (int pat : {0, 1}) {
			printf("  ds_write_b%-3d %s:", W * 8, pat == 0 ? "lane*13+3 (misaligned)" : "aligned               ");
			for (int active : {64, 32, 16, 8, 4, 1}) {
				const int waves = 16, iters = 2000;
				if (W == 8) hipLaunchKernelGGL(k_masked<8>, dim3(1), dim3(64 * waves), 0, 0, dout, iters, active, pat, dcy);
				else hipLaunchKernelGGL(k_masked<16>, dim3(1), dim3(64 * waves), 0, 0, dout, iters, active, pat, dcy);
				hipDeviceSynchronize();
				u64 cy[16], mx = 0;
				hipMemcpy(cy, dcy, 8 * waves, hipMemcpyDeviceToHost);
				for (int i = 0; i < waves; i++) mx = cy[i] > mx ? cy[i] : mx;
				printf("  %2d lanes %6.1f", active, (double)mx / ((double)iters * 4 * waves));
			}
			printf("\n");
		}

	printf("B: sequence-shaped global stores, 24 waves/CU resident (6144 waves x 64 KiB regions)\n");
	const u64 nw = 6144 * 4;
	u8 *dst; hipMalloc(&dst, nw * 65536 + 4096);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int cfg = 0; cfg < 4; cfg++) {
		const int reps = 2; float ms = 0;
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			if (cfg == 0) hipLaunchKernelGGL((k_gstore<16, 0>), dim3(nw / 4), dim3(256), 0, 0, dst, reps);
			if (cfg == 1) hipLaunchKernelGGL((k_gstore<8, 0>), dim3(nw / 4), dim3(256), 0, 0, dst, reps);
			if (cfg == 2) hipLaunchKernelGGL((k_gstore<8, 1>), dim3(nw / 4), dim3(256), 0, 0, dst, reps);
			if (cfg == 3) hipLaunchKernelGGL((k_gstore<16, 1>), dim3(nw / 4), dim3(256), 0, 0, dst, reps);
			hipEventRecord(e1); hipEventSynchronize(e1);
			hipEventElapsedTime(&ms, e0, e1);
		}
		const char *nm[] = {"aligned 16 B/lane (coalesced)", "aligned 8 B/lane", "8 B at 12 l + 3 (8 of 12 bytes)", "16 B at 24 l + 3 (16 of 24 bytes)"};
		double stores = (double)nw * reps * (65536.0 / (64.0 * (cfg == 0 ? 16 : cfg == 1 ? 8 : cfg == 2 ? 12 : 24)));
		printf("  %-36s %7.3f ms  %8.1f GB/s of region covered  %6.1f cycles per wave-store per CU\n", nm[cfg], ms,
		       (double)nw * 65536 * reps / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / stores);
	}

	printf("C: global -> LDS staging, 1 KiB per wave step (6144 x 4 waves x 64 KiB)\n");
	std::vector<u8> hsrc(nw * 65536);
	for (size_t i = 0; i < hsrc.size(); i++) hsrc[i] = (u8)(i * 2654435761u >> 11);
	hipMemcpy(dst, hsrc.data(), hsrc.size(), hipMemcpyHostToDevice);
	u32 *sums; hipMalloc(&sums, nw * 64 * 4);
	std::vector<u32> s0(nw * 64), s1(nw * 64);
	for (int dma = 0; dma < 2; dma++) {
		float ms = 0;
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(k_dma, dim3(nw / 4), dim3(256), 0, 0, dst, sums, dma, 2);
			hipEventRecord(e1); hipEventSynchronize(e1);
			hipEventElapsedTime(&ms, e0, e1);
		}
		hipMemcpy(dma ? s1.data() : s0.data(), sums, nw * 64 * 4, hipMemcpyDeviceToHost);
		printf("  %-28s %7.3f ms  %8.1f GB/s\n", dma ? "global_load_lds_dwordx4" : "global_load + ds_write_b128", ms,
		       (double)nw * 65536 * 2 / ms / 1e6);
	}
	size_t bad = 0;
	for (size_t i = 0; i < s0.size(); i++) bad += s0[i] != s1[i];
	printf("  LDS-DMA result equals the register-staged one: %s (%zu differing sums)\n", bad ? "NO" : "yes", bad);
	return 0;
}
