// microbenchmark: throughput of LDS access instructions on gfx950 by width, alignment and address
// pattern (what an LZ77 copy through an LDS window can afford).  hipcc --offload-arch=gfx950 -O3.
// Output: LDS-unit cycles per wave-instruction (elapsed cycles * 1 / instructions issued on the CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;

enum { R8, R16, R32, R64, R128, W8, W16, W32, W64, W128, BPERM, NOPS };
static const char *opname[] = {"ds_read_u8", "ds_read_u16", "ds_read_b32", "ds_read_b64", "ds_read_b128",
	"ds_write_b8", "ds_write_b16", "ds_write_b32", "ds_write_b64", "ds_write_b128", "ds_bpermute"};
static const int opw[] = {1, 2, 4, 8, 16, 1, 2, 4, 8, 16, 4};

typedef u32 v2u __attribute__((ext_vector_type(2)));
typedef u32 v4u __attribute__((ext_vector_type(4)));

#define RD8(INS, T, INIT)                                                                          \
	{                                                                                          \
		T r0 = INIT, r1 = INIT, r2 = INIT, r3 = INIT, r4 = INIT, r5 = INIT, r6 = INIT, r7 = INIT; \
		asm volatile(INS " %0, %8\n\t" INS " %1, %8 offset:2048\n\t" INS " %2, %8 offset:4096\n\t" \
			     INS " %3, %8 offset:6144\n\t" INS " %4, %8 offset:8192\n\t" INS " %5, %8 offset:10240\n\t" \
			     INS " %6, %8 offset:12288\n\t" INS " %7, %8 offset:14336\n\ts_waitcnt lgkmcnt(0)" \
			     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) \
			     : "v"(addr) : "memory");                                              \
		acc ^= fold(r0) ^ fold(r1) ^ fold(r2) ^ fold(r3) ^ fold(r4) ^ fold(r5) ^ fold(r6) ^ fold(r7); \
	}
#define WR8(INS, V)                                                                                \
	asm volatile(INS " %0, %1\n\t" INS " %0, %1 offset:2048\n\t" INS " %0, %1 offset:4096\n\t"  \
		     INS " %0, %1 offset:6144\n\t" INS " %0, %1 offset:8192\n\t" INS " %0, %1 offset:10240\n\t" \
		     INS " %0, %1 offset:12288\n\t" INS " %0, %1 offset:14336\n\ts_waitcnt lgkmcnt(0)" \
		     : : "v"(addr), "v"(V) : "memory");

__device__ __forceinline__ u32 fold(u32 x) { return x; }
__device__ __forceinline__ u32 fold(v2u x) { return x.x ^ x.y; }
__device__ __forceinline__ u32 fold(v4u x) { return x.x ^ x.y ^ x.z ^ x.w; }

// pattern 0: lane * width (aligned, conflict-free); 1: + 1 byte; 2: random byte address in 2 KiB;
// 3: lane * 13 + 3 (sequence-like: arbitrary alignment, ascending); 4: lane*width + 2; 5: random, 4-aligned
template <int OP>
__global__ void __launch_bounds__(1024) k_lds(u32 *out, int iters, int pat, u64 *cycles)
{
	__shared__ __attribute__((aligned(16))) u8 lds[16384 + 2048 + 64];
	const int tid = threadIdx.x, lane = tid & 63;
	for (int i = tid; i < (int)sizeof(lds); i += blockDim.x)
		lds[i] = (u8)(i * 7 + 3);
	__syncthreads();
	const int w = OP == R8 || OP == W8 ? 1 : OP == R16 || OP == W16 ? 2 : OP == R32 || OP == W32 || OP == BPERM ? 4 : OP == R64 || OP == W64 ? 8 : 16;
	u32 acc = 0;
	u32 base = (u32)(size_t)lds; // LDS address of the array (low 32 bits of the generic pointer = LDS offset)
	u32 wv = 0x01020304u * (u32)(tid + 1);
	v2u wv2 = {wv, ~wv};
	v4u wv4 = {wv, ~wv, wv * 3, wv * 5};
	u64 t0 = clock64();
	for (int it = 0; it < iters; it++) {
		u32 off;
		if (pat == 0) off = (u32)lane * (w < 4 ? 4 : w);
		else if (pat == 1) off = ((u32)lane * (w < 4 ? 4 : w) + 1);
		else if (pat == 2) off = ((u32)(lane + 64 * it + tid) * 2654435761u >> 13) & 2047;
		else if (pat == 3) off = (u32)lane * 13 + 3 + (it & 7);
		else if (pat == 4) off = ((u32)lane * (w < 4 ? 4 : w) + 2);
		else off = (((u32)(lane + 64 * it + tid) * 2654435761u >> 13) & 2047) & ~3u;
		u32 addr = base + off;
		if (OP == R8) RD8("ds_read_u8", u32, 0)
		if (OP == R16) RD8("ds_read_u16", u32, 0)
		if (OP == R32) RD8("ds_read_b32", u32, 0)
		if (OP == R64) RD8("ds_read_b64", v2u, ((v2u){0, 0}))
		if (OP == R128) RD8("ds_read_b128", v4u, ((v4u){0, 0, 0, 0}))
		if (OP == W8) WR8("ds_write_b8", wv)
		if (OP == W16) WR8("ds_write_b16", wv)
		if (OP == W32) WR8("ds_write_b32", wv)
		if (OP == W64) WR8("ds_write_b64", wv2)
		if (OP == W128) WR8("ds_write_b128", wv4)
		if (OP == BPERM) {
			for (int k = 0; k < 8; k++)
				acc ^= (u32)__builtin_amdgcn_ds_bpermute((int)((off + k) << 2), (int)(acc + wv));
		}
	}
	u64 t1 = clock64();
	__syncthreads();
	if (lane == 0)
		cycles[tid >> 6] = t1 - t0;
	out[tid] = acc;
}

// correctness of misaligned accesses: lane l reads T at byte offset l*13+3 and writes it at 8192 + l*17 + 1
__global__ void k_check(u8 *out)
{
	__shared__ __attribute__((aligned(16))) u8 lds[16384];
	const int lane = threadIdx.x;
	for (int i = lane; i < 16384; i += 64)
		lds[i] = (u8)(i * 7 + 3);
	__syncthreads();
	u32 base = (u32)(size_t)lds;
	u32 a = base + lane * 13 + 3, d = base + 8192 + lane * 40 + 1;
	u32 r32; v2u r64;
	asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r32), "=&v"(r64) : "v"(a) : "memory");
	asm volatile("ds_write_b32 %0, %1\n\tds_write_b64 %0, %2 offset:8\n\tds_write_b16 %0, %1 offset:20\n\ts_waitcnt lgkmcnt(0)" : : "v"(d), "v"(r32), "v"(r64) : "memory");
	__syncthreads();
	for (int i = lane; i < 16384; i += 64)
		out[i] = lds[i];
}

template <int OP> static void run(u32 *dout, u64 *dcy)
{
	const int iters = 1000;
	for (int waves : {4, 16}) {
		printf("%-14s waves/CU=%2d :", opname[OP], waves);
		for (int pat = 0; pat < 6; pat++) {
			hipLaunchKernelGGL(k_lds<OP>, dim3(1), dim3(64 * waves), 0, 0, dout, iters, pat, dcy);
			hipDeviceSynchronize();
			u64 cy[16];
			hipMemcpy(cy, dcy, 8 * waves, hipMemcpyDeviceToHost);
			u64 mx = 0;
			for (int i = 0; i < waves; i++) mx = cy[i] > mx ? cy[i] : mx;
			printf("  p%d %6.1f", pat, (double)mx / ((double)iters * 8 * waves));
		}
		printf("   (cycles per wave-instr on the CU; %d B/lane)\n", opw[OP]);
	}
}

int main()
{
	u32 *dout; u64 *dcy; u8 *dchk;
	hipMalloc(&dout, 4096 * 4); hipMalloc(&dcy, 8 * 64); hipMalloc(&dchk, 16384);
	printf("patterns: p0 lane*w aligned | p1 +1 byte | p2 random byte addr | p3 lane*13+3 | p4 +2 bytes | p5 random 4-aligned\n");
	run<R8>(dout, dcy); run<R16>(dout, dcy); run<R32>(dout, dcy); run<R64>(dout, dcy); run<R128>(dout, dcy);
	run<W8>(dout, dcy); run<W16>(dout, dcy); run<W32>(dout, dcy); run<W64>(dout, dcy); run<W128>(dout, dcy);
	run<BPERM>(dout, dcy);
	hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dchk);
	std::vector<u8> h(16384);
	hipMemcpy(h.data(), dchk, 16384, hipMemcpyDeviceToHost);
	int bad32 = 0, bad64 = 0, bad16 = 0;
	for (int l = 0; l < 64; l++) {
		int s = l * 13 + 3, d = 8192 + l * 40 + 1;
		for (int k = 0; k < 4; k++) bad32 += h[d + k] != (u8)((s + k) * 7 + 3);
		for (int k = 0; k < 8; k++) bad64 += h[d + 8 + k] != (u8)((s + k) * 7 + 3);
		for (int k = 0; k < 2; k++) bad16 += h[d + 20 + k] != (u8)((s + k) * 7 + 3);
	}
	printf("misaligned access correctness: b32 bad=%d b64 bad=%d b16 bad=%d\n", bad32, bad64, bad16);
	return 0;
}
