// microbenchmark (round 3): LDS cycles per wave instruction by width, address pattern and active lanes --
// what the LZ4 copy stage pays for its sequence-shaped LDS traffic.  16 waves on one CU issue the same
// instruction 4x per iteration; the figure is CU cycles per wave instruction (= LDS pipe occupancy).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_cost.hip -o tools/ubench/lds_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
typedef u32 v2u __attribute__((ext_vector_type(2)));
typedef u32 v4u __attribute__((ext_vector_type(4)));

// OP: 0 w8 1 w16 2 w32 3 w64 4 w128 5 r8 6 r32 7 r64 8 r128
template <int OP>
__global__ void __launch_bounds__(1024) k_lds(u32 *out, int iters, int active, int pat, u64 *cycles)
{
	__shared__ __attribute__((aligned(16))) u8 lds[16384 + 8192 + 64];
	const int tid = threadIdx.x, lane = tid & 63;
	for (int i = tid; i < (int)sizeof(lds); i += blockDim.x)
		lds[i] = (u8)i;
	__syncthreads();
	const u32 base = (u32)(size_t)lds + (u32)(tid >> 6) * 64;
	u32 wv = 0x01020304u * (u32)(tid + 1);
	v2u w2 = {wv, ~wv};
	v4u w4 = {wv, ~wv, wv * 3, wv * 5};
	const bool on = (lane % (64 / active)) == 0;
	u32 h = (u32)lane * 2654435761u;
	u32 off;
	switch (pat) {
	case 0: off = (u32)lane * (OP == 0 || OP == 5 ? 1 : OP == 1 ? 2 : OP == 2 || OP == 6 ? 4 : OP == 3 || OP == 7 ? 8 : 16); break; // dense, aligned
	case 1: off = (u32)lane * 12; break;                 // sequence-shaped, dword aligned
	case 2: off = (u32)lane * 13 + 3; break;             // sequence-shaped, byte aligned
	case 3: off = ((h >> 20) & 1023) * 4; break;         // random dwords
	case 4: off = ((h >> 19) & 2047) * 2 + 1; break;     // random odd bytes
	case 5: off = (u32)lane * 20 + 8; break;             // 4-aligned, 8-misaligned for 64-bit, stride 20
	case 6: off = (u32)lane * 40; break;                 // 8-aligned, stride 40 (a per-lane window)
	default: off = ((h >> 21) & 511) * 8; break;         // random 8-aligned
	}
	const u32 addr = base + off;
	u32 acc = 0;
	u64 t0 = clock64();
	for (int it = 0; it < iters; it++) {
		if (on) {
			if (OP == 0) asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %1 offset:2048\n\tds_write_b8 %0, %1 offset:4096\n\tds_write_b8 %0, %1 offset:6144\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(wv) : "memory");
			if (OP == 1) asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %1 offset:2048\n\tds_write_b16 %0, %1 offset:4096\n\tds_write_b16 %0, %1 offset:6144\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(wv) : "memory");
			if (OP == 2) asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:2048\n\tds_write_b32 %0, %1 offset:4096\n\tds_write_b32 %0, %1 offset:6144\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(wv) : "memory");
			if (OP == 3) asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:2048\n\tds_write_b64 %0, %1 offset:4096\n\tds_write_b64 %0, %1 offset:6144\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(w2) : "memory");
			if (OP == 4) asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:2048\n\tds_write_b128 %0, %1 offset:4096\n\tds_write_b128 %0, %1 offset:6144\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(w4) : "memory");
			if (OP == 5) { u32 a, b, c, d; asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:2048\n\tds_read_u8 %2, %4 offset:4096\n\tds_read_u8 %3, %4 offset:6144\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory"); acc += a ^ b ^ c ^ d; }
			if (OP == 6) { u32 a, b, c, d; asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:2048\n\tds_read_b32 %2, %4 offset:4096\n\tds_read_b32 %3, %4 offset:6144\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory"); acc += a ^ b ^ c ^ d; }
			if (OP == 7) { v2u a, b, c, d; asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:2048\n\tds_read_b64 %2, %4 offset:4096\n\tds_read_b64 %3, %4 offset:6144\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory"); acc += a.x ^ b.y ^ c.x ^ d.y; }
			if (OP == 8) { v4u a, b, c, d; asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory"); acc += a.x ^ b.y ^ c.z ^ d.w; }
		}
	}
	u64 t1 = clock64();
	__syncthreads();
	if (lane == 0)
		cycles[tid >> 6] = t1 - t0;
	out[tid] = lds[tid] + acc;
}

template <int OP> static double run(u32 *dout, u64 *dcy, int active, int pat)
{
	const int waves = 16, iters = 1000;
	hipLaunchKernelGGL(k_lds<OP>, dim3(1), dim3(64 * waves), 0, 0, dout, iters, active, pat, dcy);
	hipDeviceSynchronize();
	u64 cy[16], mx = 0;
	hipMemcpy(cy, dcy, 8 * waves, hipMemcpyDeviceToHost);
	for (int i = 0; i < waves; i++) mx = cy[i] > mx ? cy[i] : mx;
	// clock64 ticks at 100 MHz on gfx9 (s_memrealtime) or shader clock (s_memtime): report raw and let the
	// aligned dense row calibrate
	return (double)mx / ((double)iters * 4 * waves);
}

int main()
{
	u32 *dout; u64 *dcy;
	hipMalloc(&dout, 4096 * 4); hipMalloc(&dcy, 8 * 64);
	const char *ops[] = {"write_b8", "write_b16", "write_b32", "write_b64", "write_b128", "read_u8", "read_b32", "read_b64", "read_b128"};
	const char *pats[] = {"dense aligned", "12*l (dword aligned)", "13*l+3 (byte aligned)", "random dwords", "random odd bytes", "20*l+8", "40*l (8-aligned)", "random 8-aligned"};
	printf("cycles per wave instruction (16 waves on one CU), active lanes 64 / 16 / 4\n");
	for (int op = 0; op < 9; op++)
		for (int pat = 0; pat < 8; pat++) {
			double r[3]; int k = 0;
			for (int active : {64, 16, 4}) {
				double v = 0;
				switch (op) {
				case 0: v = run<0>(dout, dcy, active, pat); break; case 1: v = run<1>(dout, dcy, active, pat); break;
				case 2: v = run<2>(dout, dcy, active, pat); break; case 3: v = run<3>(dout, dcy, active, pat); break;
				case 4: v = run<4>(dout, dcy, active, pat); break; case 5: v = run<5>(dout, dcy, active, pat); break;
				case 6: v = run<6>(dout, dcy, active, pat); break; case 7: v = run<7>(dout, dcy, active, pat); break;
				default: v = run<8>(dout, dcy, active, pat); break;
				}
				r[k++] = v;
			}
			printf("  %-10s %-24s %7.1f %7.1f %7.1f\n", ops[op], pats[pat], r[0], r[1], r[2]);
		}
	return 0;
}
