/*
 * line_copy_step.hip -- the inner loop of the line-oriented LZ4 copy stage ("copy5") that round 5 simulated
 * (tools/sim_copy5.py, profiles/r05_sweeps/copy5_sim.txt) and did not build: compiled for gfx950 so that the instruction
 * count of ONE piece step is read from the ISA instead of assumed (the simulation's model had 28).
 *
 *   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/ubench/line_copy_step.hip -o /tmp/line_copy_step.s
 *   (count the instructions between the loop's header and its back edge; tools/sim_copy5.py prints steps per batch)
 *
 * A lane owns one 16-byte line of the batch's output.  Per step it takes the piece (literal or match segment cut at the
 * line) that contains its cursor: the sequence's fields from an LDS table, the source address, a readiness test on the
 * done-mask of the batch's lines (a source inside this batch must be complete), ONE unaligned 16-byte read placed so
 * that byte k of what is read is byte k of the line, a threshold merge over the line register, the cursor and the
 * sequence index advanced, the line written to the ring when it is full.  Lockstep scheme A2 of the simulation.  Not a
 * decoder: the batch set-up (fields, prefix sum, cut, owner look-up: copy3's measured 300 instructions per batch) and
 * the flush are outside; a match that overlaps its own line's unfinished bytes commits the line register to the ring
 * first (one more LDS write) -- the byte-serial handling of offsets below 16 is not here either.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 v4 __attribute__((vector_size(16)));
#define WIN 4096u
struct SeqRec { u32 op, lit, ml_off, lsrc; }; /* ml | off << 16 */

static __device__ __forceinline__ v4 ld16(const u8 *p) { v4 v; __builtin_memcpy(&v, p, 16); return v; }
/* bytes >= t (0..16) of the line from `n`, the others from `a` */
static __device__ __forceinline__ v4 merge_from(v4 a, v4 n, u32 t)
{
	v4 r;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int lo = (int)(8u * t) - 32 * k; /* first bit of dword k that comes from n */
		const u32 m = lo <= 0 ? 0xFFFFFFFFu : lo >= 32 ? 0u : 0xFFFFFFFFu << lo;
		r[k] = (a[k] & ~m) | (n[k] & m);
	}
	return r;
}

extern "C" __global__ void __launch_bounds__(64)
line_copy_steps(const u8 *__restrict__ comp, u8 *__restrict__ out, const u32 *__restrict__ batch /* o0, o_end, owner per line ... */,
		u32 nbatch)
{
	__shared__ __attribute__((aligned(16))) u8 ring[WIN + 16];
	__shared__ __attribute__((aligned(16))) SeqRec tab[64];
	const u32 lane = threadIdx.x;
	for (u32 bi = blockIdx.x; bi < nbatch; bi += gridDim.x) {
		const u32 *B = batch + (size_t)bi * 80;
		const u32 o0 = B[0], o_end = B[1], L0 = o0 >> 4, nl = ((o_end - 1) >> 4) - L0 + 1;
		tab[lane] = *(const SeqRec *)(B + 16) ; /* (stand-in for the batch set-up) */
		__syncthreads();
		const bool live = lane < nl;
		const u32 b = (L0 + lane) << 4;
		u32 s = B[8 + (lane >> 3)] & 63u; /* (stand-in for the owner look-up) */
		u32 x = b > o0 ? b : o0;
		const u32 xend = b + 16 < o_end ? b + 16 : o_end;
		v4 acc = *(const v4 *)(ring + (b & (WIN - 1)));
		bool done = !live;
		const u32 ring_lo = o_end + 16 > WIN ? o_end + 16 - WIN : 0;
		/* ------------------------------------------------------------------ the piece loop: one step = one piece */
		for (;;) {
			const u64 dm = __builtin_amdgcn_ballot_w64(done);
			if (dm == ~0ull)
				break;
			const SeqRec q = tab[s];
			const u32 lend = q.op + q.lit, ml = q.ml_off & 0xFFFFu, off = q.ml_off >> 16, send = lend + ml;
			const bool islit = x < lend;
			u32 e = islit ? lend : send;
			e = e < xend ? e : xend;
			const u32 t = x - b;
			bool ready = true, own = false;
			u32 sp = 0;
			if (!islit) {
				sp = x - off;
				if (off < e - x)
					e = x + off; /* a piece never reads what it writes */
				const u32 sl0 = (sp >> 4) - L0, sl1 = ((sp + (e - x) - 1) >> 4) - L0; /* as signed: < 0 = before the batch */
				const bool in0 = (int)sl0 >= 0, in1 = (int)sl1 >= 0;
				own = (in0 && sl0 == lane) || (in1 && sl1 == lane);
				ready = (!in0 || sl0 == lane || ((dm >> (sl0 & 63)) & 1)) && (!in1 || sl1 == lane || ((dm >> (sl1 & 63)) & 1));
			}
			const bool go = !done && ready;
			if (go && own)
				*(v4 *)(ring + (b & (WIN - 1))) = acc; /* the source reaches into this line's own finished bytes */
			__builtin_amdgcn_wave_barrier();
			v4 n = acc;
			if (go) {
				if (islit)
					n = ld16(comp + q.lsrc + (x - q.op) - t);
				else if (sp < ring_lo)
					n = ld16(out + sp - t);
				else
					n = ld16(ring + ((sp - t) & (WIN - 1)));
				acc = merge_from(acc, n, t);
				x = e;
				s += x == send ? 1u : 0u;
				if (x == xend) {
					*(v4 *)(ring + (b & (WIN - 1))) = acc;
					if ((b & (WIN - 1)) == 0)
						*(v4 *)(ring + WIN) = acc;
					done = true;
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
		if (live)
			*(v4 *)(out + b) = acc;
		__syncthreads();
	}
}
