/*
 * lz4_dec_parse4.hip -- LZ4 frame decoder, parse stage, round 4 ("parse4").
 *
 * Second kernel of the pipeline described in lz4_dec_split.hip (replaces, together with the frames and copy
 * kernels, LZ4F_decompress at /root/reference/lib/lz4-mt_decompress.c:349-362): lane per 64 KiB block, serial
 * token walk, u16 token positions to the token list.
 *
 * Its predecessor (round 3's parse3, deleted; numbers in profiles/r04_sweeps/parse4_vs_parse3.txt) was measured at 147
 * instructions per step with two waves per SIMD (a wave issues one instruction per 5-8 cycles: the step IS the
 * time, 5 500 steps per block): 3.34 ms per 8 GiB.  What this kernel does differently, all of it to shrink the step
 * (80 instructions, 2.35 - 2.5 ms; the rest is the latency of the chain -- three LDS round trips and ~60 dependent
 * vector instructions per step, with only the 131 072 blocks of an 8 GiB batch to run side by side):
 *
 *   - the step reads exactly the three bytes it needs (token, literal-length byte, match-length byte) with three
 *     byte reads of the lane's LDS ring -- byte reads cost the LDS pipe the same at any address (tools/ubench/
 *     lds_cost.hip) -- instead of a 12-byte window (three dword reads, three funnel shifts, a 64-bit shift);
 *   - everything that is not "a sequence with at most one length byte of each kind that is followed by another
 *     sequence" -- 255-continuation bytes, literal runs above 124, the block's last sequence, malformed input --
 *     leaves the step through ONE wave-wide test into a path that parses the token from global memory with all
 *     the checks; the step itself carries no end-of-block logic;
 *   - the output-size check moved out of the step (the sum is checked once, at the end).  That is safe because a
 *     block's compressed size is capped at 64 KiB by zmt_dec_frames_kernel (bsz > ZMT_BLOCK is ST_BAD_BLOCK there,
 *     lz4_dec_split.hip): at most cs / 3 tokens of at most cs bytes each cannot overflow the u32 sum, a malformed
 *     block is merely walked to its end before it is flagged;
 *   - refill: 4 lanes of a quad move the 64-byte units of the quad's 4 rows, the unit positions travel by DPP
 *     quad broadcasts (parse3: ds_bpermute), twice 4 loads every 8 steps;
 *   - token positions leave through a 16-entry tile per lane at wave-uniform points (every 8 steps, with the
 *     refill), so the store is not re-tested every step;
 *   - the loop's exit test runs every 8 steps.
 *
 * Blocks whose wave-relative position does not fit 31 bits (records far apart in a caller-supplied layout; the host
 * engines never produce one) are walked token by token through the global-memory path: slow, correct, no sentinel
 * collisions.
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define P4_BLK_STORED 0x80000000u
#define P4_BLK_EMPTY 0xFFFFFFFFu
#define P4_RING 256u
#define P4_UNIT 64u
#define P4_RSTRIDE (P4_RING + 16u) /* ring + mirror of its first 16 bytes (a[1] of the last byte) */
#define P4_CADENCE 8u
#define P4_FAR 0x7FFFFF00u /* "the whole rest of the block is in the ring" */
#define P4_NONE 0xFFFFFFFFu
#define P4_BIGLIT 120u /* a literal run above this goes the slow way (the ring holds 192 bytes in front of a token at least) */

typedef u32 p4v4 __attribute__((vector_size(16)));

static __device__ __forceinline__ u64 p4_tok_base(u64 coff, u32 gb) { return ((coff / 3) & ~63ull) + 128ull * gb; }

extern "C" __global__ void __launch_bounds__(64)
zmt_dec_parse4_kernel(const u8 *__restrict__ stream, u64 stream_bytes, const u64 *__restrict__ blk_coff,
		      const u32 *__restrict__ blk_csize, const u64 *__restrict__ nblk_ptr, u16 *__restrict__ tok,
		      u32 *__restrict__ blk_ntok, u32 *__restrict__ blk_olen)
{
	__shared__ __attribute__((aligned(16))) u8 ring_lds[64 * P4_RSTRIDE];
	__shared__ __attribute__((aligned(16))) u8 tile_lds[64 * 32];
	__shared__ __attribute__((aligned(16))) u8 dump_lds[64];
	const int lane = wv_lane();
	const u32 gb = blockIdx.x * 64 + (u32)lane;
	const u64 nblk = *nblk_ptr;
	if ((u64)blockIdx.x * 64 >= nblk)
		return;
	const bool exists = (u64)gb < nblk;
	const u32 cs_raw = exists ? blk_csize[gb] : P4_BLK_EMPTY;
	const bool parse = exists && cs_raw != P4_BLK_EMPTY && !(cs_raw & P4_BLK_STORED);
	const u32 cs = parse ? cs_raw : 0;
	const u64 coff = parse ? blk_coff[gb] : 0;
	const u64 tbase = p4_tok_base(coff, gb);
	u64 cmin = parse ? coff : ~0ull;
	for (int d = 32; d; d >>= 1) {
		const u32 lo = wv_shfl((u32)cmin, lane ^ d), hi = wv_shfl((u32)(cmin >> 32), lane ^ d);
		const u64 o = (u64)hi << 32 | lo;
		cmin = o < cmin ? o : cmin;
	}
	if (!wv_any(parse)) {
		if (exists) {
			blk_ntok[gb] = 0;
			blk_olen[gb] = (cs_raw == P4_BLK_EMPTY) ? 0 : (cs_raw & 0x7FFFFFFFu);
		}
		return;
	}
	/* wave-relative 32-bit coordinates g = stream offset - abase (g % 64 = position inside a refill unit); a block
	 * that does not fit them is walked through the global-memory path only, in block-relative coordinates */
	const u64 abase = ((u64)wv_readfirst((u32)(cmin >> 32)) << 32 | wv_readfirst((u32)cmin)) & ~127ull; /* (uniform: scalar base for the refill loads) */
	const bool outl = parse && (coff - abase + cs >= (u64)P4_FAR);
	const u32 boff = outl ? 0u : (u32)(coff - abase);
	const u32 gend = boff + cs;
	const u8 *const src = stream + coff;
	const u8 *const gsrc = stream + abase;
	u8 *const myring = ring_lds + (u32)lane * P4_RSTRIDE;
	u16 *const mytile = (u16 *)(tile_lds + (u32)lane * 32u);
	u16 *const mytok = tok + tbase;
	const u32 piece16 = 16u * ((u32)lane & 3u);
	u8 *const quadrow = ring_lds + ((u32)lane & ~3u) * P4_RSTRIDE + piece16; /* + i * P4_RSTRIDE: row 4 k + i of the quad */

	u32 g = boff, opos = 0, n = 0, nst = 0;
	/* the ring holds [ghi - 256, ghi); [ghi, greq) is in flight; ghi = P4_FAR once the block's last unit is in */
	u32 ghi = outl ? 0u : (boff & ~(P4_UNIT - 1)), greq = outl ? gend : ghi;
	u32 pend0 = P4_NONE, pend1 = P4_NONE; /* g of the units this lane's row has in flight */
	p4v4 pv[8];
	bool ok = true, done = !parse;

	/* requests: up to two units per row; the loads are unconditional (a row that asked for nothing reads the wave's
	 * first line again), 16-byte aligned; they may run up to 63 bytes past the block, inside the stream's slack */
#define P4_ISSUE()                                                                                                 \
	do {                                                                                                       \
		if (!done && !outl && g >= greq && ghi != P4_FAR)                                                  \
			ghi = greq = g & ~(P4_UNIT - 1); /* the walk ran past everything requested: restart there */ \
		const bool need0_ = !done & (greq < gend) & (greq <= g + (P4_RING - P4_UNIT));                     \
		const bool need1_ = need0_ & (greq + P4_UNIT < gend) & (greq <= g + (P4_RING - 2 * P4_UNIT));      \
		pend0 = need0_ ? greq : P4_NONE;                                                                   \
		pend1 = need1_ ? greq + P4_UNIT : P4_NONE;                                                         \
		greq += need1_ ? 2 * P4_UNIT : need0_ ? P4_UNIT : 0;                                               \
		P4_LOAD(0, 0, pend0);                                                                              \
		P4_LOAD(1, 1, pend0);                                                                              \
		P4_LOAD(2, 2, pend0);                                                                              \
		P4_LOAD(3, 3, pend0);                                                                              \
		P4_LOAD(4, 0, pend1);                                                                              \
		P4_LOAD(5, 1, pend1);                                                                              \
		P4_LOAD(6, 2, pend1);                                                                              \
		P4_LOAD(7, 3, pend1);                                                                              \
	} while (0)
#define P4_LOAD(SLOT, I, PEND)                                                                                     \
	do {                                                                                                       \
		const u32 pg_ = wv_quad(PEND, I);                                                                  \
		pv[SLOT] = *(const p4v4 *)(gsrc + ((pg_ != P4_NONE ? pg_ : 0u) + piece16));                        \
	} while (0)
	/* landings: unit of row 4 k + i arrives in the 4 lanes of quad k */
#define P4_STORE(SLOT, I, PEND)                                                                                    \
	do {                                                                                                       \
		const u32 pg_ = wv_quad(PEND, I);                                                                  \
		const bool live_ = pg_ != P4_NONE;                                                                 \
		const u32 ro_ = pg_ & (P4_RING - 1);                                                               \
		u8 *const d_ = live_ ? quadrow + (I) * P4_RSTRIDE + ro_ : dump_lds + piece16;                      \
		*(p4v4 *)d_ = pv[SLOT];                                                                            \
		/* mirror of the ring's first 16 bytes; every other lane stores its piece where it already is */  \
		*(p4v4 *)((live_ && ro_ == 0 && piece16 == 0) ? d_ + P4_RING : d_) = pv[SLOT];                     \
	} while (0)
#define P4_LAND()                                                                                                  \
	do {                                                                                                       \
		wv_sync();                                                                                         \
		P4_STORE(0, 0, pend0);                                                                             \
		P4_STORE(1, 1, pend0);                                                                             \
		P4_STORE(2, 2, pend0);                                                                             \
		P4_STORE(3, 3, pend0);                                                                             \
		P4_STORE(4, 0, pend1);                                                                             \
		P4_STORE(5, 1, pend1);                                                                             \
		P4_STORE(6, 2, pend1);                                                                             \
		P4_STORE(7, 3, pend1);                                                                             \
		wv_sync();                                                                                         \
		if (pend0 != P4_NONE)                                                                              \
			ghi = pend0 + P4_UNIT;                                                                     \
		if (pend1 != P4_NONE)                                                                              \
			ghi = pend1 + P4_UNIT;                                                                     \
		if (ghi >= gend && ghi != P4_FAR && !done && !outl)                                                \
			ghi = P4_FAR;                                                                              \
	} while (0)

	P4_ISSUE();
	P4_LAND();
	P4_ISSUE();
	for (;;) {
		/* ---------------- every 8 steps: token groups out, landings, requests, exit test ---------------- */
		if (n - nst >= 8u) {
			*(p4v4 *)(mytok + nst) = *(const p4v4 *)(mytile + (nst & 8u));
			nst += 8u;
		}
		if (!wv_any(!done))
			break;
		P4_LAND();
		P4_ISSUE();
		ZMT_UNROLL
		for (u32 sub = 0; sub < P4_CADENCE; sub++) {
			/* ---------------- one sequence per lane ---------------- */
			const u8 *const a = myring + (g & (P4_RING - 1));
			const u32 tk = a[0], b1 = a[1]; /* a[1] of the ring's last byte is the mirror's first */
			const u32 avail = ghi - g;      /* as a signed number: bytes of the ring in front of g */
			const u32 L4 = tk >> 4, M4 = tk & 15u;
			const bool lx = L4 == 15u, mx = M4 == 15u;
			const u32 x1 = lx ? b1 : 0u;
			const u32 lit = L4 + x1;
			const u32 idx = lit + (lx ? 4u : 3u); /* the match-length byte, if there is one, sits at g + idx */
			const u32 gi = g + idx;
#ifdef P4_EXTRA_TRIP
			/* developer A/B: one more dependent LDS round trip on the step's chain (what is a round trip worth?) */
			const u32 b2x = myring[gi & (P4_RING - 1)];
			const u32 b2 = myring[(gi + b2x * (gridDim.y - 1u)) & (P4_RING - 1)];
#else
			const u32 b2 = myring[gi & (P4_RING - 1)];
#endif
			const u32 x2 = mx ? b2 : 0u;
			const u32 ml = M4 + 4u + x2;
			const u32 nxt = gi + (mx ? 1u : 0u);
			const bool have = (int)idx < (int)avail; /* token .. match-length byte are in the ring (idx >= 3) */
			/* not in the step: a literal run above 120 (a 255 length byte is one), a 255 match-length byte, the block's
			 * last sequence (or anything else that leaves no room for a next token).  A long run may never fit the
			 * ring: it goes as soon as its two length bytes are there */
			const bool oddA = x1 > P4_BIGLIT - 15u, oddB = x2 == 255u, oddC = nxt >= gend;
			const bool slow = (!done) & (outl | (oddA & ((int)avail >= 2)) | (have & (oddB | oddC)));
			const bool fast = (!done) & have & !(oddA | oddB | oddC);
			if (slow) {
				/* the frame-serial decoder's arithmetic, from global memory, every check made */
				const u32 pos = g - boff;
				u32 t = src[pos], l2 = t >> 4, h = pos + 1, m2 = 0;
				bool last = false;
				if (l2 == 15) {
					u32 b;
					do {
						if (h >= cs) {
							ok = false;
							break;
						}
						b = src[h++];
						l2 += b;
					} while (b == 255);
				}
				const u32 lend = h + l2;
				u32 m = lend + 2;
				if (ok && (lend > cs || lend < h))
					ok = false;
				if (ok) {
					if (lend == cs) {
						last = true; /* last sequence: literals only */
						m = cs;
					} else if (lend + 2 > cs) {
						ok = false;
					} else {
						m2 = t & 15;
						if (m2 == 15) {
							u32 b;
							do {
								if (m >= cs) {
									ok = false;
									break;
								}
								b = src[m++];
								m2 += b;
							} while (b == 255);
						}
						m2 += 4;
						if (m >= cs)
							ok = false; /* a block cannot end with a match */
					}
				}
				if (ok) {
					mytile[n & 15u] = (u16)pos;
					n++;
					opos += l2 + m2;
					g = boff + m;
				}
				if (!ok || last)
					done = true;
			}
			/* ---------------- token list ---------------- */
			if (fast) {
				mytile[n & 15u] = (u16)(g - boff);
				n++;
				opos += lit + ml;
				g = nxt;
			}
		}
	}
	/* what the 8-step points have not stored yet: at most two groups (entries past n are unused slack of the list) */
	while (nst < n) {
		*(p4v4 *)(mytok + nst) = *(const p4v4 *)(mytile + (nst & 8u));
		nst += 8u;
	}
	if (opos > ZMT_BLOCK)
		ok = false;
	if (exists) {
		if (parse) {
			blk_ntok[gb] = ok ? n : 0;
			blk_olen[gb] = ok ? opos : 0xFFFFFFFFu;
		} else {
			blk_ntok[gb] = 0;
			blk_olen[gb] = (cs_raw == P4_BLK_EMPTY) ? 0 : (cs_raw & 0x7FFFFFFFu);
		}
	}
#undef P4_ISSUE
#undef P4_LOAD
#undef P4_STORE
#undef P4_LAND
}
