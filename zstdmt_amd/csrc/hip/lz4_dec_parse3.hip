/*
 * lz4_dec_parse3.hip -- LZ4 frame decoder, parse stage of the round-3 pipeline ("parse3").
 *
 * Second kernel of the pipeline described in lz4_dec_split.hip (replaces, together with the
 * frames and copy kernels, LZ4F_decompress at /root/reference/lib/lz4-mt_decompress.c:349-362):
 * lane per 64 KiB block, serial token walk, u16 token positions to the token list.  The kernel is
 * bound by instruction issue (two waves per SIMD is all the blocks of an 8 GiB batch give, and a wave
 * issues one instruction per ~5 cycles), so the step is an instruction budget, not a feature list:
 *
 *   - one 12-byte ring read gives the token, the literal-length byte and (for literal runs <= 5) the
 *     match-length byte; a second ring read is taken only in the steps where some lane needs it;
 *     the block's last sequence is recognised on the same path; 255-continuation bytes and malformed
 *     input go through a generic path that reads global memory (rare);
 *   - a lane that runs out of ring data simply idles until the next refill round (no memory round trip
 *     per token, no wave-wide urgent rounds): rings are four 64-byte units, every 8 steps a round tops
 *     up to TWO units per lane (4 lanes per unit, 16 units per load instruction), i.e. twice what a lane
 *     consumes on average, so a lane that fell behind catches up;
 *   - token positions leave through an 8-entry LDS tile PER LANE: 16 bytes per lane every 8 tokens,
 *     no cross-lane transposition, no wave sync;
 *   - nothing else: cutting the list into the batches the copy kernel executes was done here at first (one
 *     descriptor per batch, ~85 of 190 instructions per step); the copy kernel now finds its batches itself
 *     with two ballots on fields its lanes compute anyway, and this kernel's step lost the batch logic.
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define P3_BLK_STORED 0x80000000u
#define P3_BLK_EMPTY 0xFFFFFFFFu
#define P3_RING 256u
#define P3_UNIT 64u
#define P3_RSTRIDE (P3_RING + 16u) /* ring + mirror of its first 16 bytes (dword reads never wrap) */
#ifndef P3_CADENCE
#define P3_CADENCE 8u
#endif
#define P3_FAR 0x7FFFFF00u /* "the whole rest of the block is in the ring" */
#define P3_NONE 0xFFFFFFFFu

typedef u32 p3v4 __attribute__((vector_size(16)));

static __device__ __forceinline__ u64 p3_tok_base(u64 coff, u32 gb) { return ((coff / 3) & ~63ull) + 128ull * gb; }

extern "C" __global__ void __launch_bounds__(64)
zmt_dec_parse3_kernel(const u8 *__restrict__ stream, u64 stream_bytes, const u64 *__restrict__ blk_coff,
		      const u32 *__restrict__ blk_csize, const u64 *__restrict__ nblk_ptr, u16 *__restrict__ tok,
		      u32 *__restrict__ blk_ntok, u32 *__restrict__ blk_olen)
{
	__shared__ __attribute__((aligned(16))) u8 ring_lds[64 * P3_RSTRIDE];
	__shared__ __attribute__((aligned(16))) u8 tile_lds[64 * 16];
	__shared__ __attribute__((aligned(16))) u8 dump_lds[64];
	const int lane = wv_lane();
	const u32 gb = blockIdx.x * 64 + (u32)lane;
	const u64 nblk = *nblk_ptr;
	if ((u64)blockIdx.x * 64 >= nblk)
		return;
	const bool exists = (u64)gb < nblk;
	const u32 cs_raw = exists ? blk_csize[gb] : P3_BLK_EMPTY;
	const bool parse = exists && cs_raw != P3_BLK_EMPTY && !(cs_raw & P3_BLK_STORED);
	const u32 cs = parse ? cs_raw : 0;
	const u64 coff = parse ? blk_coff[gb] : 0;
	const u64 tbase = p3_tok_base(coff, gb);
	/* wave-relative 32-bit coordinates: 64 consecutive blocks span < 2^31 bytes of stream */
	u64 cmin = parse ? coff : ~0ull;
	for (int d = 32; d; d >>= 1) {
		const u32 lo = wv_shfl((u32)cmin, lane ^ d), hi = wv_shfl((u32)(cmin >> 32), lane ^ d);
		const u64 o = (u64)hi << 32 | lo;
		cmin = o < cmin ? o : cmin;
	}
	if (!wv_any(parse)) {
		if (exists) {
			blk_ntok[gb] = 0;
			blk_olen[gb] = (cs_raw == P3_BLK_EMPTY) ? 0 : (cs_raw & 0x7FFFFFFFu);
		}
		return;
	}
	const u64 abase = cmin & ~127ull; /* g = stream offset - abase: g % 64 is the position inside a refill unit */
	const u32 boff = (u32)(coff - abase);
	const u32 gend = boff + cs;
	const u8 *const src = stream + coff;
	const u8 *const gsrc = stream + abase;
	u8 *const myring = ring_lds + (u32)lane * P3_RSTRIDE;
	u16 *const mytile = (u16 *)(tile_lds + (u32)lane * 16u);
	u16 *const mytok = tok + tbase;

	u32 g = boff, opos = 0, n = 0;
	/* the ring holds [ghi - 256, ghi); [ghi, greq) is in flight; ghi = P3_FAR once the block's last unit is in */
	u32 ghi = boff & ~(P3_UNIT - 1), greq = ghi;
	u32 pend0 = P3_NONE, pend1 = P3_NONE; /* g of the units this lane has in flight */
	bool pend_any = false;
	p3v4 pv[8];
	for (int i = 0; i < 8; i++)
		pv[i] = (p3v4){0, 0, 0, 0};
	bool ok = true, done = !parse;
	const int rgrp = lane >> 2, rpiece = lane & 3; /* refill: 4 lanes per 64-byte unit */

	for (u32 step = 0;; step++) {
		/* ---------------- refill round: every P3_CADENCE steps, back to back at the start ---------------- */
		if ((step & (P3_CADENCE - 1)) == 0 || step < 3) {
			if (pend_any) {
				wv_sync();
				/* land: unit u of row r arrives in pv[4 u + i] of lanes 4 (r % 16) .. +3, i = r / 16 */
				ZMT_UNROLL
				for (int u = 0; u < 2; u++) {
					u32 pgs[4];
					ZMT_UNROLL
					for (int i = 0; i < 4; i++)
						pgs[i] = wv_shfl(u ? pend1 : pend0, 16 * i + rgrp);
					ZMT_UNROLL
					for (int i = 0; i < 4; i++) {
						const int r = 16 * i + rgrp;
						const bool live = pgs[i] != P3_NONE;
						const u32 ro = pgs[i] & (P3_RING - 1);
						u8 *const row = ring_lds + (u32)r * P3_RSTRIDE;
						u8 *const d = live ? row + ro + 16u * (u32)rpiece : dump_lds + 16u * (u32)rpiece;
						*(p3v4 *)d = pv[4 * u + i];
						/* mirror of the ring's first 16 bytes; every other lane stores its piece where it already is */
						*(p3v4 *)((live && ro == 0 && rpiece == 0) ? row + P3_RING : d) = pv[4 * u + i];
					}
				}
				wv_sync();
				if (pend0 != P3_NONE)
					ghi = pend0 + P3_UNIT;
				if (pend1 != P3_NONE)
					ghi = pend1 + P3_UNIT;
				if (ghi >= gend && ghi != P3_FAR && !done)
					ghi = P3_FAR;
			}
			/* everything fetched or requested lies behind the parse position: restart the ring there */
			if (!done && g >= greq && ghi != P3_FAR)
				ghi = greq = g & ~(P3_UNIT - 1);
			/* the slot to be overwritten holds [greq - 256, greq - 192): parsed already?  Up to two units */
			const bool need0 = !done & (greq < gend) & (greq <= g + (P3_RING - P3_UNIT));
			const bool need1 = need0 & (greq + P3_UNIT < gend) & (greq <= g + (P3_RING - 2 * P3_UNIT));
			pend0 = need0 ? greq : P3_NONE;
			pend1 = need1 ? greq + P3_UNIT : P3_NONE;
			greq += need1 ? 2 * P3_UNIT : need0 ? P3_UNIT : 0;
			pend_any = wv_any(need0);
			if (pend_any) {
				/* unconditional loads (an exec-masked load needs its destination initialised and ends in a
				 * wait of its own): a unit nobody asked for reads the wave's first line again.  16-byte
				 * aligned; may run up to 63 bytes past the block, inside the stream's slack (include/gpumt.h) */
				ZMT_UNROLL
				for (int u = 0; u < 2; u++) {
					u32 rgs[4];
					ZMT_UNROLL
					for (int i = 0; i < 4; i++) {
						const u32 pg = wv_shfl(u ? pend1 : pend0, 16 * i + rgrp);
						rgs[i] = pg != P3_NONE ? pg : 0u;
					}
					ZMT_UNROLL
					for (int i = 0; i < 4; i++)
						pv[4 * u + i] = *(const p3v4 *)(gsrc + rgs[i] + 16u * (u32)rpiece);
				}
			}
		}
		/* ---------------- one sequence per lane ---------------- */
		const bool in12 = !done & ((int)(ghi - g) >= 12);
		const u32 ro = g & (P3_RING - 1);
		const u32 *const wp = (const u32 *)(myring + (ro & ~3u)); /* may run into the mirror */
		const u32 w0 = wp[0], w1 = wp[1], w2 = wp[2];
		const u32 lo = wv_alignbyte(w1, w0, ro), hi = wv_alignbyte(w2, w1, ro);
		const u32 b8 = wv_alignbyte(0u, w2, ro) & 255u;
		const u32 L4 = (lo >> 4) & 15u, M4 = lo & 15u;
		const u32 b1 = (lo >> 8) & 255u;
		const bool lx = L4 == 15u, mx = M4 == 15u;
		const u32 lit = L4 + (lx ? b1 : 0u);
		const u32 idx = lit + (lx ? 4u : 3u); /* index of the match-length byte, if there is one */
		u32 b2 = idx < 8u ? (u32)((((u64)hi << 32) | lo) >> (8u * (idx & 7u))) & 255u : b8;
		const bool is_last = g + idx - 2u == gend; /* the literals end the block: its last sequence */
		const bool need2 = in12 & mx & (idx > 8u) & !is_last;
		if (wv_any(need2)) {
			/* the match-length byte lies beyond the 9 bytes the window is good for (literal run of 6 or more) */
			if (need2)
				b2 = (((int)(ghi - g) > (int)idx) & (idx < P3_RING - 2 * P3_UNIT)) ? myring[(g + idx) & (P3_RING - 1)] : 255u;
		}
		const u32 ml = is_last ? 0u : M4 + 4u + (mx ? b2 : 0u);
		const u32 nxt = is_last ? gend : g + idx + (mx ? 1u : 0u);
		const u32 oend = opos + lit + ml;
		/* not on this path: 255-continuation bytes (literal run >= 270, match >= 274, or a match-length byte the ring
		 * does not hold), a match that ends the block or runs past it, too much output: the frame-serial decoder's
		 * arithmetic from global memory */
		const bool odd = (lx & (b1 == 255u)) | (((mx & (b2 == 255u)) | (nxt >= gend)) & !is_last) | (oend > ZMT_BLOCK);
		const bool fast = in12 & !odd;
		const bool gen = in12 & odd;
		u32 e_lit = lit, e_ml = ml, e_nxt = nxt;
		bool emit = fast, last = fast & is_last;
		if (wv_any(gen)) {
			if (gen) {
				const u32 pos = g - boff;
				u32 t = src[pos], l2 = t >> 4, h = pos + 1;
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
				if (ok && (lend > cs || lend < h))
					ok = false;
				if (ok) {
					emit = true;
					e_lit = l2;
					e_ml = 0;
					if (lend == cs) {
						last = true; /* last sequence: literals only */
						e_nxt = gend;
					} else if (lend + 2 > cs) {
						ok = false;
					} else {
						u32 m2 = t & 15, m = lend + 2;
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
						e_ml = m2 + 4;
						if (m >= cs)
							ok = false; /* a block cannot end with a match */
						e_nxt = boff + m;
					}
					if (opos + e_lit + e_ml > ZMT_BLOCK || opos + e_lit + e_ml < opos)
						ok = false;
				}
				if (!ok) {
					done = true;
					emit = false;
				}
			}
		}
		/* ---------------- token list ---------------- */
		if (emit) {
			mytile[n & 7u] = (u16)(g - boff);
			n++;
			opos += e_lit + e_ml;
			g = e_nxt;
			if (last)
				done = true;
			if (((n & 7u) == 0) | done) {
				/* the 8-entry group that holds sequence n - 1 (entries past n are unused slack of the list) */
				*(p3v4 *)(mytok + ((n - 1u) & ~7u)) = *(const p3v4 *)mytile;
			}
		}
		if (!wv_any(!done))
			break;
	}
	if (exists) {
		if (parse) {
			blk_ntok[gb] = ok ? n : 0;
			blk_olen[gb] = ok ? opos : 0xFFFFFFFFu;
		} else {
			blk_ntok[gb] = 0;
			blk_olen[gb] = (cs_raw == P3_BLK_EMPTY) ? 0 : (cs_raw & 0x7FFFFFFFu);
		}
	}
}
