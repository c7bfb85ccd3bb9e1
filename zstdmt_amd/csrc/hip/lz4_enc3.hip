/*
 * lz4_enc3.hip -- LZ4 frame encoder v3 (default): same bit-exact greedy parse as lz4_enc.hip
 * (reference call site /root/reference/lib/lz4-mt_compress.c:281), rebuilt
 * around what the MI355X measurements say is expensive: dependent global loads and wide gathers.
 *
 *  - The chunk's input is streamed through a 1 KiB LDS ring (512-byte coalesced refills), so hashing,
 *    the ip side of catch-up / match counting, the re-match test and literal sources are LDS reads.
 *  - The search evaluates the reference's probe sequence in batches of 10, 16, 32, 64, 64, ...
 *    lanes (lane j = probe kbase + j).  Most matches sit within the first few probes, and every probe
 *    is a candidate fetched from memory: the first batch is as small as pays (ENC3_B0 below).  In-batch duplicate hashes
 *    are detected exactly (LDS atomic-or bitmap) and resolved by a readlane loop only when present.
 *  - One cooperative fetch of [match-32, match+96) then serves catch-up (backward) and the match
 *    length (forward) for the common case; longer runs continue 64 bytes at a time.
 *  - Sequences are not written out one by one: they collect in registers (lane = number mod 64) and leave 64 at a
 *    time, every lane writing its own (seq3_flush); the output position is advanced and tested per sequence, as the
 *    reference does; the candidate's 24-byte window comes as one 16-byte and one 8-byte load.  288 -> 256 ms per
 *    8 GiB (round 5).
 *  - For chunks <= 128 KiB the 4096-entry table stores 17-bit positions as u16 + one bit
 *    (8.5 KiB instead of 16 KiB).  A chunk-wave is a latency-bound dependent chain and the encoder's
 *    time follows 1 / (waves per CU) (14 waves 131.7 ms per 2 GiB, 12: 157, 9: 186, 5: 289), so LDS is
 *    budgeted for 16 waves per CU = 10 240 B: table 8 704, 1 KiB ring + mirror, 256 B duplicate filter,
 *    144 B match window.  Larger rings serve more candidates from LDS and lose more in waves than they
 *    gain (2 KiB at 14 waves: 131.7 ms, 1 KiB at 16 waves: 120.1 ms, 8 KiB at 9 waves: 171 ms).
 *
 * Table modes: T_U16  single block <= 64 KiB (byU16: 8192 x u16, 13-bit hash of 4 bytes)
 *              T_P17  linked blocks, chunk <= 128 KiB (byU32 semantics, 17-bit entries)
 *              T_U32  linked blocks, any chunk size (4096 x u32)
 */
#include "lz4_enc_shared.h"

template <int TM, bool PROF>
static __device__ u32 encode_block3(u32 *tlo, u32 *thi, u32 *bitmap, InRing &R, u32 pos, u32 len, u8 *dst,
				    u32 cap, int lane)
{
	const u8 *chunk = R.chunk;
	const u32 iend = pos + len;
	const u32 mflimit_p1 = iend - MFLIMIT + 1;
	const u32 matchlimit = iend - LASTLITERALS;
	const u32 low = (TM == T_U16) ? pos : 0;
	const bool second = (pos >> 16) != 0; /* T_P17: the 17th bit of every position of this block */
	u32 ip = pos, anchor = pos, op = 0;
	Seq3 sq = {0, 0, 0, 0, 0};
	u32 nsq = 0; /* sequences of this block so far; those not written yet: nsq & 63 */
	/* 1 after a match: the search opens with the reference's immediate re-match probe at ip.  That test is a
	 * probe like the others (look T[h(ip)] up, insert ip, compare 4 bytes), the search it falls into when it
	 * fails continues at ip + 1, ip + 2, ... -- so it rides in lane 0 of the search's first batch and the two
	 * memory round trips (its candidate, the first batch's candidates) become one */
	u32 rmode = 0;

	if (len < MFLIMIT + 1)
		goto last_literals;

	ring_want(R, ip, lane);
	{
		const u64 x = wv_readfirst((u32)in_ld64(R, ip)) | (u64)wv_readfirst((u32)(in_ld64(R, ip) >> 32)) << 32;
		if (lane == 0)
			t_write<TM>(tlo, thi, hash3<TM>(TM == T_U16 ? (u64)(u32)x : x), ip, second);
	}
	ip++;

	for (;;) {
		u32 match;
		u32 quick; /* bit 31: catch-up and match length settled from the batch's own loads; back << 8 | fwd */
		/* ---------------- search ---------------- */
		{
			const u32 ipr = ip;          /* position of the re-match probe (rmode) */
			const u32 ip0 = ip + rmode;  /* probe k of the reference's search loop is at probe_pos3(ip0, k) */
			u32 kbase = 0, bsz = ENC3_B0 + rmode, r = rmode;
			for (u32 batch = 0;; batch++) {
				/* probes k <= 64 are consecutive positions (wave-uniform test: no divergent schedule arithmetic);
				 * lane 0 of a batch that opens with the re-match probe is then at ip0 - 1 = ipr by itself */
				const bool consec = kbase + bsz - r <= 65;
				u32 gap = 1;
				u32 cur = ip0 + kbase + (u32)lane - r;
				if (E_RARE(!consec))
					cur = probe_pos3(ip0, kbase + (u32)lane - r, &gap);
				const bool valid = (u32)lane < bsz && cur + gap <= mflimit_p1;
				/* T[h(ip - 2)] = ip - 2 precedes the re-match lookup: an idle lane hashes it with the batch */
				const bool ins2 = (r != 0) & (lane == 63);
				if (ins2)
					cur = ipr - 2;
				const u32 cur0 = wv_readlane(cur, 0); /* first (lowest) probe of the batch */
				const u64 vm = wv_ballot(valid);
				if (E_RARE(vm == 0))
					goto last_literals;
				EPC(R, 7);
				ring_want(R, cur0, lane);
				/* the probe's neighbourhood [cur - 8, cur + 16): x = 8 bytes at cur for the hash, xb / x1 in front
				 * and behind for the quick extension.  Consecutive probes lie inside what ring_want just made
				 * resident ([cur0 - 256, cur0 + 256), ip - 2 too): no per-lane residency test, every lane reads.
				 * Seven ALIGNED dwords + funnel shifts: an LDS access that is not dword aligned costs the CU's
				 * LDS pipe a cycle per active lane (tools/ubench/lds_cost.hip), and 16 chunk-waves share it */
				u64 x, xb = 0, x1 = 0;
				if (consec) {
					E_ASSERT(!(valid | ins2) || cur < 8 || ring_has(R, cur - 8, 24));
					const u32 pb = cur - 8;
					const u32 *const w = (const u32 *)(R.ring + (pb & (IRING - 1) & ~3u)); /* + 28 <= IRING + IMIRROR */
					const u32 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5], w6 = w[6];
					xb = (u64)wv_alignbyte(w1, w0, pb) | (u64)wv_alignbyte(w2, w1, pb) << 32;
					x = (u64)wv_alignbyte(w3, w2, pb) | (u64)wv_alignbyte(w4, w3, pb) << 32;
					x1 = (u64)wv_alignbyte(w5, w4, pb) | (u64)wv_alignbyte(w6, w5, pb) << 32;
				} else {
					x = (valid | ins2) ? in_ld64(R, cur) : 0;
				}
				const u32 h = hash3<TM>(TM == T_U16 ? (u64)(u32)x : x);
				if (ins2)
					t_write<TM>(tlo, thi, h, cur, second);
				wv_sync();
				u32 cand = t_read<TM>(tlo, thi, h); /* (idle lanes read too: no exec-mask region) */
				u32 prev_dup = 64, next_dup = 64;
				{
					EPC(R, 0);
					/* the candidate's neighbourhood [cand - 8, cand + 16) in one go, always from memory (the ring holds
					 * the same bytes for recent candidates, but a batch waits for its farthest one anyway, and one plain
					 * global load path is cheaper than a per-lane choice between LDS and memory): the 4-byte test, and
					 * for most matches all that catch-up and the match length need, so that the 128-byte window (a second
					 * round trip) is the exception.  The loads leave with the table's answer, BEFORE the duplicate filter
					 * has spoken: its atomic returns while they are under way, and the rare batch with a duplicate asks
					 * again for the lanes whose candidate an earlier probe of the batch replaces.
					 * Lanes without a candidate read chunk[0, 8) (no exec-mask region; a final record of 13..15 bytes has
					 * nothing readable at chunk + 16) */
					bool dist_ok = (TM == T_U16) || cand + DIST_MAX >= cur;
					bool probe = valid && dist_ok;
					bool wide = probe && cand >= 8; /* else (chunk start) 8 bytes at cand only, no quick extension */
					u32 a0 = !probe ? 0u : wide ? cand - 8 : cand;
					const u8 *gp = chunk + a0;
					/* (16 + 8 bytes: two vector-memory instructions instead of three.  A lane without a wide window reads
					 * [cand, cand + 16) -- readable: cand + 16 <= iend + 3, and d_in extends 64 readable bytes past n (include/gpumt.h; the host engines ask for n + 512) -- and uses
					 * its first half; its third read repeats the first) */
					u64 l0, l1, l2;
					ENC3_LD16(gp, l0, l1);
					l2 = ld64u(wide ? gp + 16 : gp);
					/* in-batch duplicates of a hash: every probe sets its bit of the folded filter (only the probes: an LDS
					 * atomic costs by active lanes) -- behind the loads, its wait would stand in front of them */
					u32 dold = 0;
					if (valid)
						dold = lds_or(&bitmap[(h & (BM_BITS - 1)) >> 5], 1u << (h & 31));
					const bool any_dup = wv_any(((dold >> (h & 31)) & 1) != 0);
					wv_sync();
					if (valid)
						bitmap[(h & (BM_BITS - 1)) >> 5] = 0;
					if (E_RARE(any_dup)) {
						for (u32 i = 0; i < bsz; i++) {
							const u32 hi_ = wv_readlane(h, (int)i);
							const bool vi = (vm >> i) & 1;
							if (vi && valid && hi_ == h) {
								if (i < (u32)lane)
									prev_dup = i;
								else if (i > (u32)lane && next_dup == 64)
									next_dup = i;
							}
						}
						/* an earlier probe of this batch had put its position there (the exchange is an LDS round
						 * trip: only on this rare path) */
						const u32 pc = wv_shfl(cur, (int)(prev_dup & 63));
						if (prev_dup < 64)
							cand = pc;
						dist_ok = (TM == T_U16) || cand + DIST_MAX >= cur;
						probe = valid && dist_ok;
						wide = probe && cand >= 8;
						a0 = !probe ? 0u : wide ? cand - 8 : cand;
						gp = chunk + a0;
						ENC3_LD16(gp, l0, l1);
						l2 = ld64u(wide ? gp + 16 : gp);
					}
					const u64 g0 = l0, g1 = wide ? l1 : l0, g2 = l2;
					const bool m = probe && (u32)g1 == (u32)x;
					const u64 mm = wv_ballot(m);
					if (PROF) R.pc[6] += 1;
					if (mm) {
						/* the first probe that verifies is the match; the insertions up to it are committed */
						const u32 jstar = (u32)wv_ffs(mm) - 1;
						if ((u32)lane <= jstar && !(next_dup <= jstar))
							t_write<TM>(tlo, thi, h, cur, second);
						wv_sync();
						EPC(R, 1);
						/* every lane works its own extension out (a handful of VALU steps, no branch); the
						 * winner's is taken.  fwd: equal bytes after the 4, 12 looked at; back: equal bytes in
						 * front, 8 looked at.  Settled when a difference (or the limit) lies inside them */
						const u32 room_ip = cur - anchor, room_m = cand - low;
						const u32 nb = room_ip < room_m ? room_ip : room_m;
						const u32 flimit = matchlimit - (cur + MINMATCH);
						const u64 df = ((x >> 32) | (x1 << 32)) ^ ((g1 >> 32) | (g2 << 32));
						const u32 d2 = (u32)(x1 >> 32) ^ (u32)(g2 >> 32);
						/* equal low bytes of the 96-bit difference d2 : df (ffs - 1 of 0 is the largest u32) */
						const u32 fbits_lo = (u32)wv_ffs(df) - 1u;
						const u32 fbits_hi = 64u + (u32)__builtin_ctzll((u64)d2 | 1ull << 32);
						u32 eqf = (fbits_lo < fbits_hi ? fbits_lo : fbits_hi) >> 3;
						const bool fdec = eqf < 12u || flimit <= 12u;
						if (eqf > flimit)
							eqf = flimit;
						const u64 db = xb ^ g0;
						u32 eqb = ((u32)__builtin_clzll(db | 1ull) + (db == 0)) >> 3; /* equal high bytes */
						const bool bdec = nb == 0 || eqb < 8u || nb <= 8u;
						if (eqb > nb)
							eqb = nb;
						const bool qok = consec && wide && cur0 >= 8 && fdec && bdec;
						const u32 qi = qok ? 0x80000000u | eqb << 8 | eqf : 0u;
						ip = wv_readlane(cur, (int)jstar);
						match = wv_readlane(cand, (int)jstar);
						quick = wv_readlane(qi, (int)jstar);
						break;
					}
					/* no probe verified: all of them are inserted, and an invalid one ends the block */
					if (valid && next_dup == 64)
						t_write<TM>(tlo, thi, h, cur, second);
					wv_sync();
					EPC(R, 1);
					if (E_RARE((u32)wv_popc(vm) < bsz))
						goto last_literals;
				}
				kbase += bsz - r;
				r = 0;
				bsz = (batch == 0) ? ENC3_B1 : (batch == 1 ? 32 : 64);
			}
		}
		/* ---------------- extend the match both ways ----------------
		 * Usually settled by the search batch's own loads (`quick`).  Else: the match side is made readable
		 * from LDS -- the input ring when the candidate is recent, else a 128-byte window fetched from memory
		 * --, backward (catch-up) and forward (match length) compares are independent -- the forward count
		 * from the probe position is the same whatever the catch-up finds -- so both LDS reads are in flight
		 * together. */
		u32 fwd; /* equal bytes following the 4 that matched at ip */
		if (quick >> 31) {
			const u32 back = (quick >> 8) & 0xFFu;
			fwd = (quick & 0xFFu) + back;
			ip -= back;
			match -= back;
		} else {
			ring_want(R, ip, lane); /* [ip, ip + 68) resident whatever the probe spacing was */
			mside_prepare(R, match, lane);
			u32 room = ip - anchor;
			if (match - low < room)
				room = match - low;
			const u32 nb = room < 64 ? room : 64;
			const u32 flimit = matchlimit - (ip + MINMATCH);
			bool eqb = false, stopf = true;
			if ((u32)lane < nb) {
				/* the bytes before ip are in the ring unless it was restarted less than nb bytes
				 * ago; the bytes before the match are in the ring, or (up to 32 of them) in the
				 * window -- wave-uniform tests, else the generic readers */
				const u32 a = ip >= R.rlo + nb ? (u32)R.ring[(ip - 1 - (u32)lane) & (IRING - 1)] : in_ld8(R, ip - 1 - (u32)lane);
				const u32 b = R.mbase == 0xFFFFFFFFu ? (u32)R.ring[(match - 1 - (u32)lane) & (IRING - 1)]
					      : (nb <= 32 && match >= 32) ? (u32)R.mwin[match - 1 - (u32)lane - R.mbase]
									  : m_ld8(R, match - 1 - (u32)lane);
				eqb = a == b;
			}
			if ((u32)lane < flimit)
				stopf = in_fwd8(R, ip + MINMATCH + (u32)lane) != m_fwd8(R, match + MINMATCH + (u32)lane);
			const u64 neb = ~wv_ballot(eqb);
			const u64 smf = wv_ballot(stopf);
			u32 back = neb ? (u32)wv_ffs(neb) - 1 : 64;
			if (back > nb)
				back = nb;
			fwd = smf ? (u32)wv_ffs(smf) - 1 : 64;
			if (E_RARE(back == 64)) { /* rare: catch-up continues beyond 64 bytes */
				u32 ip2 = ip - 64, m2 = match - 64;
				for (;;) {
					u32 r2 = ip2 - anchor;
					if (m2 - low < r2)
						r2 = m2 - low;
					if (r2 == 0)
						break;
					const u32 n2 = r2 < 64 ? r2 : 64;
					bool e2 = false;
					if ((u32)lane < n2)
						e2 = in_ld8(R, ip2 - 1 - (u32)lane) == m_ld8(R, m2 - 1 - (u32)lane);
					const u64 ne2 = ~wv_ballot(e2);
					u32 t2 = ne2 ? (u32)wv_ffs(ne2) - 1 : 64;
					if (t2 > n2)
						t2 = n2;
					ip2 -= t2;
					m2 -= t2;
					back += t2;
					if (t2 < 64)
						break;
				}
			}
			if (E_RARE(!smf)) { /* rare: the match runs on beyond 64 bytes */
				u32 base = 64;
				for (;;) {
					ring_want(R, ip + MINMATCH + base, lane);
					const u32 i2 = base + (u32)lane;
					bool st2 = true;
					if (i2 < flimit)
						st2 = in_ld8(R, ip + MINMATCH + i2) != m_ld8(R, match + MINMATCH + i2);
					const u64 sm2 = wv_ballot(st2);
					if (sm2) {
						fwd = base + (u32)wv_ffs(sm2) - 1;
						break;
					}
					base += 64;
				}
			}
			/* after catch-up the match starts `back` bytes earlier; its code length grows by that */
			ip -= back;
			match -= back;
			fwd += back;
		}
		EPC(R, 2);
		{
			/* ---------------- the sequence: its place in the output, its numbers into the collecting registers ---------------- */
			const u32 lit = ip - anchor, mc = fwd;
			const u32 token = op;
			/* what the sequence adds to the output: token, literals, offset, and the length bytes of runs of 15 and more */
			u32 adv = lit + 3u;
			if (E_RARE((lit | mc) >= 15u))
				adv += (lit >= 15 ? (lit - 15) / 255 + 1 : 0u) + (mc >= 15 ? (mc - 15) / 255 + 1 : 0u);
			/* the reference's two output-limit tests (before the literals: op + 1 + lit + 8 + lit / 255 > cap; before the
			 * offset: op' + 2 + 6 + (mc + 240) / 255 > cap) both pass when op + adv + 6 <= cap -- the second IS that, the
			 * first asks less (lit / 255 <= the literal run's length bytes) -- so one compare covers the common case and
			 * the two tests decide, in the reference's order, only near the limit */
			if (E_RARE(op + adv + 6u > cap)) {
				const u32 o1 = op + 1u;
				if (o1 + lit + (2 + 1 + LASTLITERALS) + lit / 255 > cap)
					return 0;
				const u32 o2 = o1 + (lit >= 15 ? (lit - 15) / 255 + 1 : 0u) + lit;
				if (o2 + 2 + (1 + LASTLITERALS) + (mc + 240) / 255 > cap)
					return 0;
			}
			op += adv;
			{
				const bool me = (u32)lane == (nsq & 63u);
				sq.tok = me ? token : sq.tok;
				sq.src = me ? anchor : sq.src;
				sq.lit = me ? lit : sq.lit;
				sq.mc = me ? mc : sq.mc;
				sq.off = me ? ip - match : sq.off;
			}
			nsq++;
			ip += mc + MINMATCH;
			anchor = ip;
			if ((nsq & 63u) == 0)
				seq3_flush(sq, 64, chunk, dst, lane);
			EPC(R, 4);
		}
		if (E_RARE(ip >= mflimit_p1))
			goto block_done;
		rmode = 1; /* T[h(ip - 2)] = ip - 2, the re-match test at ip and the search behind it: next batch */
	}
block_done:
last_literals:
	if ((nsq & 63u) != 0) /* (a block that fails below is stored raw: writing what it had is harmless) */
		seq3_flush(sq, nsq & 63u, chunk, dst, lane);
	{
		u32 run = iend - anchor;
		if (op + run + 1 + (run + 255 - 15) / 255 > cap)
			return 0;
		if (run >= 15) {
			if (lane == 0)
				dst[op] = 15 << 4;
			op++;
			op += put_len_ext3(dst + op, run - 15, lane);
		} else {
			if (lane == 0)
				dst[op] = (u8)(run << 4);
			op++;
		}
		copy_literals(R, dst + op, anchor, run, lane);
		op += run;
	}
	return op;
}

struct Enc3 {
	template <int TM, bool PROF>
	static __device__ __forceinline__ u32 block(u32 *tlo, u32 *thi, u32 *bitmap, InRing &R, u32 pos, u32 len, u8 *dst, u32 cap, int lane)
	{
		return encode_block3<TM, PROF>(tlo, thi, bitmap, R, pos, len, dst, cap, lane);
	}
};

#define ENC3_KERNEL(NAME, TM, TABBYTES, PROFILE)                                                  \
	extern "C" __global__ void __launch_bounds__(64)                                            \
	NAME(const u8 *__restrict__ in, u64 n, u32 chunk, u32 rec0, u32 nrec,                       \
	     u8 *__restrict__ slots, u64 slot_stride, u32 *__restrict__ rec_len,                   \
	     const u32 *__restrict__ chk, unsigned long long *prof)                                \
	{                                                                                          \
		__shared__ __attribute__((aligned(16))) u32 tlo[(TABBYTES) / 4];                   \
		__shared__ u32 thi[128];                                                           \
		__shared__ u32 bitmap[BM_BITS / 32];                                               \
		__shared__ __attribute__((aligned(16))) u8 ring[IRING + IMIRROR];                  \
		__shared__ __attribute__((aligned(16))) u8 mwin[MWIN + 16];                        \
		enc_frame_body<TM, PROFILE, Enc3>(tlo, thi, bitmap, ring, mwin, in, n, chunk, rec0, nrec, slots, \
			      slot_stride, rec_len, chk, prof);                                                     \
	}

ENC3_KERNEL(zmt_lz4_enc3_u16_kernel, T_U16, 16384, false)
ENC3_KERNEL(zmt_lz4_enc3_p17_kernel, T_P17, 8192, false)
ENC3_KERNEL(zmt_lz4_enc3_u32_kernel, T_U32, 16384, false)
#ifndef ZMT_EMU
/* developer: the same kernel with the phase counters (gpumt_set_variant("profile", 9), tools/enc_prof.py) */
ENC3_KERNEL(zmt_lz4_enc3_p17_prof_kernel, T_P17, 8192, true)
#endif
