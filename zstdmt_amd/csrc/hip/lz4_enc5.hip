/*
 * lz4_enc5.hip -- LZ4 frame encoder v5 (round 6, the default for levels 1-2): the bit-exact greedy parse of lz4_enc3.hip
 * (reference call site /root/reference/lib/lz4-mt_compress.c:281), one memory round trip per
 * 64-POSITION WINDOW instead of one per sequence, and the parse through a window in vector code.
 *
 * What lz4_enc3.hip's counters said (profiles/r05_sq_counters.json): a chunk-wave is one dependent chain -- ring read, hash,
 * table read, the candidates' loads from memory, vote, table writes, extension, emit: ~360 instructions and ~4 500 cycles per
 * sequence -- and the LDS table (8.5 KiB per chunk) caps the chains at 16 per CU.  Its probe batch fetched the candidates of
 * the next 10 positions; a sequence of the bench text is 12 bytes long.
 *
 * Here a wave looks at 64 consecutive positions at once (lane j = position w0 + j).  Every lane hashes its position, reads
 * the table AS IT IS AT THE WINDOW'S START and fetches its candidate's neighbourhood [cand - 8, cand + 24) -- the loads the
 * probe batch made, one round trip -- and knows by itself whether its candidate verifies and how far the match would reach
 * (20 bytes forwards, 8 backwards: lz4_enc3.hip's "quick" extension).  Then the reference's parse is walked through the
 * window:
 *   - every lane also plays "a search starts here": its match lane = the first verifying lane at or behind it (one 64-bit
 *     shift + count-trailing-zeros of the ballot), that lane's numbers by two ds_bpermute, hence the lane where the NEXT
 *     search starts.  A scalar loop of ONE v_readlane per sequence follows these links from the window's first search and
 *     marks the starts (a run of "easy" sequences); everything else about the run's sequences -- literal run, catch-up,
 *     lengths, offset, their output positions (one wave prefix sum, the output limit tested per sequence as the reference
 *     does), the move into the collecting registers (rank by mbcnt, one LDS byte per sequence, three ds_bpermute), the set of
 *     inserted positions -- is lane-parallel: 89 % of the bench text's sequences;
 *   - a search that is not easy (its match needs more bytes than the lane fetched, its candidate lies inside the window, a
 *     twin that might verify sits in its range, it leaves the window) is done by the wave the general way, one at a time
 *     (the code of lz4_enc3.hip: input ring, 128-byte match window), and the run goes on behind it.
 * 5 sequences per window on the bench text, for about the candidates per sequence the probe batches fetched.
 *
 * Exactness.  The reference reads T[h(p)] and writes T[h(p)] = p at every probed position p, in position order, and inserts
 * ip - 2 behind every match; positions inside a match are neither read nor written.  A lane's table value is therefore
 * right unless a position of the same window with the same hash was INSERTED before it is probed: a probe of the search
 * in progress (all lanes from the search's start up to it) or a position committed by an earlier search of the window.
 * `prev` = the nearest earlier lane with the same hash ("twin": one LDS bitmap whose returning atomic names the lanes that found
 * their bit taken, one ballot per such lane settles it exactly; 64 positions on 4 096 entries: two of three windows have a pair).  Whether a
 * lane verifies against its twin is settled once per window (one ds_bpermute); which of the two candidates counts depends on
 * the parse, so a twin that might verify ends a lane-parallel run and is resolved by the general path, where the set of
 * inserted lanes is known (chains of three and more by a walk).  Table writes of a window are made at its end, later
 * positions over earlier ones, so that a block that fails the output limit leaves exactly the insertions the reference
 * made up to the failing sequence.  A search that runs out of the window goes on in the next one while its probes are
 * consecutive positions (the reference's first 65 probes of a search); one that is still running behind that
 * (incompressible data: skip acceleration) takes lz4_enc3.hip's probe batches with the exact step schedule.
 *
 * [MI355X, 8 GiB of the bench text, 128 KiB chunks] 255.9 ms (lz4_enc3) -> 105 ms -> 93 ms (the instruction count taken down:
 * profiles/r06_sweeps/lz4_enc5_steps.txt), bit-identical; on PRNG bytes and at 64 KiB / 1 MiB / 4 MiB chunks it is 1.2 - 2.9 x
 * lz4_enc3.  What bounds it now: its instruction count, scalar and vector alike (~0.16 ms per instruction per window; 0.71 scalar
 * instructions per cycle per CU of a ceiling of ~0.8 at 16 waves) and 5.2 TB/s of candidate lines (487 GB per launch).  GPUMT_LZ4_ENC=3 / gpumt_set_variant("lz4_enc", 3) selects lz4_enc3.hip.
 */
/* (measured and not kept: a 512-byte input ring refilled 256 bytes at a time -- IRING 512, IPIECE 256, IAHEAD 160, the match
 * window without its slack: 9 632 bytes of LDS = 17 chunk-waves per CU instead of 16 -- 107.1 ms against 105.4: the refills
 * and the match-window fetches of the smaller ring cost more than the 17th wave returns) */
#include "lz4_enc_shared.h"

#ifndef ENC5_TAIL
#define ENC5_TAIL 3u /* a search that would start in the last ENC5_TAIL lanes of its window opens a new window instead */
#endif
#ifndef ENC5_KMAX
#define ENC5_KMAX 64u /* a search that leaves its window goes on in a new window if its next probe is number <= ENC5_KMAX, else in probe batches */
#endif
#define E5_NONE 64u
#define E5_FWD 20u /* bytes behind a match's first four that a lane compares by itself */
static_assert(BM_BITS >= 2048u, "two folded filters of 1024 bits");

#ifdef ZMT_EMU
/* developer statistics of the emulator build (tools/emu_enc5_stats.py): lane 0 counts */
extern "C" { unsigned long long zmt_e5_stat[16]; }
#define E5_STAT(i, n) do { if (lane == 0) zmt_e5_stat[i] += (n); } while (0)
#else
#define E5_STAT(i, n) do { } while (0)
#endif

enum { E5_NEXT = 0, E5_LAST = 1, E5_DONE = 2, E5_FAIL = 3, E5_SLOW = 4 };

/* the parse state of a block */
struct Enc5St {
	u32 ip, anchor, op, nsq;
	u32 cbase; /* number of the sequence in lane 0 of the collecting registers: sequence n waits in lane n - cbase */
	Seq3 sq;
};

/* ---------------- extend the match both ways, then the sequence into the collecting registers (lz4_enc3.hip's code) ----
 * in: st.ip = the position whose four bytes matched those at `match`; quick (bit 31) = catch-up and length settled by the
 * probe's own loads.  out: st.ip = st.anchor = the position behind the match.  false: the block does not fit (stored raw) */
template <int TM>
static __device__ __forceinline__ bool e5_finish(Enc5St &st, InRing &R, u32 match, u32 quick, u32 low, u32 matchlimit,
						  u32 cap, const u8 *chunk, u8 *dst, int lane)
{
	u32 ip = st.ip;
	const u32 anchor = st.anchor;
	u32 fwd; /* equal bytes following the 4 that matched at ip */
	if (quick >> 31) {
		const u32 back = (quick >> 8) & 0xFFu;
		fwd = (quick & 0xFFu) + back;
		ip -= back;
		match -= back;
	} else {
		ring_want(R, ip, lane); /* [ip, ip + 68) resident */
		mside_prepare(R, match, lane);
		u32 room = ip - anchor;
		if (match - low < room)
			room = match - low;
		const u32 nb = room < 64 ? room : 64;
		const u32 flimit = matchlimit - (ip + MINMATCH);
		bool eqb = false, stopf = true;
		if ((u32)lane < nb) {
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
		ip -= back;
		match -= back;
		fwd += back;
	}
	/* ---------------- the sequence: its place in the output, its numbers into the collecting registers ---------------- */
	const u32 lit = ip - anchor, mc = fwd;
	const u32 op = st.op;
	u32 adv = lit + 3u;
	if (E_RARE((lit | mc) >= 15u))
		adv += (lit >= 15 ? (lit - 15) / 255 + 1 : 0u) + (mc >= 15 ? (mc - 15) / 255 + 1 : 0u);
	/* the reference's two output-limit tests, one compare in the common case (lz4_enc3.hip) */
	if (E_RARE(op + adv + 6u > cap)) {
		const u32 o1 = op + 1u;
		if (o1 + lit + (2 + 1 + LASTLITERALS) + lit / 255 > cap)
			return false;
		const u32 o2 = o1 + (lit >= 15 ? (lit - 15) / 255 + 1 : 0u) + lit;
		if (o2 + 2 + (1 + LASTLITERALS) + (mc + 240) / 255 > cap)
			return false;
	}
	st.op = op + adv;
	{
		const bool me = (u32)lane == st.nsq - st.cbase;
		st.sq.tok = me ? op : st.sq.tok;
		st.sq.src = me ? anchor : st.sq.src;
		st.sq.lit = me ? lit : st.sq.lit;
		st.sq.mc = me ? mc : st.sq.mc;
		st.sq.off = me ? ip - match : st.sq.off;
	}
	st.nsq++;
	ip += mc + MINMATCH;
	st.ip = ip;
	st.anchor = ip;
	if (st.nsq - st.cbase == 64u) {
		seq3_flush(st.sq, 64, chunk, dst, lane);
		st.cbase = st.nsq;
	}
	return true;
}

/* ---------------- a search that has left the consecutive part of its schedule: probe batches of 64 (lz4_enc3.hip's loop
 * without the quick extension).  Probe k of the search is at probe_pos3(ip0, k); kbase = the next probe's number.
 * true: *ipm / *match = the first probe that verifies and its candidate (insertions up to it are committed);
 * false: the search ran into the block's end (every valid probe is inserted) */
template <int TM>
static __device__ bool e5_search_batches(u32 *tlo, u32 *thi, u32 *bitmap, InRing &R, u32 ip0, u32 kbase, u32 mflimit_p1,
					 bool second, u32 *ipm, u32 *match, int lane)
{
	const u8 *chunk = R.chunk;
	for (;;) {
		u32 gap = 1;
		const u32 cur = probe_pos3(ip0, kbase + (u32)lane, &gap);
		const bool valid = cur + gap <= mflimit_p1;
		const u64 vm = wv_ballot(valid);
		if (vm == 0)
			return false;
		ring_want(R, wv_readlane(cur, 0), lane);
		const u64 x = valid ? in_ld64(R, cur) : 0;
		const u32 h = hash3<TM>(TM == T_U16 ? (u64)(u32)x : x);
		u32 cand = t_read<TM>(tlo, thi, h);
		u32 prev_dup = 64, next_dup = 64;
		u32 dold = 0;
		if (valid)
			dold = lds_or(&bitmap[(h & (BM_BITS - 1)) >> 5], 1u << (h & 31));
		const bool any_dup = wv_any(((dold >> (h & 31)) & 1) != 0);
		wv_sync();
		if (valid)
			bitmap[(h & (BM_BITS - 1)) >> 5] = 0;
		if (any_dup) {
			for (u32 i = 0; i < 64; i++) {
				const u32 hi_ = wv_readlane(h, (int)i);
				const bool vi = (vm >> i) & 1;
				if (vi && valid && hi_ == h) {
					if (i < (u32)lane)
						prev_dup = i;
					else if (i > (u32)lane && next_dup == 64)
						next_dup = i;
				}
			}
			const u32 pc = wv_shfl(cur, (int)(prev_dup & 63));
			if (prev_dup < 64)
				cand = pc;
		}
		const bool probe = valid && ((TM == T_U16) || cand + DIST_MAX >= cur);
		const u32 v4 = ld32u(chunk + (probe ? cand : 0u));
		const u64 mm = wv_ballot(probe && v4 == (u32)x);
		if (mm) {
			const u32 jstar = (u32)wv_ffs(mm) - 1;
			if ((u32)lane <= jstar && !(next_dup <= jstar))
				t_write<TM>(tlo, thi, h, cur, second);
			wv_sync();
			*ipm = wv_readlane(cur, (int)jstar);
			*match = wv_readlane(cand, (int)jstar);
			return true;
		}
		if (valid && next_dup == 64)
			t_write<TM>(tlo, thi, h, cur, second);
		wv_sync();
		if ((u32)wv_popc(vm) < 64)
			return false;
		kbase += 64;
	}
}

template <int TM, bool PROF>
static __device__ u32 encode_block5(u32 *tlo, u32 *thi, u32 *bitmap, InRing &R, u32 pos, u32 len, u8 *dst,
				    u32 cap, int lane)
{
	const u8 *chunk = R.chunk;
	const u32 iend = pos + len;
	const u32 mflimit_p1 = iend - MFLIMIT + 1;
	const u32 matchlimit = iend - LASTLITERALS;
	const u32 low = (TM == T_U16) ? pos : 0;
	const bool second = (pos >> 16) != 0; /* T_P17: the 17th bit of every position of this block */
	Enc5St st;
	st.ip = pos;
	st.anchor = pos;
	st.op = 0;
	st.nsq = 0;
	st.cbase = 0;
	st.sq = Seq3{0, 0, 0, 0, 0};
	u32 rmode = 0; /* 1: the next probe is the re-match probe at ip, behind T[h(ip - 2)] = ip - 2 */
	u32 ip0;       /* position of probe 0 of the search in progress */

	if (len < MFLIMIT + 1)
		goto last_literals;

	ring_want(R, st.ip, lane);
	{
		const u64 x = wv_readfirst((u32)in_ld64(R, st.ip)) | (u64)wv_readfirst((u32)(in_ld64(R, st.ip) >> 32)) << 32;
		if (lane == 0)
			t_write<TM>(tlo, thi, hash3<TM>(TM == T_U16 ? (u64)(u32)x : x), st.ip, second);
	}
	st.ip++;
	ip0 = st.ip;

	for (;;) {
		/* ======================= one window: lane j = position w0 + j ======================= */
		const u32 w0 = st.ip;
		if (w0 >= mflimit_p1) /* not even the first probe is allowed: the block's last literals */
			goto last_literals;
		if (rmode)
			ip0 = w0 + 1; /* lane 0 is the re-match probe, probe 0 of the search behind it is at w0 + 1 */
		/* a window yields at most 16 sequences: room for them in the collecting registers.  The flush rides on the window's
		 * memory round trip: its literals are asked for with the candidates, its stores leave once those have landed */
		const bool fl = st.nsq - st.cbase > 48u;
		const u32 jend = mflimit_p1 - w0 < 64u ? mflimit_p1 - w0 : 64u; /* lanes whose position may be probed: cur + 1 <= mflimit_p1 */
		const u32 cur = w0 + (u32)lane;
		const bool pvalid = (u32)lane < jend;
		ring_want(R, w0, lane);
		/* the position's neighbourhood [cur - 8, cur + 16) from the input ring: seven aligned dwords + funnel shifts
		 * (lz4_enc3.hip); [w0 - 256, w0 + 256) is resident */
		u64 x, xb, x1, x2; /* [cur - 8, cur), [cur, cur + 8), [cur + 8, cur + 16), [cur + 16, cur + 24) */
		{
			const u32 pb = cur - 8;
			const u32 *const w = (const u32 *)(R.ring + (pb & (IRING - 1) & ~3u)); /* + 36 <= IRING + IMIRROR */
			const u32 w0_ = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5], w6 = w[6], w7 = w[7], w8 = w[8];
			xb = (u64)wv_alignbyte(w1, w0_, pb) | (u64)wv_alignbyte(w2, w1, pb) << 32;
			x = (u64)wv_alignbyte(w3, w2, pb) | (u64)wv_alignbyte(w4, w3, pb) << 32;
			x1 = (u64)wv_alignbyte(w5, w4, pb) | (u64)wv_alignbyte(w6, w5, pb) << 32;
			x2 = (u64)wv_alignbyte(w7, w6, pb) | (u64)wv_alignbyte(w8, w7, pb) << 32;
		}
		const u32 h = hash3<TM>(TM == T_U16 ? (u64)(u32)x : x);
		if (rmode) {
			/* T[h(w0 - 2)] = w0 - 2 precedes the re-match lookup; its eight bytes are the last two of xb and six of x */
			const u64 xm2 = (xb >> 48) | (x << 16);
			const u32 h2 = hash3<TM>(TM == T_U16 ? (u64)(u32)xm2 : xm2);
			if (lane == 0)
				t_write<TM>(tlo, thi, h2, w0 - 2, second);
		}
		wv_sync();
		const u32 cand0 = t_read<TM>(tlo, thi, h); /* the table at the window's start */
		/* the candidate's neighbourhood [cand - 8, cand + 24) from memory, 16 + 16 bytes (lz4_enc3.hip fetched 16 + 8: the same
		 * two instructions and, seven times of eight, the same line; 20 bytes behind the match's first four settle 60 % of the
		 * matches the 12 left open); lanes without a candidate read chunk[0, 16) */
		const bool dist_ok = (TM == T_U16) || cand0 + DIST_MAX >= cur;
		const bool probe = pvalid && dist_ok;
		const bool wide = probe && cand0 >= 8;
		u64 l0, l1, l2, l3;
		{
			const u32 a0 = !probe ? 0u : wide ? cand0 - 8 : cand0;
			const u8 *gp = chunk + a0;
			ENC3_LD16(gp, l0, l1);
			ENC3_LD16(wide ? gp + 16 : gp, l2, l3);
		}
		u64 fq0 = 0, fq1 = 0;
		if (fl)
			seq3_flush_ask(st.sq, st.nsq - st.cbase, chunk, lane, fq0, fq1);
		/* ---- positions of the window with the same hash: prev = the nearest earlier one (E5_NONE: none).  ONE filter of 2 048 bits
		 * on the hash's low 11 bits, behind the loads: the returning atomic tells every lane but the first of a bit that the bit
		 * was taken ("late": one lane per window on average, half of them chance meetings on 11 bits); for a late lane the set of
		 * lanes with its hash is one ballot, and every lane of the set finds its nearest earlier member by itself -- pairs and
		 * longer chains alike.  (Before: two folded filters of 1 024 bits -- a second atomic, a read, a barrier -- and a scalar
		 * pass over both lanes of every pair the fold threw together: four lanes per window on average) ---- */
		/* (ballots below are taken of ONE compare each and combined as scalar masks: a ballot of `a && b` compiles to mask algebra
		 * plus a select and a second compare that re-materialise the mask it already has) */
		const u64 vmask = jend >= 64u ? ~0ull : (1ull << jend) - 1ull; /* lanes whose position may be probed */
		u32 prev = E5_NONE;
		u64 dmask = 0;
		{
			const u32 fw = (h & 2047u) >> 5, fb = 1u << (h & 31u);
			u64 late;
			/* (every lane of a window is a position that may be probed except at a block's end: the common case has no
			 * exec-mask region around the filter's steps) */
			if (E_RARE(jend < 64u)) {
				u32 o1_ = 0;
				if (pvalid)
					o1_ = lds_or(&bitmap[fw], fb);
				late = wv_ballot((o1_ & fb) != 0); /* (0 for a lane without a position; the emulator's lanes meet here: every atomic is made before a bit is cleared) */
				if (pvalid)
					bitmap[fw] = 0;
			} else {
				const u32 o1_ = lds_or(&bitmap[fw], fb);
				late = wv_ballot((o1_ & fb) != 0);
				bitmap[fw] = 0;
			}
			while (late) {
				const int i = wv_ffs(late) - 1;
				const u32 hi_ = wv_readlane(h, i);
				const u64 E = wv_ballot(h == hi_) & vmask;
				late &= ~E;
				const u64 below = E & ((1ull << (u32)lane) - 1ull);
				if (((E >> (u32)lane) & 1ull) && below != 0)
					prev = 63u - (u32)__builtin_clzll(below);
			}
			dmask = wv_ballot(prev != E5_NONE);
		}
		/* (every load of the window has landed before the flush's first store leaves: see vm_landed) */
		vm_landed<false>(l0, l1);
		vm_landed<false>(l2, l3);
		if (fl) {
			vm_landed<true>(fq0, fq1);
			seq3_flush_put(st.sq, st.nsq - st.cbase, chunk, dst, lane, fq0, fq1);
			st.cbase = st.nsq;
		}
		/* ---- every lane: does its candidate verify, and how far do the 12 bytes behind / the 8 in front agree ---- */
		const u64 g0 = l0, g1 = wide ? l1 : l0, g2 = l2, g3 = l3;
		const bool ver0 = probe && (u32)g1 == (u32)x;
		u32 eqf, eqb_raw;
		bool qstat;
		{
			/* fwd: equal bytes behind the 4, E5_FWD = 20 looked at; back: equal bytes in front, 8 looked at.  Settled when a
			 * difference (or the limit) lies inside them */
			const u32 flimit = matchlimit - (cur + MINMATCH);
			const u32 da = (u32)(x >> 32) ^ (u32)(g1 >> 32);
			const u64 db_ = x1 ^ g2, dc = x2 ^ g3;
			/* the first differing bit of the 4 + 8 + 8 bytes as a chain of minima: the three-way choice compiled to two nested
			 * exec-mask regions -- a dozen scalar instructions per window on the pipe this kernel saturates */
			{
				const u32 za = da ? (u32)__builtin_ctz(da) : 255u; /* (255: "none here", above every sum below) */
				const u32 zb = db_ ? (u32)__builtin_ctzll(db_) : 255u;
				const u32 zc = dc ? (u32)__builtin_ctzll(dc) : 64u;
				const u32 zc2 = 64u + zc, zbc = 32u + (zb < zc2 ? zb : zc2);
				const u32 z = za < zbc ? za : zbc;
				eqf = z >> 3; /* 4 + 8 + 8 equal bytes: 160 >> 3 = E5_FWD */
			}
			const bool fdec = eqf < E5_FWD || flimit <= E5_FWD;
			if (eqf > flimit)
				eqf = flimit;
			const u64 db = xb ^ g0;
			eqb_raw = ((u32)__builtin_clzll(db | 1ull) + (db == 0)) >> 3;
			qstat = wide && cur >= 8 && fdec;
		}
		const u32 room_m = cand0 - low;

		/* ======================= the reference's parse through the window ======================= */
		E5_STAT(0, 1);
		E5_STAT(1, dmask != 0);
		/* twins (two of 64 positions share one of 4 096 table entries in half of all windows): what a lane's probe finds
		 * when its twin was inserted before it -- the twin's position; does it verify? -- is settled here, once */
		bool tver = false, deep = false;
		u64 TV = 0; /* lanes that verify against their twin */
		if (dmask != 0) {
			const u32 xp = wv_shfl((u32)x, (int)(prev & 63u));
			const u32 pp = wv_shfl(prev, (int)(prev & 63u));
			tver = prev != E5_NONE && pvalid && (u32)x == xp; /* (less than 64 bytes back: inside the distance limit) */
			TV = wv_ballot((u32)x == xp) & dmask & vmask;
			deep = wv_any(pp != E5_NONE); /* three positions with one hash: the general walk below (a lane without a twin reads lane 0's, which has none) */
		}
		/* the lanes whose candidate verifies: the four bytes agree, the position may be probed, the candidate is within reach */
		const u64 V0 = wv_ballot((u32)g1 == (u32)x) & vmask & (TM == T_U16 ? ~0ull : wv_ballot(cand0 + DIST_MAX >= cur));
		/* the lanes the search in progress may probe: while its probes are consecutive positions (probe k at ip0 + k for
		 * k <= 64); a search that starts inside the window ends behind the window */
		u64 lmask = vmask;
		u32 klim = ip0 + 65u - w0;
		if (E_RARE(klim < 64u))
			lmask &= (1ull << klim) - 1ull;
		/* ---- the chain of easy sequences, every lane as the start of a search (ENC5_VEC): the search from lane t finds its
		 * match at lane mt = the first verifying lane from t on, which ends at lane t_next = mt + 4 + (its forward count).
		 * Easy = the match lane's own loads settle both extensions and no twin that might verify lies in [t, mt] (those
		 * searches, and the ones whose match needs the general extension, go one at a time through the code below).  Once
		 * per window: mt, the match lane's numbers (one ds_bpermute each), t_next ---- */
		/* twins whose candidate depends on the parse: those that verify against their twin or against the table's entry;
		 * with three positions of one hash in the window the twin compared here need not be the one that counts: all of them */
		const u64 smask = deep ? dmask : dmask & (TV | V0);
		u32 v_m, v_pk, v_cd;
		bool v_none, v_twin;
		{
			const u64 vr = (V0 & vmask) >> (u32)lane;
			const u64 sr = smask >> (u32)lane;
			const u32 mrel = vr ? (u32)__builtin_ctzll(vr) : 64u;
			const u32 srel = sr ? (u32)__builtin_ctzll(sr) : 64u;
			v_none = mrel == 64u;
			v_twin = srel <= mrel;
			v_m = ((u32)lane + mrel) & 63u;
			/* what a search start needs of its match lane: forward count (5 bits) | backward count << 5 | min(room of the
			 * candidate, 9) << 9 | "settled by its own loads" << 13 */
			const u32 pk = eqf | eqb_raw << 5 | (room_m < 9u ? room_m : 9u) << 9 | (qstat ? 1u << 13 : 0u);
			v_pk = wv_shfl(pk, (int)v_m);
			v_cd = wv_shfl(cand0, (int)v_m);
		}
		/* a search start's link for the marking loop, as far as the window knows it: the lane behind its match | 0x40: the run's
		 * last (the match ends the block, or the next search would start in the window's tail) | 0x80: not in a run */
		u32 code_w;
		{
			const u32 tn = v_m + 4u + (v_pk & 31u);
			/* (measured and not kept: runs marked as if no twin mattered and cut at the first search a twin's insertion proves wrong
			 * -- 24 % fewer one-at-a-time sequences, 103.3-103.8 ms against 101.9: the check costs more than the quick serial
			 * sequences it saves; profiles/r06_sweeps/lz4_enc5_steps.txt) */
			const bool out = v_none | v_twin | (((v_pk >> 13) & 1u) == 0u);
			const bool fin = (tn + ENC5_TAIL > 64u) | (w0 + tn >= mflimit_p1);
			code_w = out ? 0x80u : fin ? 0x40u : tn;
		}
		u64 I = 0; /* lanes whose position the parse inserted */
		u32 s = 0; /* lane of the search's next probe */
		u32 wend = E5_NEXT;
		for (;;) {
#ifndef ENC5_NOVEC
			/* ================= a run of easy sequences from lane s on, lane-parallel ================= */
			{
				/* the sequence of the search from lane t: catch-up is bounded by the literals in front -- the block's
				 * anchor for the run's first search, its own start for the others -- and by the candidate's room */
				const u32 anchor_t = (u32)lane == s ? st.anchor : cur;
				const u32 mpos = w0 + v_m;
				const u32 room_ip = mpos - anchor_t;
				const u32 r9 = (v_pk >> 9) & 15u;
				const u32 nb = room_ip < r9 ? room_ip : r9;
				const u32 ebr = (v_pk >> 5) & 15u;
				const bool bdec = nb == 0 || ebr < 8u || nb <= 8u;
				const u32 eqb = ebr > nb ? nb : ebr;
				const u32 t_next = v_m + 4u + (v_pk & 31u);
				/* (a search that began in front of the window must find its match while its probes are consecutive
				 * positions: the lane behind them is klim, 64 and more for every other search) */
				/* below 64: the run goes on at that lane; 0x40: this search is the run's last (its match ends the block, or the
				 * next search would start in the window's tail); 0x80: not in the run (no match in the window, or not easy).  One
				 * test per sequence in the loop; starts come in lane order, so the last one is A's highest bit */
				/* (what does not depend on the run's first lane is the window's: code_w; the catch-up is undecided iff all 8 bytes
				 * in front agree and 9 may be taken -- nb <= 9, ebr <= 8 -- one add and one compare instead of mask algebra on the
				 * scalar pipe) */
				(void)bdec;
				const bool hardx = (nb + ebr >= 17u) | (((u32)lane == s) & (v_m >= klim));
				const u32 code = hardx ? 0x80u : code_w;
				u64 A = 0;
				u32 tl = s;
				{
					u32 t = s, c = wv_readlane(code, (int)s);
					while (c < 0x40u) {
						A |= 1ull << t;
						t = c;
						c = wv_readlane(code, (int)t);
					}
					if (c == 0x40u)
						A |= 1ull << t;
				}
				/* the lanes the run's searches probe: from every start to its match */
				u64 P = 0;
				if (A != 0) {
					const u64 upto = A & ((2ull << (u32)lane) - 1ull);
					const u32 ts = 63u - (u32)__builtin_clzll(upto | 1ull); /* the last start at or in front of this lane */
					const u32 mts = wv_shfl(v_m, (int)ts);
					P = wv_ballot((u32)lane <= mts) & ~((A & (0ull - A)) - 1ull); /* (from the run's first start on) */
					tl = 63u - (u32)__builtin_clzll(A);
				}
				if (A != 0) {
					const bool inA = (A >> (u32)lane) & 1ull;
					const u32 lit = mpos - eqb - anchor_t, mc = (v_pk & 31u) + eqb;
					const u32 el = (lit + 240u) / 255u; /* = lit >= 15 ? (lit - 15) / 255 + 1 : 0, without the exec-mask region */
					const u32 adv = inA ? lit + 3u + el + (mc >= 15u ? 1u : 0u) : 0u;
					const u32 incl = wv_scan_incl(adv);
					const u32 opt = st.op + incl - adv;
					const u32 run_out = wv_readlane(incl, 63);
					/* the output limit: one compare per sequence covers both of the reference's tests (e5_finish), and the run's
					 * last sequence stands for all of them (the positions only grow); a run with a sequence near the limit is
					 * left to the code below, one search at a time */
					if (E_RARE(st.op + run_out + 6u > cap)) {
						A = 0;
					} else {
						const u32 nseq = (u32)wv_popc(A);
						/* into the collecting registers: the run's r-th sequence goes to lane (number - cbase); the lanes
						 * that start a search say where they are (window bytes of the match side as scratch) */
						if (inA)
							R.mwin[wv_mbcnt(A)] = (u8)lane;
						wv_sync();
						const u32 rd = (u32)lane - (st.nsq - st.cbase);
						const bool take = rd < nseq;
						const int from = (int)R.mwin[take ? rd : 0u];
						const u32 g_tl = wv_shfl(opt | lit << 16, from);
						const u32 g_om = wv_shfl((mpos - v_cd) | mc << 16, from);
						const u32 g_src = wv_shfl(anchor_t, from);
						st.sq.tok = take ? g_tl & 0xFFFFu : st.sq.tok;
						st.sq.lit = take ? g_tl >> 16 : st.sq.lit;
						st.sq.off = take ? g_om & 0xFFFFu : st.sq.off;
						st.sq.mc = take ? g_om >> 16 : st.sq.mc;
						st.sq.src = take ? g_src : st.sq.src;
						wv_sync();
						st.op += run_out;
						st.nsq += nseq;
						if (E_RARE(st.nsq - st.cbase == 64u)) {
							seq3_flush(st.sq, 64, chunk, dst, lane);
							st.cbase = st.nsq;
						}
						/* the run's insertions: every lane from a search's start to its match, and the position two
						 * in front of every search start but the run's first (ip - 2 behind a match) */
						I |= P | ((A & ~(1ull << s)) >> 2);
						/* behind the run's last match */
						const u32 e_last = wv_readlane(t_next, (int)tl);
						st.ip = w0 + e_last;
						st.anchor = st.ip;
						E5_STAT(10, nseq);
						if (E_RARE(st.ip >= mflimit_p1)) {
							wend = E5_DONE;
							break;
						}
						rmode = 1;
						if (st.ip + ENC5_TAIL > w0 + 64u)
							break; /* a new window at ip (its set-up makes the insertion of ip - 2) */
						s = e_last;
						I |= 1ull << (s - 2u);
						ip0 = st.ip + 1u;
						lmask = vmask;
						klim = 128u;
					}
				}
			}
#endif
			/* ================= one search from lane s, the general way ================= */
			const u64 range = lmask & ~((1ull << s) - 1ull);
			u64 mm;
			u32 q = E5_NONE;
			E5_STAT(2, 1);
			if ((dmask & range) == 0) {
				mm = V0 & range;
			} else if (!deep) {
				/* a probe sees the latest insertion of its hash: its twin's position if that was inserted -- a probe of this
				 * search (every lane from s on) or a committed one -- else what the table held at the window's start */
				E5_STAT(3, 1);
				const bool pin = prev != E5_NONE && (prev >= s || ((I >> (prev & 63u)) & 1ull));
				q = pin ? prev : E5_NONE;
				mm = wv_ballot(pin ? tver : ver0) & range;
			} else {
				E5_STAT(9, 1);
				/* chains of three and more: walk to the nearest earlier position with the same hash that was inserted */
				q = prev;
				for (;;) {
					const bool skip = q != E5_NONE && q < s && !((I >> (q & 63u)) & 1ull);
					if (!wv_any(skip))
						break;
					const u32 pq = wv_shfl(prev, (int)(q & 63u));
					if (skip)
						q = pq;
				}
				const u32 xq = wv_shfl((u32)x, (int)(q & 63u));
				mm = wv_ballot(q != E5_NONE ? (pvalid && (u32)x == xq) : ver0) & range;
			}
			if (mm == 0) {
				E5_STAT(8, 1);
				I |= range; /* every probe of the range was made, and inserted */
				if (lmask == vmask && jend < 64u) {
					wend = E5_LAST; /* the search ran into the block's end */
					break;
				}
				const u32 lim = lmask == ~0ull ? 64u : (u32)wv_ffs(~lmask) - 1u;
				st.ip = w0 + lim; /* the search's next probe */
				rmode = 0;
				if (lim < 64u || st.ip - ip0 > ENC5_KMAX)
					wend = E5_SLOW;
				break;
			}
			const u32 m = (u32)wv_ffs(mm) - 1u;
			I |= range & ((2ull << m) - 1ull);
			u32 match, quick = 0;
			const u32 qm = wv_readlane(q, (int)m);
			if (E_RARE(qm != E5_NONE)) {
				E5_STAT(4, 1);
				match = w0 + qm; /* a position of this window: the general extension reads it from the ring */
			} else {
				/* catch-up is bounded by the literals in front (anchor) and by the candidate's room */
				const u32 room_ip = cur - st.anchor;
				const u32 nb = room_ip < room_m ? room_ip : room_m;
				const bool bdec = nb == 0 || eqb_raw < 8u || nb <= 8u;
				const u32 eqb = eqb_raw > nb ? nb : eqb_raw;
				const u32 qi = (qstat && bdec) ? 0x80000000u | eqb << 8 | eqf : 0u;
				match = wv_readlane(cand0, (int)m);
				quick = wv_readlane(qi, (int)m);
#ifdef ZMT_EMU
				{
					const u32 s11 = wv_readlane((u32)(!qstat), (int)m), s12 = wv_readlane((u32)(qstat && !bdec), (int)m);
					const u32 s13 = wv_readlane((u32)(!wide || cur < 8), (int)m);
					E5_STAT(11, s11);
					E5_STAT(12, s12);
					E5_STAT(13, s13);
				}
#endif
			}
			E5_STAT(5, 1);
			E5_STAT(6, (quick >> 31) == 0);
			st.ip = w0 + m;
			if (E_RARE(!e5_finish<TM>(st, R, match, quick, low, matchlimit, cap, chunk, dst, lane))) {
				wend = E5_FAIL;
				break;
			}
			if (E_RARE(st.ip >= mflimit_p1)) {
				wend = E5_DONE;
				break;
			}
			/* next: T[h(ip - 2)] = ip - 2, the re-match probe at ip, the search behind it */
			rmode = 1;
			if (st.ip + ENC5_TAIL > w0 + 64u)
				break; /* a new window at ip (its set-up makes the insertion) */
			s = st.ip - w0;
			I |= 1ull << (s - 2u);
			ip0 = st.ip + 1u;
			lmask = vmask;
			klim = 128u;
		}
		/* ---- the window's insertions, later positions over earlier ones: lanes without an earlier twin first (no two of
		 * them share an entry), then those with one, in order ---- */
		{
			const bool ins = (I >> lane) & 1ull;
			if (ins && !((dmask >> lane) & 1ull))
				t_write<TM>(tlo, thi, h, cur, second);
			u64 t = dmask & I;
			while (E_RARE(t != 0)) {
				const int i = wv_ffs(t) - 1;
				t &= t - 1;
				wv_sync();
				if (lane == i)
					t_write<TM>(tlo, thi, h, cur, second);
			}
			wv_sync();
		}
		if (E_RARE(wend != E5_NEXT)) {
			if (wend == E5_FAIL)
				return 0;
			if (wend == E5_LAST)
				goto last_literals;
			if (wend == E5_DONE)
				goto block_done;
			E5_STAT(7, 1);
			/* E5_SLOW: the search goes on in probe batches */
			u32 ipm, match;
			if (!e5_search_batches<TM>(tlo, thi, bitmap, R, ip0, st.ip - ip0, mflimit_p1, second, &ipm, &match, lane))
				goto last_literals;
			st.ip = ipm;
			if (!e5_finish<TM>(st, R, match, 0u, low, matchlimit, cap, chunk, dst, lane))
				return 0;
			if (st.ip >= mflimit_p1)
				goto block_done;
			rmode = 1;
		}
	}
block_done:
last_literals:
	if (st.nsq != st.cbase) /* (a block that fails below is stored raw: writing what it had is harmless) */
		seq3_flush(st.sq, st.nsq - st.cbase, chunk, dst, lane);
	{
		u32 op = st.op;
		const u32 run = iend - st.anchor;
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
		copy_literals(R, dst + op, st.anchor, run, lane);
		op += run;
		return op;
	}
}

struct Enc5 {
	template <int TM, bool PROF>
	static __device__ __forceinline__ u32 block(u32 *tlo, u32 *thi, u32 *bitmap, InRing &R, u32 pos, u32 len, u8 *dst, u32 cap, int lane)
	{
		return encode_block5<TM, PROF>(tlo, thi, bitmap, R, pos, len, dst, cap, lane);
	}
};

#define ENC5_KERNEL(NAME, TM, TABBYTES)                                                           \
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
		enc_frame_body<TM, false, Enc5>(tlo, thi, bitmap, ring, mwin, in, n, chunk, rec0, nrec, slots, \
			      slot_stride, rec_len, chk, prof);                                                     \
	}

ENC5_KERNEL(zmt_lz4_enc5_u16_kernel, T_U16, 16384)
ENC5_KERNEL(zmt_lz4_enc5_p17_kernel, T_P17, 8192)
ENC5_KERNEL(zmt_lz4_enc5_u32_kernel, T_U32, 16384)

