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
#include "lz4_common.h"

#define MINMATCH 4u
#define MFLIMIT 12u
#define LASTLITERALS 5u
#define DIST_MAX 65535u
#ifndef IRING
#define IRING 1024u /* input ring bytes (power of two): look-ahead + recent history that serves near candidates */
#endif
#ifndef BM_BITS
#define BM_BITS 2048u /* bits of the in-batch duplicate filter (a power of two >= 1024): hashes are folded onto
                       * it, a false "duplicate" only sends the batch through the exact readlane loop */
#endif
/* probes of a search's first batch (+ the re-match probe) and of its second; the later ones take 32 and 64.  Most matches
 * sit within the first few probes, and every probe is a candidate fetched from memory: [MI355X, 8 GiB] 4 + 12: 308.8 ms,
 * 8 + 16: 288.7, 10 + 16: 287.6, 12 + 16: 290.1, 16 + 16: 304.4 (rounds 1-3), 24 + 24: 369.2, 32 + 32: 479.8
 * (profiles/r04_sweeps/lz4_enc3_steps.txt).  The split does not change what is found, only in how many steps */
#ifndef ENC3_B0
#define ENC3_B0 10u
#define ENC3_B1 16u
#endif

#define IPIECE 512u /* refill granule: 8 bytes per lane */
/* bytes ring_want() makes resident ahead of a position: >= 72 (a batch of 64 consecutive probes reads
 * 8 bytes each) and small enough that IRING - IAHEAD - IPIECE >= 64 bytes of history stay behind it
 * (catch-up compares 64 bytes backwards, the re-match step reads ip - 2) */
#define IAHEAD (IRING >= 2048u ? 512u : 256u)
static_assert(IRING >= IAHEAD + IPIECE + 64u, "the input ring must keep 64 bytes of history");
#define IMIRROR 32u /* the search reads 24 bytes from one wrapped address */
/* uniform branches are what a single wave pays most for: keep the common path falling through */
#define E_RARE(c) __builtin_expect(!!(c), 0)

enum { T_U16 = 0, T_P17 = 1, T_U32 = 2 };
#ifdef ZMT_EMU
#include <assert.h>
#define E_ASSERT(c) assert(c)
#else
#define E_ASSERT(c) do { } while (0)
#endif

#ifdef ZMT_EMU
static inline u32 lds_or(u32 *p, u32 v)
{
	u32 o = *p;
	*p = o | v;
	return o;
}
#else
static __device__ __forceinline__ u32 lds_or(u32 *p, u32 v) { return atomicOr(p, v); }
#endif

template <int TM> static __device__ __forceinline__ u32 hash3(u64 x)
{
	if (TM == T_U16)
		return ((u32)x * 2654435761u) >> 19;
	return (u32)(((x << 24) * 889523592379ULL) >> 52);
}

/* ---- hash table, three storage layouts; `lo` is the main array, `hi` the 17th bits (T_P17) ---- */
/* (measured and dropped in round 5, both bit-identical: not reading the 17th-bit plane in a chunk's first block, where no entry
 * has it set -- 259.2 -> 263.4 ms per 8 GiB, the wave-uniform branch costs more than the read --, and one way out of the batch
 * loop through a `found` flag instead of the jumps to last_literals: 263.4 -> 265.1) */
template <int TM> static __device__ __forceinline__ u32 t_read(const u32 *lo, const u32 *hi, u32 h)
{
	if (TM == T_U32)
		return lo[h];
	u32 v = ((const u16 *)lo)[h];
	if (TM == T_P17)
		v |= ((hi[h >> 5] >> (h & 31)) & 1) << 16;
	return v;
}
/* per-lane store; several lanes may hit different entries of one hi word -> an atomic there.  Positions only grow
 * inside a chunk and the table starts zeroed, so the 17th bit of an entry only ever goes 0 -> 1: the first 64 KiB
 * block writes none, the second block ORs (`second` = this block starts at 64 KiB, wave-uniform) -- no AND, and no
 * per-lane choice between two atomics */
template <int TM> static __device__ __forceinline__ void t_write(u32 *lo, u32 *hi, u32 h, u32 v, bool second)
{
	if (TM == T_U32) {
		lo[h] = v;
		return;
	}
	((u16 *)lo)[h] = (u16)v;
	if (TM == T_P17) {
		E_ASSERT((v >> 16) == (second ? 1u : 0u));
		if (second)
			lds_or(&hi[h >> 5], 1u << (h & 31));
	}
}

/* probe k of a search that starts at ip: position and the gap to probe k+1 (see lz4_enc.hip) */
static __device__ __forceinline__ u32 probe_pos3(u32 ip, u32 k, u32 *gap)
{
	if (k <= 64) {
		*gap = 1;
		return ip + k;
	}
	const u32 t = k - 65, q = t >> 6, r = t & 63;
	*gap = q + 2;
	return ip + 65 + 64 * (q * (q + 3) / 2) + (q + 2) * r;
}

static __device__ __forceinline__ u32 put_len_ext3(u8 *op, u32 rem, int lane)
{
	u32 n255 = rem / 255;
	for (u32 i = (u32)lane; i < n255; i += 64)
		op[i] = 255;
	if (lane == 0)
		op[n255] = (u8)(rem - n255 * 255);
	return n255 + 1;
}

/* ---- input ring: ring[p & (IRING-1)] = chunk[p] for p in [rhi - IRING, rhi) ---- */
struct InRing {
	u8 *ring;
	const u8 *chunk;
	u32 rlo;     /* ring is valid for chunk positions [max(rlo, rhi - IRING), rhi) */
	u32 rhi;     /* end of what has been loaded (multiple of IPIECE) */
	u32 limit;   /* never load at or beyond this chunk position (readable bytes of the input) */
	u8 *mwin;    /* 128-byte window of the match side: mwin[i] = chunk[mbase + i] */
	u32 mbase;
	u64 pc[8];   /* phase cycle counters: only the _prof instantiation of the kernel touches them (a run-time
	              * flag cost every sequence eight scalar compare + branch pairs and 18 SGPRs) */
	u64 tq;
};
#ifndef ZMT_EMU
#define EPC(R, i) do { if (PROF) { u64 t_ = (u64)clock64(); (R).pc[i] += t_ - (R).tq; (R).tq = t_; } } while (0)
#else
#define EPC(R, i) do { } while (0)
#endif

/* make [pos, pos + IAHEAD) (clipped to the input) resident */
static __device__ __forceinline__ void ring_want(InRing &R, u32 pos, int lane)
{
	const u32 want_hi = pos + IAHEAD;
	R.rhi = wv_readfirst(R.rhi); /* wave-uniform by construction; said so that the tests below are scalar */
	R.rlo = wv_readfirst(R.rlo);
	if (want_hi <= R.rhi) /* the common case first: one scalar compare */
		return;
	if (want_hi > R.rhi + 2 * IRING) {
		/* far jump (long literal run or match): restart the ring one piece in front of the new position, so that
		 * the invariant of every other call -- [pos - 256, pos + IAHEAD) resident, clipped at the chunk's start
		 * -- holds here too and the encoder's fast paths need no test on rlo */
		const u32 b = pos & ~(IPIECE - 1);
		R.rhi = b >= IPIECE ? b - IPIECE : 0;
		R.rlo = R.rhi;
	}
	/* (the piece count is fixed before the loop and the tail is read without a per-lane branch: with the exit test
	 * inside, the compiler threaded the lanes' `p + 8 <= limit` into the loop exit, which made rhi -- and with it
	 * every test on the ring's state in the encoder -- lane-varying: exec-mask code around each of them) */
	const u32 stop = want_hi < R.limit ? want_hi : R.limit;
	const u32 npiece = stop > R.rhi ? (stop - R.rhi + IPIECE - 1) / IPIECE : 0;
	u32 rhi = R.rhi;
	for (u32 k = 0; k < npiece; k++) {
		const u32 p = rhi + 8u * (u32)lane;
		/* limit >= 8: a lane whose 8 bytes cross the limit reads the last 8 readable ones and shifts */
		const u32 pp = p + 8 <= R.limit ? p : R.limit - 8;
		const u32 sh = p - pp;
		u64 a = ld64u(R.chunk + pp);
		a = sh < 8 ? a >> (8 * (sh & 7)) : 0;
		wv_sync();
		u8 *d = R.ring + (p & (IRING - 1));
		*(u64 *)d = a;
		if ((p & (IRING - 1)) < IMIRROR) /* mirror: multi-byte reads never wrap */
			*(u64 *)(d + IRING) = a;
		wv_sync();
		rhi += IPIECE;
	}
	R.rhi = rhi;
}
/* is chunk[p .. p+n) readable from the ring? (n <= IMIRROR beyond a wrap) */
static __device__ __forceinline__ bool ring_has(const InRing &R, u32 p, u32 n)
{
	return p >= R.rlo && p + IRING >= R.rhi && p + n <= R.rhi;
}
static __device__ __forceinline__ u64 in_ld64(const InRing &R, u32 p)
{
	return ring_has(R, p, 8) ? ld64u(R.ring + (p & (IRING - 1))) : ld64u(R.chunk + p);
}
static __device__ __forceinline__ u32 in_ld8(const InRing &R, u32 p)
{
	return ring_has(R, p, 1) ? (u32)R.ring[p & (IRING - 1)] : (u32)R.chunk[p];
}

/* fetch chunk[m-32 .. m+96) (clipped at 0) into the match window: one global round trip that
 * serves the re-match test, catch-up (<= 32 bytes back) and the match count (<= 64 forward) */
#define MWIN 128u
static __device__ __forceinline__ void mwin_fetch(InRing &R, u32 m, int lane)
{
	const u32 base = m >= 32 ? m - 32 : 0;
	const u32 p = base + 2u * (u32)lane;
	u32 v = 0;
	if (p + 2 <= R.limit)
		v = ld16u(R.chunk + p);
	else if (p < R.limit)
		v = R.chunk[p];
	wv_sync();
	*(u16 *)(R.mwin + 2u * (u32)lane) = (u16)v;
	wv_sync();
	R.mbase = base;
}
/* make the match side around m readable from LDS: nothing to do when the input ring still holds
 * [m-64, m+132), else one global fetch into the window (wave-uniform decision) */
static __device__ __forceinline__ void mside_prepare(InRing &R, u32 m, int lane)
{
	const u32 lo = m >= 64 ? m - 64 : 0;
	if (ring_has(R, lo, m + 132 - lo)) {
		R.mbase = 0xFFFFFFFFu; /* window unused: m_ld8 goes to the ring */
		return;
	}
	mwin_fetch(R, m, lane);
}
/* chunk[p] of the match side: ring, else window, else memory */
static __device__ __forceinline__ u32 m_ld8(const InRing &R, u32 p)
{
	if (ring_has(R, p, 1))
		return (u32)R.ring[p & (IRING - 1)];
	return (p >= R.mbase && p - R.mbase < MWIN) ? (u32)R.mwin[p - R.mbase] : (u32)R.chunk[p];
}

/* forward compares right after ring_want(ip) / mside_prepare(match): the ip side [ip, ip + 68) is in
 * the ring and the match side [m, m + 68) in the ring or in the window (wave-uniform which), so the
 * per-lane residency tests of in_ld8 / m_ld8 are not needed */
static __device__ __forceinline__ u32 in_fwd8(const InRing &R, u32 p) { return (u32)R.ring[p & (IRING - 1)]; }
static __device__ __forceinline__ u32 m_fwd8(const InRing &R, u32 p)
{
	return R.mbase == 0xFFFFFFFFu ? (u32)R.ring[p & (IRING - 1)] : (u32)R.mwin[p - R.mbase];
}

/* cooperative literal copy chunk[a .. a+n) -> d, from the ring when it is there */
static __device__ __forceinline__ void copy_literals(const InRing &R, u8 *d, u32 a, u32 n, int lane)
{
	if (n <= 256 && ring_has(R, a, n)) {
		for (u32 i = (u32)lane; i < n; i += 64)
			d[i] = R.ring[(a + i) & (IRING - 1)];
	} else {
		wave_copy(d, R.chunk + a, n, lane);
	}
}

/* Sequences are not written out one by one -- the token, the literals, the offset and the length bytes of a sequence are four
 * to six dependent steps of scalar code, an LDS read and single-lane stores on the chunk's chain, and at 16 waves per CU every
 * instruction of a wave costs it 20-30 cycles -- but collect in registers (lane = sequence number mod 64: where its token
 * goes, where its literals come from, the three numbers) and leave 64 at a time, every lane writing its own sequence
 * [MI355X, 8 GiB: 288.0 -> 269.6 ms, with the merged limit test 259.2; profiles/r05_sweeps/lz4_enc3_steps.txt].  The output position is still advanced and
 * tested against the limit sequence by sequence, exactly as the reference does: a block that fails fails at the same
 * sequence, with the same table insertions behind it. */
struct Seq3 {
	u32 tok, src, lit, mc, off; /* position of the token in dst, of the literals in the chunk; literal run, match code, offset */
};
/* length bytes of a run code (v >= 0): n255 bytes of 255 and the rest, written by the lane itself (the loop runs for runs of
 * 270 and more: rare) */
static __device__ __forceinline__ u32 seq3_len_ext(u8 *o, u32 v)
{
	const u32 n255 = v / 255;
	for (u32 i = 0; i < n255; i++)
		o[i] = 255;
	o[n255] = (u8)(v - n255 * 255);
	return n255 + 1;
}
static __device__ void seq3_flush(const Seq3 &q, u32 cnt, const u8 *chunk, u8 *dst, int lane)
{
	const bool act = (u32)lane < cnt;
	const u32 lit = act ? q.lit : 0, mc = q.mc;
	u8 *o = dst + q.tok;
	if (act) {
		o[0] = (u8)((lit >= 15 ? 15u : lit) << 4 | (mc >= 15 ? 15u : mc));
		o++;
		if (E_RARE(lit >= 15))
			o += seq3_len_ext(o, lit - 15);
	}
	/* literals: exact, in pieces that may overlap each other (first and last piece of 8 or of 4, single bytes below
	 * 4); runs above 64 bytes by the whole wave, one at a time */
	const u8 *s_ = chunk + q.src;
	if (act && lit <= 64) {
		if (lit >= 8) {
			for (u32 i = 8; i + 8 < lit; i += 8) {
				const u64 v = ld64u(s_ + i);
				__builtin_memcpy(o + i, &v, 8);
			}
			const u64 a = ld64u(s_), b = ld64u(s_ + lit - 8);
			__builtin_memcpy(o, &a, 8);
			__builtin_memcpy(o + lit - 8, &b, 8);
		} else if (lit >= 4) {
			const u32 a = ld32u(s_), b = ld32u(s_ + lit - 4);
			st32u(o, a);
			st32u(o + lit - 4, b);
		} else {
			for (u32 i = 0; i < lit; i++)
				o[i] = s_[i];
		}
	}
	u64 big = wv_ballot(act && lit > 64);
	while (E_RARE(big != 0)) {
		const int j = wv_ffs(big) - 1;
		big &= big - 1;
		const u32 oj = wv_readlane((u32)(o - dst), j), sj = wv_readlane(q.src, j), lj = wv_readlane(lit, j);
		wave_copy(dst + oj, chunk + sj, lj, lane);
	}
	if (act) {
		o += lit;
		st16u(o, q.off);
		o += 2;
		if (E_RARE(mc >= 15))
			(void)seq3_len_ext(o, mc - 15);
	}
}

/* 16 bytes at any address as two 8-byte halves: one global_load_dwordx4 */
#define ENC3_LD16(P, LO, HI)                                                                        \
	do {                                                                                        \
		struct { u64 a, b; } q_;                                                            \
		__builtin_memcpy(&q_, (P), 16);                                                     \
		(LO) = q_.a;                                                                        \
		(HI) = q_.b;                                                                        \
	} while (0)

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
					 * [cand, cand + 16) -- readable: cand + 16 <= iend + 3, the input carries 8 bytes of slack -- and uses
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

template <int TM, bool PROF>
static __device__ __forceinline__ void enc3_body(u32 *tlo, u32 *thi, u32 *bitmap, u8 *ring, u8 *mwin,
						  const u8 *__restrict__ in, u64 n, u32 chunk, u32 rec0, u32 nrec,
						  u8 *__restrict__ slots, u64 slot_stride,
						  u32 *__restrict__ rec_len, const u32 *__restrict__ chk,
						  unsigned long long *prof)
{
	const u32 rec = rec0 + blockIdx.x;
	const int lane = wv_lane();
	if (rec >= nrec)
		return;
	const u64 start = (u64)rec * chunk;
	const u32 len = (u32)((n - start) < (u64)chunk ? (n - start) : (u64)chunk);
	/* a record of <= 64 KiB is a single independent block with the byU16 table: it belongs to the
	 * T_U16 kernel (launched for the ragged last record), everything longer to the linked-block
	 * kernels */
	if ((TM == T_U16) != (len <= ZMT_BLOCK))
		return;
	const u8 *src = in + start;
	u8 *dst = slots + (u64)rec * slot_stride;
	const u32 hdr = len ? 15 : 7;
	u32 op = 12 + hdr;

	if (lane == 0) {
		u8 d[10];
		st32u(dst, ZMT_SKIP_MAGIC);
		st32u(dst + 4, 4);
		st32u(dst + 12, ZMT_LZ4F_MAGIC);
		d[0] = (u8)(0x40 | (TM == T_U16 ? 0x20 : 0) | (len ? 0x08 : 0) | 0x04);
		d[1] = 0x40;
		for (int i = 0; i < 8; i++)
			d[2 + i] = (i < 4) ? (u8)(len >> (8 * i)) : 0;
		for (u32 i = 0; i < hdr - 5; i++)
			dst[16 + i] = d[i];
		dst[12 + hdr - 1] = (u8)(xxh32_short(d, hdr - 5) >> 8);
	}
	const u32 tab_words = (TM == T_P17) ? 2048 : 4096;
	for (u32 i = (u32)lane; i < tab_words; i += 64)
		tlo[i] = 0;
	for (u32 i = (u32)lane; i < 128; i += 64) {
		if (i < BM_BITS / 32)
			bitmap[i] = 0;
		if (TM == T_P17)
			thi[i] = 0;
	}
	wv_sync();

	InRing R;
	R.ring = ring;
	R.mwin = mwin;
	R.mbase = 0;
	R.chunk = src;
	R.rhi = 0;
	R.rlo = 0;
	for (int i = 0; i < 8; i++)
		R.pc[i] = 0;
#ifndef ZMT_EMU
	R.tq = PROF ? (u64)clock64() : 0;
	const u64 t_begin = R.tq;
#else
	R.tq = 0;
#endif
	/* the input buffer carries >= 8 readable bytes after its end (hash reads); never go further */
	R.limit = (u32)((n - start) < (u64)chunk + 8 ? (n - start) + 8 : (u64)chunk + 8);

	for (u32 pos = 0; pos < len; pos += ZMT_BLOCK) {
		u32 blen = len - pos < ZMT_BLOCK ? len - pos : ZMT_BLOCK;
		u32 c = encode_block3<TM, PROF>(tlo, thi, bitmap, R, pos, blen, dst + op + 4, blen - 1, lane);
		u32 bh = c;
		if (c == 0) {
			wv_sync(); /* every lane's stores of the attempt lie behind it before the same bytes are rewritten */
			wave_copy(dst + op + 4, src + pos, blen, lane);
			c = blen;
			bh = blen | 0x80000000u;
		}
		if (lane == 0)
			st32u(dst + op, bh);
		op += 4 + c;
	}
	if (lane == 0) {
		st32u(dst + op, 0);
		st32u(dst + op + 4, chk[rec]);
		st32u(dst + 8, op + 8 - 12);
		rec_len[rec] = op + 8;
	}
#ifndef ZMT_EMU
	if (PROF && lane == 0) {
		for (int i = 0; i < 8; i++)
			atomicAdd(prof + i, (unsigned long long)R.pc[i]);
		atomicAdd(prof + 8, (unsigned long long)((u64)clock64() - t_begin));
		atomicAdd(prof + 9, 1ull);
	}
#endif
}

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
		enc3_body<TM, PROFILE>(tlo, thi, bitmap, ring, mwin, in, n, chunk, rec0, nrec, slots, \
			      slot_stride, rec_len, chk, prof);                                                     \
	}

ENC3_KERNEL(zmt_lz4_enc3_u16_kernel, T_U16, 16384, false)
ENC3_KERNEL(zmt_lz4_enc3_p17_kernel, T_P17, 8192, false)
ENC3_KERNEL(zmt_lz4_enc3_u32_kernel, T_U32, 16384, false)
#ifndef ZMT_EMU
/* developer: the same kernel with the phase counters (gpumt_set_variant("profile", 9), tools/enc_prof.py) */
ENC3_KERNEL(zmt_lz4_enc3_p17_prof_kernel, T_P17, 8192, true)
#endif
