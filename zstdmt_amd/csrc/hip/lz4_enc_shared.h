/*
 * lz4_enc_shared.h -- what the LZ4 fast encoders lz4_enc3.hip (probe batches) and lz4_enc5.hip (64-position windows) share:
 * constants of the block format, the hash and the three table layouts, the 1 KiB input ring with its match-side window,
 * the sequence collector that writes 64 sequences at a time, and the record around the blocks.  Reference call site:
 * LZ4F_compressFrame at /root/reference/lib/lz4-mt_compress.c:281 (SURVEY.md Appendix B restates the block encoder).
 */
#ifndef ZMT_LZ4_ENC_SHARED_H
#define ZMT_LZ4_ENC_SHARED_H
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

#ifndef IPIECE
#define IPIECE 512u /* refill granule: 8 bytes per lane (lz4_enc5.hip: 256 = the lower half of the wave) */
#endif
/* bytes ring_want() makes resident ahead of a position: >= 72 (a batch of 64 consecutive probes reads
 * 8 bytes each) and small enough that IRING - IAHEAD - IPIECE >= 64 bytes of history stay behind it
 * (catch-up compares 64 bytes backwards, the re-match step reads ip - 2) */
#ifndef IAHEAD
#define IAHEAD (IRING >= 2048u ? 512u : 256u)
#endif
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
		if (IPIECE == 512u || 8u * (u32)lane < IPIECE) { /* (a piece of 256 bytes: 32 lanes) */
			*(u64 *)d = a;
			if ((p & (IRING - 1)) < IMIRROR) /* mirror: multi-byte reads never wrap */
				*(u64 *)(d + IRING) = a;
		}
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

/* The same flush in two halves, for a caller that has other loads in flight (lz4_enc5.hip's window: its candidates): the first
 * 16 literal bytes of every collected sequence are ASKED FOR with those loads (seq3_flush_ask), and once they have all landed
 * seq3_flush_put writes every sequence out of its two registers -- a run of up to 15 literals as its 8 / 4 / 2 / 1-byte pieces,
 * exact (the byte behind a run is another lane's token) -- with no load -> store round trip of its own: gfx9 counts loads and
 * stores in one in-order counter, so seq3_flush's three length classes are three serial round trips on the chunk's chain, and its
 * stores sit in front of the next window's candidate loads.  Longer runs (rare) take seq3_flush's code. */
/* "these loaded values are needed HERE": an empty statement that reads them, so that the compiler's wait for the loads stands at
 * this point of the program and not at their first use further down -- behind stores whose completion the wait would then
 * include (one in-order counter for loads and stores), or behind a join where it can no longer tell what is outstanding */
template <bool FENCE> static __device__ __forceinline__ void vm_landed(u64 &a, u64 &b)
{
#ifndef ZMT_EMU
	u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
	if (FENCE) /* (no memory operation of the program moves across: the stores behind it stay behind it) */
		asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : : "memory");
	else
		asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
	a = (u64)a0 | (u64)a1 << 32;
	b = (u64)b0 | (u64)b1 << 32;
#else
	(void)a;
	(void)b;
#endif
}
static __device__ __forceinline__ void seq3_flush_ask(const Seq3 &q, u32 cnt, const u8 *chunk, int lane, u64 &f0, u64 &f1)
{
	/* (up to 16 bytes behind an anchor: inside the chunk or the 64 bytes of slack behind the input, include/gpumt.h) */
	const u8 *s_ = chunk + ((u32)lane < cnt ? q.src : 0u);
	ENC3_LD16(s_, f0, f1);
}
static __device__ __forceinline__ void seq3_flush_put(const Seq3 &q, u32 cnt, const u8 *chunk, u8 *dst, int lane, u64 f0, u64 f1)
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
	/* (both registers are looked at before the first store leaves, by every lane: the one wait for the loads stands in front of
	 * straight-line code, and no later join of the exec-mask regions below makes the compiler wait again -- with the stores) */
	const u64 r8 = (lit & 8u) ? f1 : f0;
	const u32 r4 = (lit & 4u) ? (u32)(r8 >> 32) : (u32)r8;
	const u32 r2 = (lit & 2u) ? r4 >> 16 : r4;
	const u32 lit15 = lit <= 15u ? lit : 0u; /* (lit = 0 for lanes without a sequence: no piece) */
	{
		u8 *d = o;
		if (lit15 & 8u)
			__builtin_memcpy(d, &f0, 8);
		d += lit15 & 8u;
		if (lit15 & 4u)
			st32u(d, (u32)r8);
		d += lit15 & 4u;
		if (lit15 & 2u)
			st16u(d, r4 & 0xFFFFu);
		d += lit15 & 2u;
		if (lit15 & 1u)
			d[0] = (u8)r2;
	}
	if (E_RARE(wv_any(lit > 15u))) {
		/* longer runs (rare): 16 .. 64 bytes by the lane in 8-byte pieces, above that by the whole wave, one at a time */
		if (lit > 15u && lit <= 64u) {
			const u8 *s_ = chunk + q.src;
			for (u32 i = 8; i + 8 < lit; i += 8) {
				const u64 v = ld64u(s_ + i);
				__builtin_memcpy(o + i, &v, 8);
			}
			const u64 b = ld64u(s_ + lit - 8);
			__builtin_memcpy(o, &f0, 8);
			__builtin_memcpy(o + lit - 8, &b, 8);
		}
		u64 big = wv_ballot(act && lit > 64);
		while (big != 0) {
			const int j = wv_ffs(big) - 1;
			big &= big - 1;
			const u32 oj = wv_readlane((u32)(o - dst), j), sj = wv_readlane(q.src, j), lj = wv_readlane(lit, j);
			wave_copy(dst + oj, chunk + sj, lj, lane);
		}
	}
	if (act) {
		o += lit;
		st16u(o, q.off);
		o += 2;
		if (E_RARE(mc >= 15))
			(void)seq3_len_ext(o, mc - 15);
	}
}


/* ---- the record around the blocks: skippable header, LZ4F frame header, block headers (stored when a block does not fit),
 * end mark and content checksum; ENC::block<TM, PROF>() is the block encoder (lz4_enc3.hip, lz4_enc5.hip) ---- */
template <int TM, bool PROF, class ENC>
static __device__ __forceinline__ void enc_frame_body(u32 *tlo, u32 *thi, u32 *bitmap, u8 *ring, u8 *mwin,
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
	/* the ring never loads beyond 8 bytes behind the input (hash reads); the candidate loads of the encoders may reach up to
	 * 24 bytes behind it, inside the 64 readable bytes include/gpumt.h requires after d_in + n */
	R.limit = (u32)((n - start) < (u64)chunk + 8 ? (n - start) + 8 : (u64)chunk + 8);

	for (u32 pos = 0; pos < len; pos += ZMT_BLOCK) {
		u32 blen = len - pos < ZMT_BLOCK ? len - pos : ZMT_BLOCK;
		u32 c = ENC::template block<TM, PROF>(tlo, thi, bitmap, R, pos, blen, dst + op + 4, blen - 1, lane);
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


#endif
