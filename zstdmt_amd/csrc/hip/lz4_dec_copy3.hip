/*
 * lz4_dec_copy3.hip -- copy stage of the LZ4 frame decoder, round 3 ("copy3").
 *
 * Wave per record, the token and batch lists of the parse kernel in, the content out (replaces LZ4F_decompress at
 * /root/reference/lib/lz4-mt_decompress.c:349-362 together with the frames / parse4 kernels).
 *
 * What the round-3 counters said about its predecessor, round 2's copy2 (profiles/r03_tcc_requests.json): it fetched 45.0 GB per
 * 8 GiB in 351.6 M requests, every one of them a full 128-byte line from memory -- one per match whose
 * source lay before its 2-4 KiB LDS window (37 % of the matches at a 4 KiB window) -- i.e. 5.2 TB/s for
 * 13.4 GB of useful bytes: bandwidth-bound on far-match lines as much as on instruction issue.
 *
 * So this kernel is built around (1) a big power-of-two LDS RING of the record's output (8 or 16 KiB per
 * wave: 19 % / 6 % of the matches are sourced before it), addressed modulo, never slid; one wave per
 * workgroup so that the LDS of a CU divides into as many waves as fit; (2) batches cut by one rule that
 * makes everything inside them simple (<= 64 "small" sequences, inside one lap of the ring, inside the
 * 1 KiB stage: two ballots on fields the lanes compute anyway), so the batch body has no per-sequence
 * special cases and the ring is never slid; (3) the compressed bytes of
 * a batch staged by ONE LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
 * VGPRs), a batch ahead, into one of two stage buffers; (4) everything unusual -- long runs, the block's
 * last sequence, lap-crossing sequences -- executed generically byte by byte THROUGH the ring, so the
 * ring history is never lost; (5) one vector-memory schedule per batch: gfx9 has ONE in-order counter
 * for loads and stores, so a wait for a load is a wait for every store before it.  The batch's loads of
 * sources before the ring go out right after the scan, are waited for after the literal copies, and only
 * then does everything else leave -- the ring-to-memory stores of what the PREVIOUS batches produced,
 * the next token positions, the next stage -- so the one full drain per batch (at its top) finds
 * operations that are most of a batch old.
 *
 * Measured on MI355X (tools/ubench/store_cost.hip): a misaligned ds_write costs one LDS cycle per ACTIVE
 * lane whatever its width (b64: 64 / 32 / 16 / 8.3 cycles at 64 / 32 / 16 / 8 lanes), an aligned one 7-13.
 * Hence: literal stores are issued only by lanes that have literals, the second match piece only by
 * matches longer than 8 bytes.
 */
#include "lz4_common.h"
#include "lz4_frame.h"

/* ONE stage buffer (round 6): the next batch's bytes are requested once this batch's literals are copied -- nothing reads the
 * stage after that, the sources fetched from before the ring stay in registers until they are stored -- so a second buffer
 * bought no overlap, and without it the 4 KiB variant needs 4 096 + 16 + 1 008 = 5 120 bytes of LDS: 32 waves per CU instead of
 * 26.  The stage follows its resident waves (profiles/r06_sweeps/copy3_wavecap.txt: 24 waves +10 %, 20 +20 %, 16 +31 %) */
#define C3_CSTAGE 976u
#define C3_CSLACK 32u
#define C3_CBUF (C3_CSTAGE + C3_CSLACK)
#define C3_BLK_STORED 0x80000000u
#define C3_XOUT 2048u /* most output bytes of one batch: two flush pieces of 1 KiB */
#define C3_NEEDS_SERIAL 100u
#define E_C3_RARE(c) __builtin_expect(!!(c), 0)

#ifndef ZMT_EMU
#define C3KT() (PROF ? (u64)clock64() : 0ull)
#else
#define C3KT() 0ull
#endif
#define C3PC(i)                                                                                    \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = C3KT();                                                     \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

static __device__ __forceinline__ void c3_st64(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }
/* four bytes at any LDS address as four byte stores.  tools/ubench/lds_cost.hip: an LDS access that is not dword
 * aligned costs the pipe one cycle per ACTIVE lane (64 for a full wave), a byte store 4.6 cycles per wave
 * instruction at any address -- so where most lanes store, four byte stores (18 cycles) beat one ds_write_b32 */
static __device__ __forceinline__ void c3_st32b(u8 *p, u32 v)
{
	/* (volatile: the compiler would merge them back into one unaligned ds_write_b32; the LDS address space is
	 * spelled out because a volatile access through a generic pointer becomes a flat_store) */
#ifdef ZMT_EMU
	volatile u8 *const q = p;
#else
	volatile __attribute__((address_space(3))) u8 *const q = (volatile __attribute__((address_space(3))) u8 *)p;
#endif
	q[0] = (u8)v;
	q[1] = (u8)(v >> 8);
	q[2] = (u8)(v >> 16);
	q[3] = (u8)(v >> 24);
}
static __device__ __forceinline__ u64 c3_tok_base(u64 coff, u32 gb) { return ((coff / 3) & ~63ull) + 128ull * gb; }
/* (the 8-byte loads of sources before the ring fetch a whole 128-byte line each; nt / sc0 / sc1 policy bits on
 * the load do not change the request size -- profiles/r03_sweeps/far_load_cache_policy.txt -- so they are plain) */

/* 1 KiB of the stream into LDS, lane l moving bytes [16 l, 16 l + 16) of it; `g` is 16-byte aligned */
static __device__ __forceinline__ void c3_stage(u8 *cb, const u8 *g, u32 nbytes, int lane)
{
	if (16u * (u32)lane < (nbytes < C3_CBUF ? nbytes : C3_CBUF)) {
#ifdef ZMT_EMU
		__builtin_memcpy(cb + 16 * lane, g + 16 * lane, 16);
#else
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 16 * lane),
						 (__attribute__((address_space(3))) void *)cb, 16, 0, 0);
#endif
	}
}
static __device__ __forceinline__ void c3_wait_vm()
{
#ifndef ZMT_EMU
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	wv_sync();
}

/* eight staged bytes at offset o (any alignment): three aligned dword reads + two funnel shifts */
static __device__ __forceinline__ u64 c3_ld64s(const u8 *base, u32 o)
{
	const u32 *w = (const u32 *)(base + (o & ~3u));
	const u32 a0 = w[0], a1 = w[1], a2 = w[2];
	return (u64)wv_alignbyte(a1, a0, o) | ((u64)wv_alignbyte(a2, a1, o) << 32);
}

template <u32 WIN, bool PROF = false> struct C3 {
	static constexpr u32 MASK = WIN - 1u;

	/* eight bytes at offset o (any alignment) of an LDS region addressed modulo m + 1 from `base` (the ring: m =
	 * MASK; a stage buffer: m = ~0): the three dwords wrap individually */
	static __device__ __forceinline__ u64 ld64m(const u8 *base, u32 o, u32 m)
	{
		const u32 a = o & ~3u;
		const u32 a0 = *(const u32 *)(base + (a & m));
		const u32 a1 = *(const u32 *)(base + ((a + 4u) & m));
		const u32 a2 = *(const u32 *)(base + ((a + 8u) & m));
		return (u64)wv_alignbyte(a1, a0, o) | ((u64)wv_alignbyte(a2, a1, o) << 32);
	}

	struct St {
		u32 opos, flushed, valid_from, fenced;
#ifdef C3_XXH_PROXY
		u32 xacc; /* developer A/B (profiles/r06_sweeps/copy3_verify_from_lds.txt): the XXH32 accumulator of lanes 0..3 */
#endif
	};

	/* ring -> memory for output positions [st.flushed, upto); the body in aligned 16-byte pieces */
	static __device__ __forceinline__ void flush_to(St &st, const u8 *ring, u8 *out, u32 upto, int lane)
	{
		u32 f = st.flushed;
		if (upto <= f)
			return;
		if (f & 15u) {
			u32 head = 16u - (f & 15u);
			if (head > upto - f)
				head = upto - f;
			if ((u32)lane < head)
				out[f + lane] = ring[(f + lane) & MASK];
			f += head;
		}
		const u32 body_end = f + ((upto - f) & ~15u);
		for (u32 pos = f + 16u * (u32)lane; pos < body_end; pos += 1024u) {
			const u8 *r = ring + (pos & MASK);
			const u64 a = *(const u64 *)r, b = *(const u64 *)(r + 8);
			c3_st64(out + pos, a);
			c3_st64(out + pos + 8, b);
		}
		if (body_end != upto && (u32)lane < upto - body_end)
			out[body_end + lane] = ring[(body_end + lane) & MASK];
		st.flushed = upto;
	}

	/* the same for the batch loop, where both ends are multiples of 16 unless a stored block or the record's start
	 * left `flushed` odd (then the general routine realigns it) */
	static __device__ __forceinline__ void flush_aligned(St &st, const u8 *ring, u8 *out, u32 upto, int lane)
	{
		if (st.flushed & 15u) {
			flush_to(st, ring, out, upto, lane);
			return;
		}
		for (u32 pos = st.flushed + 16u * (u32)lane; pos < upto; pos += 1024u) {
			const u8 *r = ring + (pos & MASK);
			const u64 a = *(const u64 *)r, b = *(const u64 *)(r + 8);
			c3_st64(out + pos, a);
			c3_st64(out + pos + 8, b);
		}
#ifdef C3_XXH_PROXY
		/* the content checksum fed from the ring during the flush, lower bound of its cost: the four accumulator chains of
		 * XXH32 (lane j: dword j of every 16-byte stripe) and nothing else -- no tail, no finalisation, no compare */
		if (lane < 4) {
			u32 acc = st.xacc;
			for (u32 pos = st.flushed; pos < upto; pos += 16u) {
				const u32 w = *(const u32 *)(ring + ((pos + 4u * (u32)lane) & MASK));
				acc += w * 2246822519u;
				acc = (acc << 13 | acc >> 19) * 2654435761u;
			}
			st.xacc = acc;
		}
#endif
		if (upto > st.flushed)
			st.flushed = upto;
	}

	/* one sequence of any shape, by the whole wave, byte by byte THROUGH the ring (fields are wave-uniform).
	 * Source bytes older than the ring come from the output in memory. */
	static __device__ __forceinline__ void generic(const u8 *lsrc, u32 lit, u32 off, u32 ml, u8 *ring, u8 *out, St &st, int lane)
	{
		while (lit) {
			const u32 c = lit < 512u ? lit : 512u;
			for (u32 i = (u32)lane; i < c; i += 64)
				ring[(st.opos + i) & MASK] = lsrc[i];
			wv_sync();
			st.opos += c;
			lsrc += c;
			lit -= c;
			flush_to(st, ring, out, st.opos & ~15u, lane);
		}
		u32 done = 0;
		while (done < ml) {
			/* a chunk never reads what it writes: at most `off` bytes unless the period is short, in which
			 * case every byte comes from the `off` bytes in front of the match */
			u32 c = ml - done < 512u ? ml - done : 512u;
			if (off >= 64u && c > off)
				c = off;
			const u32 mpos = st.opos;
			const u32 hist = mpos + c + 16u > WIN ? mpos + c + 16u - WIN : 0u; /* ring holds [hist, mpos) */
			const u32 lo = hist > st.valid_from ? hist : st.valid_from;
			const u32 first_src = off < 64u ? mpos - done - off : mpos - off;
			if (first_src < lo && st.fenced < st.flushed) {
				wave_mem_fence(); /* bytes this wave stored are about to be loaded back */
				st.fenced = st.flushed;
			}
			for (u32 i = (u32)lane; i < c; i += 64) {
				const u32 p = off < 64u ? mpos - done - off + (done + i) % off : mpos - off + i;
				const u8 b = p >= lo ? ring[p & MASK] : out[p];
				ring[(mpos + i) & MASK] = b;
			}
			wv_sync();
			st.opos += c;
			done += c;
			flush_to(st, ring, out, st.opos & ~15u, lane);
		}
	}

	/* overlapping match (offset < length) of one lane inside the ring: strictly forward */
	static __device__ __forceinline__ void match_ovl(u8 *ring, u32 mpos, u32 off, u32 ml)
	{
		u32 j = 0;
		for (u32 i = 0; i < ml; i++) {
			ring[(mpos + i) & MASK] = ring[(mpos - off + j) & MASK];
			if (++j == off)
				j = 0;
		}
	}

	/* match of 4..64 bytes at output position mpos whose source -- offset so of the region (sb, sm), see ld64m -- is
	 * complete and does not overlap it.  Stores are 4-byte pieces placed so that ONE per-lane condition covers all
	 * lengths up to 16 (every divergent `if` costs this kernel four scalar instructions, and the scalar pipe is what
	 * it saturates): bytes 0-3 and the last 4 always, bytes 4-7 and the 4 before the last 4 when the match has 8 or
	 * more; the middle of the rare long one in 8-byte steps.  match_st: the stores, given the first 8 bytes `a` and the 8
	 * bytes `b` at offset tl = ml - 8 (ml - 4 for a match shorter than 8) */
	template <bool BYTES> static __device__ __forceinline__ void match_st(u8 *ring, u32 mpos, u32 ml, u64 a, u64 b)
	{
		u8 *const d = ring + (mpos & MASK); /* a batch lies inside one lap: no wrap on the destination side */
		const bool wide = ml >= 8u;
		const u32 tl = wide ? ml - 8u : ml - 4u;
		if (BYTES) { /* the pass most lanes take */
			c3_st32b(d, (u32)a);
			c3_st32b(d + ml - 4u, wide ? (u32)(b >> 32) : (u32)b);
			if (wide) {
				c3_st32b(d + 4, (u32)(a >> 32));
				c3_st32b(d + tl, (u32)b);
			}
		} else {
			st32u(d, (u32)a);
			st32u(d + ml - 4u, wide ? (u32)(b >> 32) : (u32)b);
			if (wide) {
				st32u(d + 4, (u32)(a >> 32));
				st32u(d + tl, (u32)b);
			}
		}
	}
	template <bool BYTES> static __device__ __forceinline__ void match(u8 *ring, u32 mpos, u32 ml, const u8 *sb, u32 so, u32 sm)
	{
		u8 *const d = ring + (mpos & MASK);
		const u32 tl = ml >= 8u ? ml - 8u : ml - 4u;
		const u64 a = ld64m(sb, so, sm), b = ld64m(sb, so + tl, sm);
		if (ml > 16u) {
			for (u32 i = 8; i + 8 < ml; i += 8)
				c3_st64(d + i, ld64m(sb, so + i, sm));
		}
		match_st<BYTES>(ring, mpos, ml, a, b);
	}

	static __device__ __forceinline__ void
	body(const u8 *__restrict__ stream, u64 stream_bytes, u32 rec0, u32 nrec, u8 *out_base,
	     const u64 *__restrict__ out_off, const u32 *__restrict__ out_len, const u64 *__restrict__ blk0,
	     const u64 *__restrict__ blk_coff, const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
	     const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok, const u32 *__restrict__ blk_ntok,
	     const u32 *__restrict__ blk_olen,
	     u32 *__restrict__ status, u8 *ring, u8 *cbuf, unsigned long long *prof)
	{
		const int lane = wv_lane();
		/* per-phase cycle counters of the profiling build (developer tool, tools/dec_prof.py) */
		u64 pc[PROF ? 14 : 1] = {0}, tq = C3KT();
		const u64 t_begin = tq;
		const u32 rec = rec0 + blockIdx.x;
		if (rec >= nrec)
			return;
		if (wv_readfirst(status[rec]) != ST_OK)
			return;
		u8 *const out = out_base + out_off[rec];
		const u32 cap = out_len[rec];
		const u64 b0 = blk0[rec];
		const u32 nb = wv_readfirst(rec_nblk[rec]);
		const bool indep = wv_readfirst(rec_flags[rec]) & 1;
		u32 stc = ST_OK;
		St st;
		st.opos = st.flushed = st.valid_from = st.fenced = 0;
#ifdef C3_XXH_PROXY
		st.xacc = 0x24234428u + (u32)lane;
#endif

		for (u32 bi = 0; bi < nb && stc == ST_OK; bi++) {
			const u32 gb = (u32)(b0 + bi);
			const u32 cs = wv_readfirst(blk_csize[gb]);
			const u64 coff = blk_coff[gb];
			const u8 *const src = stream + coff;
			const u32 olen = wv_readfirst(blk_olen[gb]);
			const u32 bstart = st.opos;
			if (olen == 0xFFFFFFFFu || cap - bstart < olen) {
				stc = ST_BAD_BLOCK;
				break;
			}
			if (cs & C3_BLK_STORED) {
				const u32 bsz = cs & 0x7FFFFFFFu;
				flush_to(st, ring, out, st.opos, lane);
				wave_copy(out + st.opos, src, bsz, lane);
				st.opos += bsz;
				st.flushed = st.opos;
				st.valid_from = st.opos; /* the ring does not hold a stored block: sources in it come from memory */
				continue;
			}
			const u32 ntok = wv_readfirst(blk_ntok[gb]);
			const u64 tbase = c3_tok_base(coff, gb);
			const u16 *const tk = tok + tbase;
			const u32 low = indep ? bstart : 0;
			/* ---- the block's sequences in batches this kernel cuts itself ----
			 * A batch = the longest run of "small" sequences (literal run <= 64, match <= 64, no 255 length bytes, not
			 * the block's last) from sequence t0 on -- at most 64 -- whose compressed bytes lie inside the 1 KiB stage,
			 * whose output is at most C3_XOUT bytes and does not cross a multiple of the ring size ("lap": stores inside a
			 * batch never wrap).  The sequence that ends the run, if it is not small itself, is executed generically
			 * right behind the batch.  Token positions of the next batch are loaded as soon as the cut is known, its
			 * compressed bytes staged by one LDS-DMA into the other buffer. */
#define C3_TOK(T0) (((T0) + (u32)lane < ntok) ? (u32)tk[(T0) + (u32)lane] : 0u)
#define C3_TOK2(T0) (((T0) + 64u + ((u32)lane & 1u) < ntok) ? (u32)tk[(T0) + 64u + ((u32)lane & 1u)] : 0u)
#define C3_STAGE(C0, BUF)                                                                                          \
	do {                                                                                                       \
		const u8 *g_ = src + (C0);                                                                         \
		const u32 a_ = (u32)((size_t)g_ & 15u);                                                            \
		c3_stage(cbuf, g_ - a_, cs - (C0) + a_ + 16u, lane);                                               \
	} while (0)
			u32 t0 = 0;
			u32 q_cur = C3_TOK(0), q2_cur = C3_TOK2(0);
			if (ntok)
				C3_STAGE(wv_readlane(q_cur, 0), 0);
			while (t0 < ntok && stc == ST_OK) {
				C3PC(1);
				c3_wait_vm(); /* the stage and the token positions of this batch */
				C3PC(0);
				const u32 q = q_cur, q2 = q2_cur;
				u8 *const cb = cbuf;
				if (PROF)
					pc[PROF ? 12 : 0]++;
				/* ---------- fields of up to 64 sequences, lane = sequence ----------
				 * (written for the scalar pipe, which this kernel saturates: per-lane conditions are folded into the values
				 * that are compared -- a lane that must not pass gets 0xFFFF -- instead of mask algebra on SGPR pairs) */
				const u32 rem = ntok - t0;
				const u32 c0 = wv_readlane(q, 0);
				const u32 al = (u32)((size_t)(src + c0) & 15u);
				const u32 qr = q - c0 + al; /* this lane's token in the stage */
				u32 lit, ml, off, lsrc;
				u64 sm0, hard_m;
				{
					const bool v1 = ((u32)lane < rem ? qr : 0xFFFFu) <= C3_CSTAGE - 12u; /* the token's first bytes are staged */
					const u64 w = c3_ld64s(cb, v1 ? qr : 0u);
					const u32 wl = (u32)w;
					const u32 tokb = wl & 255u;
					const bool lx = (tokb >> 4) == 15u, mx = (tokb & 15u) == 15u;
					lit = (tokb >> 4) + (lx ? (wl >> 8) & 255u : 0u);
					lsrc = qr + 1u + (lx ? 1u : 0u);
					const u32 mo = lsrc + lit; /* where the offset sits */
					const bool v2 = (v1 ? mo : 0xFFFFu) <= C3_CSTAGE - 4u; /* ... and its last ones */
					const u32 w2 = (u32)c3_ld64s(cb, v2 ? mo : 0u);
					off = w2 & 0xFFFFu;
					ml = (tokb & 15u) + 4u + (mx ? (w2 >> 16) & 255u : 0u);
					/* a 255 length byte makes the run 270 / 274: "above 64" covers it; the block's last sequence counts as big */
					const u32 big = (u32)lane + 1u == rem ? 0xFFFFu : (lit > (v2 ? ml : 0u) ? lit : (v2 ? ml : 0u));
					sm0 = wv_ballot(wv_opaque(v2 ? big : 0xFFFFu) <= 64u);  /* small: fully staged, no run above 64, not the last */
					hard_m = wv_ballot(wv_opaque(v1 ? big : 0u) > 64u);      /* needs the generic path whatever batch it would be in */
				}
				C3PC(2);
				/* ---------- cut: the run of small sequences, then output positions, then span and lap ---------- */
				const u32 n0 = ~sm0 ? (u32)wv_ffs(~sm0) - 1u : 64u;
				const bool in_run = (u32)lane < n0;
				const u32 len = in_run ? lit + ml : 0u;
				const u32 incl = wv_scan_incl(len);
				const u32 o0 = st.opos;
				const u32 op = o0 + incl - len;
				const u32 lap_end = (o0 | MASK) + 1u;
				const u32 olim = o0 + C3_XOUT < lap_end ? o0 + C3_XOUT : lap_end;
				const u64 fit = wv_ballot(wv_opaque(in_run ? op + len : 0xFFFFFFFFu) <= olim);
				const u32 n = ~fit ? (u32)wv_ffs(~fit) - 1u : 64u; /* (the conditions are monotone in the lane) */
				/* the sequence behind the run is executed generically if it is "hard", or if it is small but crosses the
				 * lap boundary on its own (it could not open a batch either) */
				bool single = false, single_hard = false;
				if (n < 64u) {
					single_hard = (hard_m >> n) & 1u;
					const u32 op_n = wv_readlane(op, (int)n), len_n = wv_readlane(len, (int)n);
					single = single_hard | (((sm0 >> n) & 1u) && op_n < lap_end && op_n + len_n > lap_end);
				}
				const u32 lit_r = lit, ml_r = ml, off_r = off; /* (lane n's are the fields of a lap-crosser behind the run) */
				const bool act = (u32)lane < n;
				if (!act) {
					lit = 0;
					ml = 0;
					off = 1;
				}
				const u32 r = n + (single ? 1u : 0u);
				const u32 t0n = t0 + r;
				if (r == 0) {
					stc = ST_BAD_BLOCK; /* cannot happen: a small sequence that does not cross the lap fits a batch alone */
					break;
				}
				/* what leaves for memory this batch (issued together, below): the ring up to the batch's start, the
				 * token positions and the stage of the next batch */
#define C3_ISSUE()                                                                                                 \
	do {                                                                                                       \
		flush_aligned(st, ring, out, st.opos & ~15u, lane);                                                \
		q_cur = C3_TOK(t0n);                                                                               \
		q2_cur = C3_TOK2(t0n);                                                                             \
		if (t0n < ntok)                                                                                    \
			C3_STAGE(r < 64u ? wv_readlane(q, (int)(r & 63u)) : wv_readlane(q2, (int)(r & 1u)), 0);      \
	} while (0)
				if (n == 0) {
					C3_ISSUE();
				} else {
					const u32 total = wv_readlane(incl, (int)(n - 1u));
					const u32 mpos = op + lit;
					const u32 src_pos = mpos - off;
					const u32 eff = ml < off ? ml : off;
					const u32 o_end = o0 + total;
					/* (offset 0 or beyond the output so far, one compare: 0 - 1 is above everything; a lane without a sequence has offset 1 and output in front of it) */
					if (wv_any(off - 1u >= mpos - low) | (total > cap - o0)) {
						stc = ST_BAD_BLOCK;
						break;
					}
					/* the ring holds [near_lo, o0): what this batch will overwrite (plus store slack) is gone */
					const u32 hist = o_end + 16u > WIN ? o_end + 16u - WIN : 0u;
					const u32 near_lo = hist > st.valid_from ? hist : st.valid_from;
					const bool is_far = act & (src_pos < near_lo);
					const bool anyfar = wv_any(is_far);
					/* a source that straddles the start of the ring or of what memory holds, or an overlapping match
					 * sourced before the ring: possible only right after a stored block or when memory lags far behind;
					 * then this batch goes one sequence at a time */
					if ((st.valid_from > hist || near_lo + 128u > st.flushed) && anyfar &&
					    wv_any(is_far && (src_pos + eff > near_lo || src_pos + eff > st.flushed || off < ml))) {
						for (u32 i = 0; i < n; i++) {
							const u32 l1 = wv_readlane(lit, (int)i), m1 = wv_readlane(ml, (int)i);
							generic(cb + wv_readlane(lsrc, (int)i), l1, wv_readlane(off, (int)i), m1, ring, out, st, lane);
						}
						C3_ISSUE();
					} else {
						C3PC(3);
						/* ---- sources before the ring: 16 bytes of the output in memory, loads first ---- */
						u64 f0 = 0, f1 = 0;
						if (anyfar) {
							if (wv_any(wv_opaque(is_far ? src_pos + eff : 0u) > st.fenced)) {
								wave_mem_fence();
								st.fenced = st.flushed;
							}
							if (is_far) {
								f0 = ld64u(out + src_pos);
								if (ml > 8u)
									f1 = ld64u(out + src_pos + 8);
							}
						}
						C3PC(4);
						/* ---- literals: only lanes that have some; 4-byte piece for 1..4, 8-byte pieces above ---- */
#ifdef C3X_NO_LIT
						if (false) {
#else
						if (act & (lit != 0)) {
#endif
							/* 4-byte pieces; a piece may spill <= 3 bytes into the lane's own match, written below */
							u8 *const dl = ring + (op & MASK);
							const u64 a = c3_ld64s(cb, lsrc);
							c3_st32b(dl, (u32)a);
							if (lit > 4u) {
								c3_st32b(dl + 4, (u32)(a >> 32));
								if (lit > 8u) {
									c3_st64(dl + lit - 8u, c3_ld64s(cb, lsrc + lit - 8u));
									for (u32 i = 8; i + 8 < lit; i += 8)
										c3_st64(dl + i, c3_ld64s(cb, lsrc + i));
								}
							}
						}
						wv_sync();
						C3PC(5);
						if (anyfar) {
							/* the fetched sources (16 bytes per lane) stay in registers until the first match round stores them */
							if (is_far) {
								/* the rare long one: the rest straight from memory into the ring (complete, unordered) */
								for (u32 i = 16; i < ml; i += 8) {
									const u32 o = i + 8 <= ml ? i : ml - 8;
									c3_st64(ring + ((mpos + o) & MASK), ld64u(out + src_pos + o));
								}
							}
							wv_sync();
						}
						C3PC(6);
						/* every load of this batch has landed: now the stores and the prefetches leave */
						C3_ISSUE();
						C3PC(9);
						/* ---- matches: complete sources first (one path: ring or fetched slot), then watermark rounds ---- */
						const bool ovl = off < ml;
						bool fin = !(act & (ml != 0));
						{
							const bool r1 = (!fin) & (is_far | (src_pos + eff <= o0)) & (!ovl);
#ifdef C3X_NO_R1
							if (r1)
								fin = true;
							if (false) {
#else
							if (r1) {
#endif
								/* one store sequence for both kinds; a source in the ring is read here (three aligned dwords
								 * per 8 bytes), a fetched one is shifted out of its two registers */
								const u32 mlc = (is_far && ml > 16u) ? 16u : ml;
								const u32 tl = mlc >= 8u ? mlc - 8u : mlc - 4u;
								u64 a = f0, b = f1;
								if (is_far) {
									const u32 sh = 8u * (tl & 7u);
									b = tl >= 8u ? f1 : (f0 >> sh) | ((f1 << 1) << (63u - sh));
								} else {
									a = ld64m(ring, src_pos, MASK);
									b = ld64m(ring, src_pos + tl, MASK);
									if (ml > 16u) {
										u8 *const d = ring + (mpos & MASK);
										for (u32 i = 8; i + 8 < ml; i += 8)
											c3_st64(d + i, ld64m(ring, src_pos + i, MASK));
									}
								}
								match_st<true>(ring, mpos, mlc, a, b);
								fin = true;
							}
						}
						wv_sync();
						C3PC(7);
#ifdef C3X_NO_ROUNDS
						if (false) {
#else
						/* (the set of unfinished matches is kept as the wave's mask: one ballot per round, of the lanes that
						 * went, instead of the compiler's round trip through a register for every test of `fin`; overlapping
						 * matches -- 0.2 % -- are looked for once per batch, not once per round) */
						u64 unf = wv_ballot(!fin);
						if (unf != 0) {
#endif
							const bool any_ovl = (unf & wv_ballot(off < ml)) != 0;
							do {
								if (PROF)
									pc[PROF ? 13 : 0]++;
								const u32 W = wv_readlane(mpos, wv_ffs(unf) - 1);
								const bool go = (!fin) & (src_pos + eff <= W);
								if (go & (!ovl))
									match<false>(ring, mpos, ml, ring, src_pos, MASK);
								if (E_C3_RARE(any_ovl)) { /* (offset < length: 0.2 % of the matches) */
									if (go & ovl)
										match_ovl(ring, mpos, off, ml);
								}
								fin = fin | go;
								unf &= ~wv_ballot(src_pos + eff <= W); /* (= the lanes that went: the ballot of one compare) */
								wv_sync();
							} while (unf != 0);
						}
						C3PC(8);
						st.opos = o_end;
					}
				}
				if (single && !single_hard) {
					/* ---------- a small sequence across the lap boundary: byte by byte, wrapping ---------- */
					const u32 s_lit = wv_readlane(lit_r, (int)(n & 63u)), s_ml = wv_readlane(ml_r, (int)(n & 63u));
					const u32 s_off = wv_readlane(off_r, (int)(n & 63u));
					/* block position of its literals (the stage has been overwritten by the fetched sources) */
					const u32 s_lpos = wv_readlane(q + (lsrc - qr), (int)(n & 63u));
					if (s_off == 0 || s_off > st.opos + s_lit - low || s_lit + s_ml > cap - st.opos) {
						stc = ST_BAD_BLOCK;
						break;
					}
					generic(src + s_lpos, s_lit, s_off, s_ml, ring, out, st, lane);
					C3PC(10);
				} else if (single) {
					/* ---------- one sequence of any shape (fields from memory, wave-uniform) ---------- */
					const u32 qq = n < 64u ? wv_readlane(q, (int)(n & 63u)) : wv_readlane(q2, 0);
					const bool is_last = t0 + n + 1u == ntok;
					u32 t = uld8(src + qq), l2 = t >> 4, h = qq + 1;
					if (l2 == 15) {
						u32 b;
						do {
							b = uld8(src + h++);
							l2 += b;
						} while (b == 255);
					}
					u32 goff = 1, m2 = 0;
					if (!is_last) {
						u32 m = h + l2;
						goff = uld16(src + m);
						m += 2;
						m2 = t & 15;
						if (m2 == 15) {
							u32 b;
							do {
								b = uld8(src + m++);
								m2 += b;
							} while (b == 255);
						}
						m2 += 4;
						if (goff == 0 || goff > st.opos + l2 - low) {
							stc = ST_BAD_BLOCK;
							break;
						}
					}
					if (l2 + m2 > cap - st.opos) {
						stc = ST_BAD_BLOCK;
						break;
					}
					generic(src + h, l2, goff, m2, ring, out, st, lane);
					C3PC(10);
				}
				t0 = t0n;
			}
#undef C3_TOK
#undef C3_TOK2
#undef C3_STAGE
#undef C3_ISSUE
			if (stc == ST_OK && (t0 != ntok || st.opos != bstart + olen))
				stc = ST_BAD_BLOCK;
		}
		if (stc == ST_OK || stc == ST_SIZE_MISMATCH)
			flush_to(st, ring, out, st.opos, lane);
		if (stc == ST_OK && st.opos != cap)
			stc = ST_SIZE_MISMATCH;
		if (lane == 0 && stc != ST_OK)
			status[rec] = stc;
#ifdef C3_XXH_PROXY
		if (lane < 4 && st.xacc == 0x5EED5EEDu) /* (keeps the chains alive) */
			status[rec] = ST_BAD_BLOCK;
#endif
#ifndef ZMT_EMU
		if (PROF && prof && lane == 0) {
			for (int i = 0; i < 11; i++)
				atomicAdd(prof + i, (unsigned long long)pc[PROF ? i : 0]);
			atomicAdd(prof + 11, (unsigned long long)(C3KT() - t_begin));
			atomicAdd(prof + 12, (unsigned long long)pc[PROF ? 12 : 0]);
			atomicAdd(prof + 13, (unsigned long long)pc[PROF ? 13 : 0]);
		}
#endif
		(void)t_begin;
		(void)prof;
	}
};

#define C3_KERNEL(NAME, WINSZ) C3_KERNEL_(NAME, WINSZ, false, , nullptr)
#ifndef ZMT_EMU
#define C3_KERNEL_PROF(NAME, WINSZ) C3_KERNEL_(NAME, WINSZ, true, C3_PROF_ARG, prof)
#define C3_PROF_ARG , unsigned long long *prof
#else
#define C3_KERNEL_PROF(NAME, WINSZ)
#endif
#define C3_KERNEL_(NAME, WINSZ, PROFILE, EXTRA, PROFP)                                                             \
	extern "C" __global__ void __launch_bounds__(64)                                                           \
	NAME(const u8 *__restrict__ stream, u64 stream_bytes, u32 rec0, u32 nrec, u8 *out_base,                    \
	     const u64 *__restrict__ out_off, const u32 *__restrict__ out_len, const u64 *__restrict__ blk0,       \
	     const u64 *__restrict__ blk_coff, const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk, \
	     const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok, const u32 *__restrict__ blk_ntok,    \
	     const u32 *__restrict__ blk_olen,                                                                     \
	     u32 *__restrict__ status EXTRA)                                                                        \
	{                                                                                                          \
		__shared__ __attribute__((aligned(16))) u8 lds[WINSZ + 16u + C3_CBUF];                             \
		C3<WINSZ, PROFILE>::body(stream, stream_bytes, rec0, nrec, out_base, out_off, out_len, blk0,       \
					 blk_coff, blk_csize, rec_nblk, rec_flags, tok, blk_ntok, blk_olen,       \
					 status, lds, lds + WINSZ + 16u, PROFP);                                   \
	}

C3_KERNEL(zmt_dec_copy3_w4_kernel, 4096u)
C3_KERNEL(zmt_dec_copy3_w8_kernel, 8192u)
C3_KERNEL(zmt_dec_copy3_w16_kernel, 16384u)
C3_KERNEL_PROF(zmt_dec_copy3_w4_kernel_prof, 4096u)
C3_KERNEL_PROF(zmt_dec_copy3_w8_kernel_prof, 8192u)
C3_KERNEL_PROF(zmt_dec_copy3_w16_kernel_prof, 16384u)
