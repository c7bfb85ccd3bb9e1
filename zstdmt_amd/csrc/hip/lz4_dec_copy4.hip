/*
 * lz4_dec_copy4.hip -- copy stage of the LZ4 frame decoder, round 4 ("copy4").
 *
 * Wave per record, the token list of the parse kernel in, the content out (replaces LZ4F_decompress at
 * /root/reference/lib/lz4-mt_decompress.c:349-362 together with the frames / parse kernels).  The design is
 * copy3's (lz4_dec_copy3.hip: power-of-two LDS ring of the record's output, batches the wave cuts itself,
 * LDS-DMA staging a batch ahead, sources before the ring fetched from the output in memory behind the literal
 * phase, everything unusual executed generically through the ring) with TWO SEQUENCES PER LANE:
 *
 * copy3 was measured issue-bound (profiles/r03_sq_counters.json: 9.74 G wave-instructions per 8 GiB, 46 % of
 * them scalar / branch -- the uniform control flow of a batch: waits, cuts, readlanes, flush and prefetch
 * addressing, round control -- at 1.6 per cycle per CU against a ceiling of 1.7).  A batch here is up to 128
 * sequences, lane l holding sequences 2 l and 2 l + 1 (adjacent in the stream AND in the output), so the
 * per-batch scalar work is paid once per 128 sequences instead of once per 64, the prefix sum once per 128, and
 * the dependency rounds run over twice the sequences per wave instruction.  Further differences:
 *   - field decode reads 4 staged bytes per field group (two dword reads, one funnel shift) instead of 8;
 *   - one classification value per sequence (<= 64: fits a batch; 0xFFFF: needs the generic path; 0x8000: not
 *     staged) and two ballots give the cut; no separate "run of small ones" pass (lengths are summed raw, the
 *     prefix that passes is what counts);
 *   - the stage is 2 KiB (two LDS-DMA instructions), the output of a batch at most 2 KiB.
 * Ring variants other than 4 KiB stay with copy3 (gpumt_set_variant("lz4_copy", 3)).
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define C4_CSTAGE 2048u
#define C4_CSLACK 32u
#define C4_CBUF (C4_CSTAGE + C4_CSLACK) /* two of them: the next batch's bytes arrive while this one runs; once a
					  * batch's literals are copied its buffer holds the 128 x 16 bytes fetched for
					  * matches sourced before the ring */
#define C4_BLK_STORED 0x80000000u
#define C4_XOUT 2048u /* most output bytes of one batch */
#define C4_NSEQ 128u
#define C4_HARD 0xFFFFu
#define C4_UNSTAGED 0x8000u

#ifndef ZMT_EMU
#define C4KT() (PROF ? (u64)clock64() : 0ull)
#else
#define C4KT() 0ull
#endif
#define C4PC(i)                                                                                    \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = C4KT();                                                     \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

static __device__ __forceinline__ void c4_st64(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }
/* four bytes at any LDS address as four byte stores (tools/ubench/lds_cost.hip: an LDS access that is not dword
 * aligned costs the pipe one cycle per ACTIVE lane, a byte store 4.6 cycles per wave instruction at any address) */
static __device__ __forceinline__ void c4_st32b(u8 *p, u32 v)
{
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
static __device__ __forceinline__ u64 c4_tok_base(u64 coff, u32 gb) { return ((coff / 3) & ~63ull) + 128ull * gb; }

/* 2 KiB of the stream into LDS, lane l moving bytes [16 l, 16 l + 16) of each KiB; `g` is 16-byte aligned */
static __device__ __forceinline__ void c4_stage(u8 *cb, const u8 *g, u32 nbytes, int lane)
{
#ifdef ZMT_EMU
	if (16u * (u32)lane < nbytes)
		__builtin_memcpy(cb + 16 * lane, g + 16 * lane, 16);
	if (1024u + 16u * (u32)lane < nbytes)
		__builtin_memcpy(cb + 1024 + 16 * lane, g + 1024 + 16 * lane, 16);
#else
	if (16u * (u32)lane < nbytes)
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 16 * lane),
						 (__attribute__((address_space(3))) void *)cb, 16, 0, 0);
	if (1024u + 16u * (u32)lane < nbytes)
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 1024 + 16 * lane),
						 (__attribute__((address_space(3))) void *)(cb + 1024), 16, 0, 0);
#endif
}
static __device__ __forceinline__ void c4_wait_vm()
{
#ifndef ZMT_EMU
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	wv_sync();
}

/* staged bytes at offset o (any alignment): aligned dword reads + funnel shifts */
static __device__ __forceinline__ u32 c4_ld32s(const u8 *base, u32 o)
{
	const u32 *w = (const u32 *)(base + (o & ~3u));
	return wv_alignbyte(w[1], w[0], o);
}
static __device__ __forceinline__ u64 c4_ld64s(const u8 *base, u32 o)
{
	const u32 *w = (const u32 *)(base + (o & ~3u));
	const u32 a0 = w[0], a1 = w[1], a2 = w[2];
	return (u64)wv_alignbyte(a1, a0, o) | ((u64)wv_alignbyte(a2, a1, o) << 32);
}

template <u32 WIN, bool PROF = false> struct C4 {
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
			c4_st64(out + pos, a);
			c4_st64(out + pos + 8, b);
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
			c4_st64(out + pos, a);
			c4_st64(out + pos + 8, b);
		}
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
	 * complete and does not overlap it.  4-byte pieces placed so that ONE per-lane condition covers all lengths up
	 * to 16: bytes 0-3 and the last 4 always, bytes 4-7 and the 4 before the last 4 when the match has 8 or more;
	 * the middle of the rare long one in 8-byte steps */
	template <bool BYTES> static __device__ __forceinline__ void match(u8 *ring, u32 mpos, u32 ml, const u8 *sb, u32 so, u32 sm)
	{
		u8 *const d = ring + (mpos & MASK); /* a batch lies inside one lap: no wrap on the destination side */
		const bool wide = ml >= 8u;
		const u32 tl = wide ? ml - 8u : ml - 4u;
		const u64 a = ld64m(sb, so, sm), b = ld64m(sb, so + tl, sm);
		if (ml > 16u) {
			for (u32 i = 8; i + 8 < ml; i += 8)
				c4_st64(d + i, ld64m(sb, so + i, sm));
		}
		if (BYTES) { /* the pass most lanes take */
			c4_st32b(d, (u32)a);
			c4_st32b(d + ml - 4u, wide ? (u32)(b >> 32) : (u32)b);
			if (wide) {
				c4_st32b(d + 4, (u32)(a >> 32));
				c4_st32b(d + tl, (u32)b);
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

	/* literal run of 1..64 bytes from the stage to the ring; a 4-byte piece may spill <= 3 bytes into the lane's own
	 * match, which is written later */
	static __device__ __forceinline__ void literals(u8 *ring, const u8 *cb, u32 op, u32 lsrc, u32 lit)
	{
		u8 *const dl = ring + (op & MASK);
		const u64 a = c4_ld64s(cb, lsrc);
		c4_st32b(dl, (u32)a);
		if (lit > 4u) {
			c4_st32b(dl + 4, (u32)(a >> 32));
			if (lit > 8u) {
				c4_st64(dl + lit - 8u, c4_ld64s(cb, lsrc + lit - 8u));
				for (u32 i = 8; i + 8 < lit; i += 8)
					c4_st64(dl + i, c4_ld64s(cb, lsrc + i));
			}
		}
	}

	/* fields of the sequence whose token sits at stage offset qr (0xFFFF: no such sequence): literal run, match
	 * length, offset, stage offset of the literals, and the class -- <= 64: both runs <= 64 and fully staged (fits a
	 * batch); C4_HARD: a run above 64 (255 length bytes make 270 / 274) or the block's last sequence (generic path);
	 * C4_UNSTAGED: not (fully) inside the stage */
	static __device__ __forceinline__ void fields(const u8 *cb, u32 qr, bool is_last, u32 &lit, u32 &ml, u32 &off, u32 &lsrc, u32 &cls)
	{
		const bool v1 = qr <= C4_CSTAGE - 12u; /* the token's first bytes are staged */
		const u32 wl = c4_ld32s(cb, v1 ? qr : 0u);
		const u32 tokb = wl & 255u;
		const bool lx = (tokb >> 4) == 15u, mx = (tokb & 15u) == 15u;
		lit = (tokb >> 4) + (lx ? (wl >> 8) & 255u : 0u);
		lsrc = qr + 1u + (lx ? 1u : 0u);
		const u32 mo = lsrc + lit; /* where the offset sits */
		const bool v2 = (v1 ? mo : 0xFFFFu) <= C4_CSTAGE - 4u; /* ... and its last ones */
		const u32 w2 = c4_ld32s(cb, v2 ? mo : 0u);
		off = w2 & 0xFFFFu;
		ml = (tokb & 15u) + 4u + (mx ? (w2 >> 16) & 255u : 0u);
		const u32 run = lit > (v2 ? ml : 0u) ? lit : (v2 ? ml : 0u);
		const u32 big = (is_last | (run > 64u)) ? C4_HARD : run;
		cls = v2 ? big : ((v1 && big == C4_HARD) ? C4_HARD : C4_UNSTAGED);
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
		u64 pc[PROF ? 14 : 1] = {0}, tq = C4KT();
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
		const u32 iA = 2u * (u32)lane, iB = iA + 1u;
		u32 stc = ST_OK;
		St st;
		st.opos = st.flushed = st.valid_from = st.fenced = 0;

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
			if (cs & C4_BLK_STORED) {
				const u32 bsz = cs & 0x7FFFFFFFu;
				flush_to(st, ring, out, st.opos, lane);
				wave_copy(out + st.opos, src, bsz, lane);
				st.opos += bsz;
				st.flushed = st.opos;
				st.valid_from = st.opos; /* the ring does not hold a stored block: sources in it come from memory */
				continue;
			}
			const u32 ntok = wv_readfirst(blk_ntok[gb]);
			const u64 tbase = c4_tok_base(coff, gb);
			const u16 *const tk = tok + tbase;
			const u32 low = indep ? bstart : 0;
			/* ---- the block's sequences in batches this kernel cuts itself ----
			 * A batch = the longest run of sequences of class <= 64 from sequence t0 on -- at most 128 -- whose output is
			 * at most C4_XOUT bytes and does not cross a multiple of the ring size ("lap": stores inside a batch never
			 * wrap).  The sequence that ends the run is executed generically right behind the batch if it is "hard" or a
			 * lap-crosser; one that is merely not staged opens the next batch.  Token positions of the next batch are
			 * loaded as soon as the cut is known, its compressed bytes staged by LDS-DMA into the other buffer.
			 * (token positions past the block's count are slack of the list: readable, never used) */
#define C4_TOKPAIR(T0)                                                                                             \
	(((T0) + iB < ntok) ? ld32u((const u8 *)(tk + (T0) + iA)) : (((T0) + iA < ntok) ? (u32)tk[(T0) + iA] : 0u))
#define C4_TOK2(T0) (((T0) + C4_NSEQ + ((u32)lane & 1u) < ntok) ? (u32)tk[(T0) + C4_NSEQ + ((u32)lane & 1u)] : 0u)
#define C4_STAGE(C0, BUF)                                                                                          \
	do {                                                                                                       \
		const u8 *g_ = src + (C0);                                                                         \
		const u32 a_ = (u32)((size_t)g_ & 15u);                                                            \
		c4_stage(cbuf + (BUF) * C4_CBUF, g_ - a_, cs - (C0) + a_ + 16u, lane);                             \
	} while (0)
			/* fields of sequence i (uniform, i < 128) of this batch's lanes: VA of lane i / 2 if i is even, VB if odd */
#define C4_SEL(VA, VB, I) (((I) & 1u) ? wv_readlane(VB, (int)(((I) >> 1) & 63u)) : wv_readlane(VA, (int)(((I) >> 1) & 63u)))
			u32 t0 = 0, cbi = 0;
			u32 q_cur = C4_TOKPAIR(0), q2_cur = C4_TOK2(0);
			if (ntok)
				C4_STAGE(wv_readlane(q_cur, 0) & 0xFFFFu, cbi);
			while (t0 < ntok && stc == ST_OK) {
				C4PC(1);
				c4_wait_vm(); /* the stage and the token positions of this batch */
				C4PC(0);
				const u32 q = q_cur, q2 = q2_cur;
				u8 *const cb = cbuf + cbi * C4_CBUF;
				cbi ^= 1u;
				if (PROF)
					pc[PROF ? 12 : 0]++;
				/* ---------- fields of up to 128 sequences: lane l = sequences 2 l (A) and 2 l + 1 (B) ---------- */
				const u32 rem = ntok - t0;
				const u32 c0 = wv_readlane(q, 0) & 0xFFFFu;
				const u32 al = (u32)((size_t)(src + c0) & 15u);
				const u32 qA = q & 0xFFFFu, qB = q >> 16;
				const u32 qrA = iA < rem ? qA - c0 + al : 0xFFFFu; /* this lane's tokens in the stage */
				const u32 qrB = iB < rem ? qB - c0 + al : 0xFFFFu;
				u32 litA, mlA, offA, lsrcA, clsA, litB, mlB, offB, lsrcB, clsB;
				fields(cb, qrA, iA + 1u == rem, litA, mlA, offA, lsrcA, clsA);
				fields(cb, qrB, iB + 1u == rem, litB, mlB, offB, lsrcB, clsB);
				C4PC(2);
				/* ---------- output positions, then the cut: the prefix of sequences that fit (class, span, lap) ---------- */
				const u32 lenA = litA + mlA, lenB = litB + mlB; /* (raw: at most 544 each whatever the bytes were) */
				const u32 o0 = st.opos;
				const u32 endB = o0 + wv_scan_incl(lenA + lenB);
				const u32 endA = endB - lenB, opA = endA - lenA, opB = endA;
				const u32 lap_end = (o0 | MASK) + 1u;
				const u32 olim = o0 + C4_XOUT < lap_end ? o0 + C4_XOUT : lap_end;
				const u64 nfA = wv_ballot((clsA <= 64u ? endA : 0xFFFFFFFFu) > olim);
				const u64 nfB = wv_ballot((clsB <= 64u ? endB : 0xFFFFFFFFu) > olim);
				const u32 nA = nfA ? 2u * ((u32)wv_ffs(nfA) - 1u) : C4_NSEQ;
				const u32 nB = nfB ? 2u * ((u32)wv_ffs(nfB) - 1u) + 1u : C4_NSEQ;
				const u32 n = nA < nB ? nA : nB; /* (classes past the first failure do not matter) */
				/* the sequence behind the run is executed generically if it is "hard", or if it fits a batch but crosses the
				 * lap boundary on its own (it could not open a batch either) */
				bool single = false, single_hard = false;
				if (n < C4_NSEQ) {
					const u32 cls_n = C4_SEL(clsA, clsB, n);
					single_hard = cls_n == C4_HARD;
					if (cls_n <= 64u) {
						const u32 op_n = C4_SEL(opA, opB, n), end_n = C4_SEL(endA, endB, n);
						single = op_n < lap_end && end_n > lap_end;
					}
					single = single | single_hard;
				}
				const u32 litA_r = litA, mlA_r = mlA, offA_r = offA, litB_r = litB, mlB_r = mlB, offB_r = offB;
				const bool actA = iA < n, actB = iB < n;
				if (!actA) {
					litA = 0;
					mlA = 0;
					offA = 1;
				}
				if (!actB) {
					litB = 0;
					mlB = 0;
					offB = 1;
				}
				const u32 r = n + (single ? 1u : 0u);
				const u32 t0n = t0 + r;
				if (r == 0) {
					stc = ST_BAD_BLOCK; /* cannot happen: the batch's first token is staged, and a sequence that fits and
							     * does not cross the lap fits a batch alone */
					break;
				}
				/* what leaves for memory this batch (issued together, below): the ring up to the batch's start, the
				 * token positions and the stage of the next batch */
#define C4_ISSUE()                                                                                                 \
	do {                                                                                                       \
		flush_aligned(st, ring, out, st.opos & ~15u, lane);                                                \
		q_cur = C4_TOKPAIR(t0n);                                                                           \
		q2_cur = C4_TOK2(t0n);                                                                             \
		if (t0n < ntok) {                                                                                  \
			const u32 qn_ = r < C4_NSEQ ? (wv_readlane(q, (int)((r >> 1) & 63u)) >> (16u * (r & 1u))) & 0xFFFFu \
						    : wv_readlane(q2, (int)(r & 1u));                              \
			C4_STAGE(qn_, cbi);                                                                        \
		}                                                                                                  \
	} while (0)
				if (n == 0) {
					C4_ISSUE();
				} else {
					const u32 o_end = C4_SEL(endA, endB, n - 1u);
					const u32 total = o_end - o0;
					const u32 mposA = opA + litA, mposB = opB + litB;
					const u32 srcA = mposA - offA, srcB = mposB - offB;
					const u32 effA = mlA < offA ? mlA : offA, effB = mlB < offB ? mlB : offB;
					if (wv_any((actA & ((offA == 0) | (offA > mposA - low))) | (actB & ((offB == 0) | (offB > mposB - low)))) |
					    (total > cap - o0)) {
						stc = ST_BAD_BLOCK;
						break;
					}
					/* the ring holds [near_lo, o0): what this batch will overwrite (plus store slack) is gone */
					const u32 hist = o_end + 16u > WIN ? o_end + 16u - WIN : 0u;
					const u32 near_lo = hist > st.valid_from ? hist : st.valid_from;
					const bool farA = actA & (srcA < near_lo), farB = actB & (srcB < near_lo);
					const bool anyfar = wv_any(farA | farB);
					/* a source that straddles the start of the ring or of what memory holds, or an overlapping match
					 * sourced before the ring: possible only right after a stored block or when memory lags far behind;
					 * then this batch goes one sequence at a time */
					if ((st.valid_from > hist || near_lo + 128u > st.flushed) && anyfar &&
					    wv_any((farA && (srcA + effA > near_lo || srcA + effA > st.flushed || offA < mlA)) |
						   (farB && (srcB + effB > near_lo || srcB + effB > st.flushed || offB < mlB)))) {
						for (u32 i = 0; i < n; i++) {
							const u32 l1 = C4_SEL(litA, litB, i), m1 = C4_SEL(mlA, mlB, i);
							generic(cb + C4_SEL(lsrcA, lsrcB, i), l1, C4_SEL(offA, offB, i), m1, ring, out, st, lane);
						}
						C4_ISSUE();
					} else {
						C4PC(3);
						/* ---- sources before the ring: 16 bytes of the output in memory each, loads first ---- */
						u64 fA0 = 0, fA1 = 0, fB0 = 0, fB1 = 0;
						if (anyfar) {
							if (wv_any((farA & (srcA + effA > st.fenced)) | (farB & (srcB + effB > st.fenced)))) {
								wave_mem_fence();
								st.fenced = st.flushed;
							}
							if (farA) {
								fA0 = ld64u(out + srcA);
								if (mlA > 8u)
									fA1 = ld64u(out + srcA + 8);
							}
							if (farB) {
								fB0 = ld64u(out + srcB);
								if (mlB > 8u)
									fB1 = ld64u(out + srcB + 8);
							}
						}
						C4PC(4);
						/* ---- literals: only lanes that have some ---- */
						if (litA != 0)
							literals(ring, cb, opA, lsrcA, litA);
						if (litB != 0)
							literals(ring, cb, opB, lsrcB, litB);
						wv_sync();
						C4PC(5);
						if (anyfar) {
							/* the literals are out of the stage: it now holds the fetched sources, 2 x 16 bytes per lane */
							if (farA) {
								*(u64 *)(cb + 32u * (u32)lane) = fA0;
								*(u64 *)(cb + 32u * (u32)lane + 8) = fA1;
								/* the rare long one: the rest straight from memory into the ring (complete, unordered) */
								for (u32 i = 16; i < mlA; i += 8) {
									const u32 o = i + 8 <= mlA ? i : mlA - 8;
									c4_st64(ring + ((mposA + o) & MASK), ld64u(out + srcA + o));
								}
							}
							if (farB) {
								*(u64 *)(cb + 32u * (u32)lane + 16) = fB0;
								*(u64 *)(cb + 32u * (u32)lane + 24) = fB1;
								for (u32 i = 16; i < mlB; i += 8) {
									const u32 o = i + 8 <= mlB ? i : mlB - 8;
									c4_st64(ring + ((mposB + o) & MASK), ld64u(out + srcB + o));
								}
							}
							wv_sync();
						}
						C4PC(6);
						/* every load of this batch has landed: now the stores and the prefetches leave */
						C4_ISSUE();
						C4PC(9);
						/* ---- matches: complete sources first (one path: ring or fetched slot), then watermark rounds ---- */
						const bool ovlA = offA < mlA, ovlB = offB < mlB;
						bool finA = !(actA & (mlA != 0)), finB = !(actB & (mlB != 0));
						{
							const bool r1 = (!finA) & (farA | (srcA + effA <= o0)) & !ovlA;
							if (r1) {
								const u8 *const sb = farA ? cb + 32u * (u32)lane : ring;
								match<true>(ring, mposA, (farA && mlA > 16u) ? 16u : mlA, sb, farA ? 0u : srcA, farA ? ~0u : MASK);
								finA = true;
							}
						}
						{
							const bool r1 = (!finB) & (farB | (srcB + effB <= o0)) & !ovlB;
							if (r1) {
								const u8 *const sb = farB ? cb + 32u * (u32)lane + 16u : ring;
								match<true>(ring, mposB, (farB && mlB > 16u) ? 16u : mlB, sb, farB ? 0u : srcB, farB ? ~0u : MASK);
								finB = true;
							}
						}
						wv_sync();
						C4PC(7);
						for (;;) {
							const u64 unf = wv_ballot(!finA | !finB);
							if (!unf)
								break;
							if (PROF)
								pc[PROF ? 13 : 0]++;
							/* everything in front of the first unfinished match is complete */
							const u32 first = (u32)wv_ffs(unf) - 1;
							const u32 W = wv_readlane(!finA ? mposA : mposB, (int)first);
							const bool goA = (!finA) & (srcA + effA <= W), goB = (!finB) & (srcB + effB <= W);
							if (goA & !ovlA)
								match<false>(ring, mposA, mlA, ring, srcA, MASK);
							if (goB & !ovlB)
								match<false>(ring, mposB, mlB, ring, srcB, MASK);
							if (wv_any((goA & ovlA) | (goB & ovlB))) { /* (offset < length: 0.2 % of the matches) */
								if (goA & ovlA)
									match_ovl(ring, mposA, offA, mlA);
								if (goB & ovlB)
									match_ovl(ring, mposB, offB, mlB);
							}
							finA = finA | goA;
							finB = finB | goB;
							wv_sync();
						}
						C4PC(8);
						st.opos = o_end;
					}
				}
				if (single && !single_hard) {
					/* ---------- a sequence that fits a batch but crosses the lap boundary: byte by byte, wrapping ---------- */
					const u32 s_lit = C4_SEL(litA_r, litB_r, n), s_ml = C4_SEL(mlA_r, mlB_r, n);
					const u32 s_off = C4_SEL(offA_r, offB_r, n);
					/* block position of its literals (the stage has been overwritten by the fetched sources) */
					const u32 s_lpos = C4_SEL(qA + (lsrcA - qrA), qB + (lsrcB - qrB), n);
					if (s_off == 0 || s_off > st.opos + s_lit - low || s_lit + s_ml > cap - st.opos) {
						stc = ST_BAD_BLOCK;
						break;
					}
					generic(src + s_lpos, s_lit, s_off, s_ml, ring, out, st, lane);
					C4PC(10);
				} else if (single) {
					/* ---------- one sequence of any shape (fields from memory, wave-uniform) ---------- */
					const u32 qq = n < C4_NSEQ ? C4_SEL(qA, qB, n) : wv_readlane(q2, 0);
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
					C4PC(10);
				}
				t0 = t0n;
			}
#undef C4_TOKPAIR
#undef C4_TOK2
#undef C4_STAGE
#undef C4_ISSUE
#undef C4_SEL
			if (stc == ST_OK && (t0 != ntok || st.opos != bstart + olen))
				stc = ST_BAD_BLOCK;
		}
		if (stc == ST_OK || stc == ST_SIZE_MISMATCH)
			flush_to(st, ring, out, st.opos, lane);
		if (stc == ST_OK && st.opos != cap)
			stc = ST_SIZE_MISMATCH;
		if (lane == 0 && stc != ST_OK)
			status[rec] = stc;
#ifndef ZMT_EMU
		if (PROF && prof && lane == 0) {
			for (int i = 0; i < 11; i++)
				atomicAdd(prof + i, (unsigned long long)pc[PROF ? i : 0]);
			atomicAdd(prof + 11, (unsigned long long)(C4KT() - t_begin));
			atomicAdd(prof + 12, (unsigned long long)pc[PROF ? 12 : 0]);
			atomicAdd(prof + 13, (unsigned long long)pc[PROF ? 13 : 0]);
		}
#endif
		(void)t_begin;
		(void)prof;
	}
};

#define C4_KERNEL(NAME, WINSZ) C4_KERNEL_(NAME, WINSZ, false, , nullptr)
#ifndef ZMT_EMU
#define C4_KERNEL_PROF(NAME, WINSZ) C4_KERNEL_(NAME, WINSZ, true, C4_PROF_ARG, prof)
#define C4_PROF_ARG , unsigned long long *prof
#else
#define C4_KERNEL_PROF(NAME, WINSZ)
#endif
#define C4_KERNEL_(NAME, WINSZ, PROFILE, EXTRA, PROFP)                                                             \
	extern "C" __global__ void __launch_bounds__(64)                                                           \
	NAME(const u8 *__restrict__ stream, u64 stream_bytes, u32 rec0, u32 nrec, u8 *out_base,                    \
	     const u64 *__restrict__ out_off, const u32 *__restrict__ out_len, const u64 *__restrict__ blk0,       \
	     const u64 *__restrict__ blk_coff, const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk, \
	     const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok, const u32 *__restrict__ blk_ntok,    \
	     const u32 *__restrict__ blk_olen,                                                                     \
	     u32 *__restrict__ status EXTRA)                                                                        \
	{                                                                                                          \
		__shared__ __attribute__((aligned(16))) u8 lds[WINSZ + 16u + 2u * C4_CBUF];                        \
		C4<WINSZ, PROFILE>::body(stream, stream_bytes, rec0, nrec, out_base, out_off, out_len, blk0,       \
					 blk_coff, blk_csize, rec_nblk, rec_flags, tok, blk_ntok, blk_olen,       \
					 status, lds, lds + WINSZ + 16u, PROFP);                                   \
	}

C4_KERNEL(zmt_dec_copy4_kernel, 4096u)
C4_KERNEL_PROF(zmt_dec_copy4_kernel_prof, 4096u)
