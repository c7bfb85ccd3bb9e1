/*
 * lz4_dec_copy2.hip -- copy stage of the LZ4 frame decoder, second revision ("copy2").
 *
 * Same contract and inputs as zmt_dec_copy_kernel of lz4_dec_split.hip (wave per record, 64 sequences
 * per step from the token list of the parse kernel; replaces LZ4F_decompress at
 * /root/reference/lib/lz4-mt_decompress.c:349-362 together with the frames / parse kernels).
 *
 * What the SQ counters of the first copy kernel say (profiles/r01_sq_counters.json): 48 % of its
 * wave cycles are spent parked on s_waitcnt and it cannot take more than 4 waves per SIMD (128
 * VGPRs; 5 waves spill 51 registers).  The LDS pipe is NOT the limit (turning the two misaligned
 * field reads of every sequence into aligned ones bought 3 %).  So this revision is built for
 * waves per SIMD and for few dependent round trips per step:
 *   - 6 waves per SIMD: <= 80 VGPRs, 6 KiB of LDS per wave (4 KiB output window, 1 KiB staged
 *     compressed bytes, 1 KiB of slots for matches sourced before the window);
 *   - branch-free copies: literals go out as up to four 4-byte pieces (they may spill <= 3 bytes
 *     into the lane's own match, which is written afterwards), matches of <= 16 bytes as four
 *     4-byte pieces at 0, 4, len-8, len-4 with all loads ahead of all stores -- no length
 *     classes, no per-lane loops for 96 % of the sequences;
 *   - matches sourced before the window fetch 16 bytes from the output with two 8-byte global loads
 *     issued before the literal copy and parked in an LDS slot, after which they are ordinary
 *     matches whose source happens to be the slot: one code path, no per-length cases;
 *   - the running output position is summed here (DPP scan), the bidx list of the parse kernel is
 *     not read.
 * Everything unusual -- literal runs or matches above 64 bytes, the literal-only last sequence of a
 * block, overlapping matches, sources straddling the window start -- is cut out of the batch and
 * handled one sequence at a time by the whole wave, straight to global memory.
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#ifndef C2_WIN
#define C2_WIN 4096u   /* LDS output window per wave */
#endif
#ifndef C2_KEEP
#define C2_KEEP 2048u  /* history kept when the window slides */
#endif
#ifndef C2_WPS
#define C2_WPS 6       /* waves per SIMD the register budget is cut for */
#endif
#define C2_SPAN (C2_WIN - C2_KEEP) /* output bytes one sub-batch may add */
#define C2_CAP 64u
#define C2_CSTAGE 1024u
#define C2_CSLACK 32u
#define C2_FSLOTS 1024u /* 64 x 16 bytes */
#define C2_LDS_WAVE (C2_WIN + C2_CSTAGE + C2_CSLACK + C2_FSLOTS)
#define C2_BLK_STORED 0x80000000u

static __device__ __forceinline__ void c2_st64(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }
static __device__ __forceinline__ u64 c2_tok_base(u64 coff, u32 gb) { return ((coff / 3) & ~63ull) + 128ull * gb; }

/* four staged bytes at offset o (any alignment): two aligned dword reads + funnel shift */
static __device__ __forceinline__ u32 c2_ld32a(const u8 *base, u32 o)
{
	const u32 *w = (const u32 *)(base + (o & ~3u));
	return wv_alignbyte(w[1], w[0], o & 3u);
}

struct C2State {
	u32 opos, flushed, valid_from, wbase, fenced;
};

static __device__ __forceinline__ void c2_flush_to(C2State &st, const u8 *win, u8 *out, u32 upto, int lane)
{
	u32 f = st.flushed;
	if (upto <= f)
		return;
	const u8 *w = win - st.wbase;
	u32 head = (16 - (f & 15)) & 15;
	if (head > upto - f)
		head = upto - f;
	if ((u32)lane < head)
		out[f + lane] = w[f + lane];
	f += head;
	const u32 body_end = f + ((upto - f) & ~15u);
	for (u32 pos = f + 16 * (u32)lane; pos < body_end; pos += 1024) {
		const u8 *r = w + pos;
		const u64 a = *(const u64 *)r, b = *(const u64 *)(r + 8);
		c2_st64(out + pos, a);
		c2_st64(out + pos + 8, b);
	}
	if ((u32)lane < upto - body_end)
		out[body_end + lane] = w[body_end + lane];
	st.flushed = upto;
}

/* make room for output up to `end`: keep the last C2_KEEP bytes, 16-byte aligned */
static __device__ __forceinline__ void c2_reserve(C2State &st, u8 *win, u32 end, int lane)
{
	if (end - st.wbase <= C2_WIN)
		return;
	u32 nb = st.opos > C2_KEEP ? (st.opos - C2_KEEP) & ~15u : 0;
	if (nb < st.wbase)
		nb = st.wbase;
	if (st.valid_from >= st.opos || nb - st.wbase >= C2_WIN) {
		st.wbase = nb;
		return;
	}
	const u32 delta = nb - st.wbase, keep = st.opos - nb;
	wv_sync();
	for (u32 o = 0; o < keep; o += 1024) {
		const u32 i = o + 16 * (u32)lane;
		u64 a = 0, b = 0;
		if (i < keep) {
			a = *(const u64 *)(win + delta + i);
			b = *(const u64 *)(win + delta + i + 8);
		}
		wv_sync();
		if (i < keep) {
			*(u64 *)(win + i) = a;
			*(u64 *)(win + i + 8) = b;
		}
	}
	wv_sync();
	st.wbase = nb;
}

/* one unusual sequence, by the whole wave, straight to global memory (fields are wave-uniform) */
static __device__ void c2_long(const u8 *lsrc, u32 lit, u32 off, u32 ml, u8 *out, C2State &st, const u8 *win,
			       int lane)
{
	c2_flush_to(st, win, out, st.opos, lane);
	u32 opos = st.opos;
	wave_copy(out + opos, lsrc, lit, lane);
	opos += lit;
	if (ml) {
		wave_mem_fence();
		const u8 *m = out + opos - off;
		u8 *d = out + opos;
		if (off >= ml) {
			u32 i = 0;
			if (ml >= 512) {
				const u32 n4 = ml & ~255u;
				for (i = (u32)lane * 4; i < n4; i += 256)
					st32u(d + i, ld32u(m + i));
				i = n4;
			}
			for (i += (u32)lane; i < ml; i += 64)
				d[i] = m[i];
		} else {
			for (u32 i = (u32)lane; i < ml; i += 64)
				d[i] = m[i % off];
		}
		opos += ml;
	}
	wave_mem_fence();
	st.opos = opos;
	st.flushed = opos;
	st.fenced = opos;
	st.valid_from = opos;
}

/* overlapping match (offset < length) inside the window: strictly forward */
static __device__ __forceinline__ void c2_match_ovl(u8 *d, u32 off, u32 ml)
{
	const u8 *s = d - off;
	if (off >= 8) {
		u32 i = 0;
		for (; i + 8 <= ml; i += 8)
			c2_st64(d + i, ld64u(s + i));
		for (; i < ml; i++)
			d[i] = s[i];
	} else {
		u32 j = 0;
		for (u32 i = 0; i < ml; i++) {
			d[i] = s[j];
			if (++j == off)
				j = 0;
		}
	}
}

/* eight bytes at offset o of an LDS region (any alignment) as three ALIGNED dword reads and two funnel
 * shifts: an aligned ds_read costs 2.5 LDS cycles per wave instruction, a misaligned one a cycle per
 * lane (tools/ubench/lds_ops.hip), and with the copies as lean as they are here the LDS pipe is what
 * this kernel is bound by */
static __device__ __forceinline__ u64 c2_ld64a(const u8 *base, u32 o)
{
	const u32 *w = (const u32 *)(base + (o & ~3u));
	const u32 a0 = w[0], a1 = w[1], a2 = w[2];
	const u32 sh = o & 3u;
	return (u64)wv_alignbyte(a1, a0, sh) | ((u64)wv_alignbyte(a2, a1, sh) << 32);
}

/* match of <= C2_CAP bytes at window offset dofs whose source (region `sb`, offset so) is complete and
 * does not overlap it: first and last 8 (or 4) bytes from aligned reads, the middle of the rare long
 * one in 8-byte steps */
static __device__ __forceinline__ void c2_match(u8 *win, u32 dofs, const u8 *sb, u32 so, u32 ml)
{
	const bool wide = ml >= 8;
	const u32 tl = wide ? ml - 8 : ml - 4;
	const u64 a = c2_ld64a(sb, so), b = c2_ld64a(sb, so + tl);
	if (ml > 16) {
		for (u32 i = 8; i + 8 < ml; i += 8)
			c2_st64(win + dofs + i, ld64u(sb + so + i));
	}
	if (wide) {
		c2_st64(win + dofs, a);
		c2_st64(win + dofs + tl, b);
	} else {
		st32u(win + dofs, (u32)a);
		st32u(win + dofs + tl, (u32)b);
	}
}

#ifdef ZMT_EMU
#define C2_ATTR
#else
#define C2_ATTR __attribute__((amdgpu_waves_per_eu(C2_WPS, C2_WPS)))
#endif

#ifndef ZMT_EMU
#define C2KT() (PROF ? (u64)clock64() : 0ull)
#else
#define C2KT() 0ull
#endif
/* per-phase cycle counters of the profiling build (developer tool, tools/dec_prof.py) */
#define C2PC(i)                                                                                    \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = C2KT();                                                     \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

template <bool PROF>
static __device__ __forceinline__ void
c2_body(const u8 *__restrict__ stream, u64 stream_bytes, u32 rec0, u32 nrec, u8 *out_base,
	const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
	const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
	const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
	const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
	const u32 *__restrict__ blk_ntok, const u32 *__restrict__ blk_olen,
	u32 *__restrict__ status, unsigned long long *prof, u8 *lds)
{
	const int lane = wv_lane();
	u64 pc[PROF ? 12 : 1] = {0}, tq = C2KT();
	const u64 t_begin = tq;
	u64 nbatch = 0;
	const u32 wave = threadIdx.x >> 6;
	const u32 rec = rec0 + blockIdx.x * 4 + wave; /* records [rec0, nrec) of the batch */
	if (rec >= nrec)
		return;
	if (wv_readfirst(status[rec]) != ST_OK)
		return;
	u8 *const win = lds + wave * C2_LDS_WAVE;
	u8 *const cb = win + C2_WIN;                         /* staged compressed bytes of the batch */
	u8 *const fs = cb + C2_CSTAGE + C2_CSLACK;           /* slots for matches sourced before the window */
	u8 *const out = out_base + out_off[rec];
	const u32 cap = out_len[rec];
	const u64 b0 = blk0[rec];
	const u32 nb = wv_readfirst(rec_nblk[rec]);
	const bool indep = wv_readfirst(rec_flags[rec]) & 1;
	u32 stc = ST_OK;
	C2State st;
	st.opos = st.flushed = st.valid_from = st.wbase = st.fenced = 0;

	for (u32 bi = 0; bi < nb && stc == ST_OK; bi++) {
		const u32 gb = (u32)(b0 + bi);
		const u32 cs = wv_readfirst(blk_csize[gb]);
		const u64 coff = blk_coff[gb];
		const u8 *src = stream + coff;
		const u32 olen = wv_readfirst(blk_olen[gb]);
		const u32 bstart = st.opos;
		if (olen == 0xFFFFFFFFu || cap - bstart < olen) {
			stc = ST_BAD_BLOCK;
			break;
		}
		if (cs & C2_BLK_STORED) {
			const u32 bsz = cs & 0x7FFFFFFFu;
			c2_flush_to(st, win, out, st.opos, lane);
			wave_copy(out + st.opos, src, bsz, lane);
			st.opos += bsz;
			st.flushed = st.opos;
			st.valid_from = st.opos;
			continue;
		}
		const u32 ntok = wv_readfirst(blk_ntok[gb]);
		const u16 *tk = tok + c2_tok_base(coff, gb);
		const u32 low = indep ? bstart : 0;
		/* token positions two batches ahead, the batch's compressed bytes (1 KiB from its first
		 * token) one batch ahead */
#define C2_TOK(T0) (((T0) + (u32)lane < ntok) ? (u32)tk[(T0) + lane] : 0u)
#define C2_STAGE(CS0, A, B)                                                                        \
	do {                                                                                       \
		const u32 o_ = (CS0) + 16u * (u32)lane; /* <= 31 bytes past the block: stream slack */ \
		(A) = 0;                                                                           \
		(B) = 0;                                                                           \
		if (o_ < cs + 8) {                                                                 \
			(A) = ld64u(src + o_);                                                     \
			(B) = ld64u(src + o_ + 8);                                                 \
		}                                                                                  \
	} while (0)
		u32 q_cur = C2_TOK(0), q_nxt = C2_TOK(64);
		u64 ca, cbv;
		C2_STAGE(0, ca, cbv);
		u32 cs_cur = 0;
		for (u32 t0 = 0; t0 < ntok && stc == ST_OK; t0 += 64) {
			const u32 k = ntok - t0 < 64 ? ntok - t0 : 64;
			C2PC(10);
			nbatch++;
			wv_sync();
			*(u64 *)(cb + 16u * (u32)lane) = ca;
			*(u64 *)(cb + 16u * (u32)lane + 8) = cbv;
			wv_sync();
			const u32 q = q_cur;
			const u32 cs0 = cs_cur;
			{
				const u32 q_nn = C2_TOK(t0 + 128);
				const u32 cs_n = wv_readlane(q_nxt, 0);
				if (t0 + 64 < ntok)
					C2_STAGE(cs_n, ca, cbv);
				cs_cur = cs_n;
				q_cur = q_nxt;
				q_nxt = q_nn;
			}
			C2PC(0);
			const bool act0 = (u32)lane < k;
			const bool is_last = act0 && t0 + (u32)lane == ntok - 1;
			/* ---- fields of sequence t0 + lane ---- */
			u32 lit = 0, ml = 0, off = 1, lsrc = 0;
			const u32 qr = q - cs0;
			bool staged = qr + 80 <= C2_CSTAGE;
			{
				const u32 w = c2_ld32a(cb, staged ? qr : 0);
				const u32 tokb = w & 255;
				const bool lx = (tokb >> 4) == 15;
				const u32 b1 = (w >> 8) & 255;
				const u32 l_ = (tokb >> 4) + (lx ? b1 : 0);
				const u32 h = q + 1 + (lx ? 1 : 0);
				const u32 lend = h + l_;
				const bool st2 = staged && lend - cs0 + 4 <= C2_CSTAGE;
				const u32 w2 = c2_ld32a(cb, st2 ? lend - cs0 : 0);
				const bool mx = (tokb & 15) == 15;
				const u32 b2 = (w2 >> 16) & 255;
				const bool fast = staged && !(lx && b1 == 255) && (is_last || (st2 && !(mx && b2 == 255)));
				if (act0 && fast) {
					lit = l_;
					lsrc = h;
					if (!is_last) {
						off = w2 & 0xFFFF;
						ml = (tokb & 15) + (mx ? b2 : 0) + 4;
					}
				}
				if (act0 && !fast) {
					/* generic: straight from global memory (the parse kernel validated the chain) */
					const u32 t = src[q];
					u32 l2 = t >> 4, h2 = q + 1;
					if (l2 == 15) {
						u32 b;
						do {
							b = src[h2++];
							l2 += b;
						} while (b == 255);
					}
					lit = l2;
					lsrc = h2;
					if (!is_last) {
						u32 m = h2 + l2;
						off = ld16u(src + m);
						m += 2;
						ml = t & 15;
						if (ml == 15) {
							u32 b;
							do {
								b = src[m++];
								ml += b;
							} while (b == 255);
						}
						ml += 4;
					}
					staged = false;
				}
			}
			C2PC(1);
			const u32 len = lit + ml;
			const u32 incl = wv_scan_incl(len);
			const u32 op = st.opos + incl - len; /* st.opos == start of this batch */
			const u32 mpos = op + lit;
			const u32 src_pos = mpos - off;
			const u32 eff = ml < off ? ml : off;
			if (wv_any(act0 && !is_last && (off == 0 || off > mpos - low))) {
				stc = ST_BAD_BLOCK;
				break;
			}
			if (wv_readlane(incl, (int)(k - 1u)) > cap - st.opos) { /* cannot happen after the olen check */
				stc = ST_BAD_BLOCK;
				break;
			}
			/* sequences handled one at a time: long ones, the literal-only last one, fields that
			 * came from global memory */
				C2PC(2);
			u64 cutm = wv_ballot(act0 && (lit > C2_CAP || ml > C2_CAP || is_last || !staged));
			u32 lo = 0;
			while (lo < k) {
				const u64 rest = cutm & ~((1ull << lo) - 1);
				u32 hi = rest ? (u32)wv_ffs(rest) - 1 : k;
				if (hi > lo) {
					/* ---------- ordinary sequences [lo, hi) ---------- */
					const u32 sub_start = wv_readlane(op, (int)lo);
					{
						const u64 over = wv_ballot((u32)lane >= lo && (u32)lane < hi &&
									   op + len - sub_start > C2_SPAN);
						if (over)
							hi = (u32)wv_ffs(over) - 1; /* > lo: one sequence is <= 128 bytes */
					}
					st.opos = sub_start;
					c2_reserve(st, win, wv_readlane(op + len, (int)(hi - 1)), lane);
					const u32 near_lo = st.valid_from > st.wbase ? st.valid_from : st.wbase;
					{
						/* sources that straddle the window start, and overlapping matches sourced before
						 * it: one at a time */
						const u64 oddm = wv_ballot((u32)lane >= lo && (u32)lane < hi && src_pos < near_lo &&
									   (src_pos + eff > near_lo || src_pos + eff > st.flushed || off < ml));
						if (oddm) {
							const u32 l1 = (u32)wv_ffs(oddm) - 1;
							cutm |= 1ull << l1;
							if (l1 == lo)
								continue;
							hi = l1;
						}
					}
					const bool act = (u32)lane >= lo && (u32)lane < hi;
					const u32 sub_end = wv_readlane(op + len, (int)(hi - 1));
					u8 *const w0 = win - st.wbase;
#ifdef C2_ABL_NOFAR
					const bool is_far = false;
#else
					const bool is_far = act && src_pos < near_lo;
#endif
					/* ---- matches sourced before the window: 16 bytes of the output, loads first ---- */
					u64 f0 = 0, f1 = 0;
					const bool anyfar = wv_any(is_far);
					C2PC(3);
					if (anyfar) {
						if (wv_any(is_far && src_pos + eff > st.fenced)) {
							wave_mem_fence();
							st.fenced = st.flushed;
						}
						if (is_far) {
							f0 = ld64u(out + src_pos);
							if (ml > 8)
								f1 = ld64u(out + src_pos + 8);
						}
					}
					C2PC(4);
					/* ---- literals: 4-byte pieces, may spill <= 3 bytes into the lane's own match ---- */
#ifndef C2_ABL_NOLIT
					if (act && lit) {
						u8 *d = w0 + op;
						const u32 so = lsrc - cs0;
						const u64 a = c2_ld64a(cb, so);
						if (lit <= 4) {
							st32u(d, (u32)a); /* may spill <= 3 bytes into the lane's own match */
						} else {
							c2_st64(d, a);    /* 5..8: spills <= 3 bytes; longer runs end exactly */
							if (lit > 8) {
								c2_st64(d + lit - 8, c2_ld64a(cb, so + lit - 8));
								for (u32 i = 8; i + 8 < lit; i += 8)
									c2_st64(d + i, ld64u(cb + so + i));
							}
						}
					}
#endif
					C2PC(5);
					if (anyfar) {
						if (is_far) {
							*(u64 *)(fs + 16u * (u32)lane) = f0;
							*(u64 *)(fs + 16u * (u32)lane + 8) = f1;
							/* the rare long one: the rest straight from the output into the window
							 * (sources before the window are complete, nothing depends on order) */
							for (u32 i = 16; i < ml; i += 8) {
								const u32 o = i + 8 <= ml ? i : ml - 8;
								c2_st64(w0 + mpos + o, ld64u(out + src_pos + o));
							}
						}
					}
					wv_sync();
					/* ---- matches: complete sources first, then watermark rounds ---- */
					C2PC(6);
					const u8 *const sb = is_far ? fs : win;     /* source region and offset in it */
					const u32 so = is_far ? 16u * (u32)lane : src_pos - st.wbase;
					const u32 dofs = mpos - st.wbase;
					const u32 mlc = (is_far && ml > 16) ? 16u : ml; /* bytes the piece copy writes */
					const bool ovl = off < ml;
#ifdef C2_ABL_NOMATCH
					bool fin = true;
#else
					bool fin = !(act && ml != 0);
#endif
#ifdef C2_ABL_NOROUNDS
					if (!fin) {
#else
					if (!fin && (is_far || src_pos + eff <= sub_start)) {
#endif
						if (ovl)
							c2_match_ovl(w0 + mpos, off, ml);
						else
							c2_match(win, dofs, sb, so, mlc);
						fin = true;
					}
					wv_sync();
					C2PC(7);
					for (;;) {
						const u64 unf = wv_ballot(!fin);
						if (!unf)
							break;
						const u32 first = (u32)wv_ffs(unf) - 1;
						const u32 W = wv_readlane(mpos, (int)first);
						if (!fin && src_pos + eff <= W) {
							if (ovl)
								c2_match_ovl(w0 + mpos, off, ml);
							else
								c2_match(win, dofs, sb, so, mlc);
							fin = true;
						}
						wv_sync();
					}
					C2PC(8);
					st.opos = sub_end;
					{
						const u32 end = sub_end & ~15u;
#ifdef C2_ABL_NOFLUSH
						if (end > st.flushed)
							st.flushed = end;
#else
						if (end > st.flushed)
							c2_flush_to(st, win, out, end, lane);
#endif
					}
					C2PC(9);
					lo = hi;
				}
				if (lo < k && ((cutm >> lo) & 1)) {
					/* ---------- one unusual sequence ---------- */
					const u32 l_lit = wv_readlane(lit, (int)lo), l_ml = wv_readlane(ml, (int)lo);
					const u32 l_off = wv_readlane(off, (int)lo), l_src = wv_readlane(lsrc, (int)lo);
					st.opos = wv_readlane(op, (int)lo);
					c2_long(src + l_src, l_lit, l_off, l_ml, out, st, win, lane);
					lo++;
				}
			}
			if (stc == ST_OK)
				st.opos = wv_readlane(op + len, (int)(k - 1u));
		}
		if (stc == ST_OK && st.opos != bstart + olen)
			stc = ST_BAD_BLOCK;
	}
	c2_flush_to(st, win, out, st.opos, lane);
	if (stc == ST_OK && st.opos != cap)
		stc = ST_SIZE_MISMATCH;
	if (lane == 0 && stc != ST_OK)
		status[rec] = stc;
#ifndef ZMT_EMU
	if (PROF && prof && lane == 0) {
		for (int i = 0; i < 11; i++)
			atomicAdd(prof + i, (unsigned long long)pc[PROF ? i : 0]);
		atomicAdd(prof + 11, (unsigned long long)(C2KT() - t_begin));
		atomicAdd(prof + 12, (unsigned long long)nbatch);
	}
#endif
	(void)nbatch;
	(void)t_begin;
}

extern "C" __global__ void __launch_bounds__(256) C2_ATTR
zmt_dec_copy2_kernel(const u8 *__restrict__ stream, u64 stream_bytes, u32 rec0, u32 nrec, u8 *out_base,
		     const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
		     const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
		     const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
		     const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
		     const u32 *__restrict__ blk_ntok, const u32 *__restrict__ blk_olen,
		     u32 *__restrict__ status)
{
	__shared__ __attribute__((aligned(16))) u8 lds[4 * C2_LDS_WAVE];
	c2_body<false>(stream, stream_bytes, rec0, nrec, out_base, out_off, out_len, blk0, blk_coff, blk_csize, rec_nblk,
		       rec_flags, tok, blk_ntok, blk_olen, status, nullptr, lds);
}

#ifndef ZMT_EMU
extern "C" __global__ void __launch_bounds__(256) /* no register cap: the counters need room */
zmt_dec_copy2_kernel_prof(const u8 *__restrict__ stream, u64 stream_bytes, u32 rec0, u32 nrec, u8 *out_base,
			  const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
			  const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
			  const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
			  const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
			  const u32 *__restrict__ blk_ntok, const u32 *__restrict__ blk_olen,
			  u32 *__restrict__ status, unsigned long long *prof)
{
	__shared__ __attribute__((aligned(16))) u8 lds[4 * C2_LDS_WAVE];
	c2_body<true>(stream, stream_bytes, rec0, nrec, out_base, out_off, out_len, blk0, blk_coff, blk_csize, rec_nblk,
		      rec_flags, tok, blk_ntok, blk_olen, status, prof, lds);
}
#endif
