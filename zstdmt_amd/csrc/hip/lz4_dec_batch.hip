/*
 * lz4_dec_batch.hip -- LZ4 frame decoder, lane-per-sequence batches (default decoder).
 *
 * Same contract as zmt_lz4_dec_serial (lz4_dec.hip): one wave decodes one record, replacing
 * LZ4F_decompress at /root/reference/lib/lz4-mt_decompress.c:349-362.  Four independent waves
 * per 256-thread workgroup, no workgroup barriers.
 *
 * The serial kernel moves ~12 output bytes per wave step.  Here a wave decodes 64 sequences at a
 * time (~760 output bytes on enwik-like text):
 *
 *   stage    1 KiB of the compressed block is copied to LDS (coalesced 16 B/lane).
 *   spec     every lane treats 8 of the first 512 staged bytes *as if* they were tokens and
 *            computes where the following token would be ("next").  Only ~1 in 7 is a real token.
 *   walk     the real token chain is then a scalar pointer chase through registers
 *            (v_readlane per sequence, no memory access) that marks token positions in bit
 *            masks; mbcnt/popcount compaction hands sequence k's position to lane k.
 *   decode   lane k reads its own token / lengths / offset from LDS; a wave prefix sum gives
 *            every sequence its output position.
 *   literals lane k copies its literals LDS -> LDS output window with unaligned 8-byte accesses.
 *   matches  sources older than the sliding 8 KiB LDS window come from HBM/L2 (already flushed;
 *            all loads of a batch are issued together), sources in the window are copied LDS -> LDS.  85 % of matches do not depend on the same batch; the
 *            rest resolve in a few rounds under a watermark rule (a match is ready once every
 *            byte before the first unfinished match is final).
 *   flush    the batch's bytes leave the window as coalesced 16-byte stores.
 *
 * Anything unusual (literal-only last sequence of a block, lengths > 64, data not staged, stored
 * blocks, sources straddling the window/HBM frontier) takes a wave-cooperative path that handles one
 * sequence directly in global memory; highly repetitive data therefore decodes with wide copies.
 *
 * HBM traffic: compressed block read (re-staged windows hit L1/L2), output written once, far
 * match sources read back through L2/MALL.  Algorithmic bytes per record = record + content.
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define CB_STAGE 1024u
#define CB_SLACK 64u
#define SPEC_W 512u
#define WIN 8192u      /* LDS output window: linear, slides forward (no wrap-around) */
#define WIN_KEEP 4096u /* history kept when the window slides */
#define SPAN_MAX 2048u /* a batch produces at most this many bytes */
#define CAP_LEN 64u    /* literal / match lengths above this take the cooperative path */
#define NXT_LAST 0xFFFEu
#define NXT_FAR 0xFFFFu
#define TOKPOS_BYTES 128u
#define WAVE_LDS (CB_STAGE + CB_SLACK + TOKPOS_BYTES + WIN)

/* lane `lane` := val (both wave-uniform); a compare + select, no LDS */
static __device__ __forceinline__ u32 wv_writelane(u32 val, int lane, u32 old)
{
	return wv_lane() == lane ? val : old;
}

static __device__ __forceinline__ void st64u(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }

/* ---- output window: win[pos - wbase] holds chunk position pos, wbase is a multiple of 16 ---- */

/* copy len bytes from a linear source to d; exact except that 1..3 byte runs are written as one
 * dword when `slop` (the caller owns the 3 bytes that follow) */
static __device__ __forceinline__ void copy_units(u8 *d, const u8 *s, u32 len, bool slop)
{
	if (len >= 8) {
		for (u32 i = 0; i + 8 < len; i += 8)
			st64u(d + i, ld64u(s + i));
		st64u(d + len - 8, ld64u(s + len - 8));
	} else if (len >= 4) {
		u32 a = ld32u(s), b = ld32u(s + len - 4);
		st32u(d, a);
		st32u(d + len - 4, b);
	} else if (len) {
		if (slop) {
			st32u(d, ld32u(s));
		} else {
			for (u32 i = 0; i < len; i++)
				d[i] = s[i];
		}
	}
}

/* LZ4 match inside the window: d[0..ml) = (d - off)[0..ml), byte-serial semantics */
static __device__ __forceinline__ void win_match(u8 *d, u32 off, u32 ml)
{
	const u8 *s = d - off;
	if (ml >= 4 && (off >= 8 || off >= ml)) {
		/* units run in order, so with off >= 8 a unit only reads bytes already final */
		copy_units(d, s, ml, false);
	} else {
		/* short period (< 8) overlapping itself: replicate the first period */
		u32 j = 0;
		for (u32 i = 0; i < ml; i++) {
			d[i] = s[j];
			if (++j == off)
				j = 0;
		}
	}
}

/* phase timing (variant "prof" only): cycles per phase, summed over waves with atomics */
#ifdef ZMT_EMU
#define PROF_T() 0ull
#else
#define PROF_T() (PROF ? (u64)clock64() : 0ull)
#endif
#define PROF_ADD(slot, t0)                                                                        \
	do {                                                                                       \
		if (PROF) {                                                                        \
			u64 t1_ = PROF_T();                                                        \
			st.prof[slot] += t1_ - (t0);                                               \
			(t0) = t1_;                                                                \
		}                                                                                  \
	} while (0)

struct DecState {
	u64 prof[12];
	u32 opos;       /* next output position (chunk-absolute) */
	u32 flushed;    /* out[0..flushed) is in global memory */
	u32 valid_from; /* the window holds valid bytes for positions >= max(valid_from, wbase) */
	u32 wbase;      /* chunk position of win[0], multiple of 16 */
	u32 fenced;     /* out[0..fenced) is known to be complete in memory for this wave's loads */
};

/* window -> global for [st.flushed, upto); exact */
static __device__ __forceinline__ void flush_to(DecState &st, const u8 *win, u8 *out, u32 upto, int lane)
{
	u32 f = st.flushed;
	if (upto <= f)
		return;
	const u8 *w = win - st.wbase; /* w[pos] */
	/* head: up to the next multiple of 16 */
	u32 head = (16 - (f & 15)) & 15;
	if (head > upto - f)
		head = upto - f;
	if ((u32)lane < head)
		out[f + lane] = w[f + lane];
	f += head;
	u32 body_end = f + ((upto - f) & ~15u);
	for (u32 pos = f + 16 * (u32)lane; pos < body_end; pos += 1024) {
		const u8 *r = w + pos; /* 16-aligned */
		u64 a = *(const u64 *)r, b = *(const u64 *)(r + 8);
		st64u(out + pos, a);
		st64u(out + pos + 8, b);
	}
	if ((u32)lane < upto - body_end)
		out[body_end + lane] = w[body_end + lane];
	st.flushed = upto;
}

/* window -> global, full 16-byte units only; the partial tail stays pending */
static __device__ __forceinline__ void flush_aligned(DecState &st, const u8 *win, u8 *out, u32 upto, int lane)
{
	u32 end = upto & ~15u;
	if (end > st.flushed)
		flush_to(st, win, out, end, lane);
}

/* make room for output up to `end`: slide the window forward, keeping WIN_KEEP bytes of history */
static __device__ __forceinline__ void win_reserve(DecState &st, u8 *win, u32 end, int lane)
{
	if (end - st.wbase <= WIN)
		return;
	u32 nb = st.opos > WIN_KEEP ? (st.opos - WIN_KEEP) & ~15u : 0;
	if (nb < st.wbase)
		nb = st.wbase;
	if (st.valid_from >= st.opos || nb - st.wbase >= WIN) {
		/* nothing valid to keep (e.g. right after a long cooperative copy) */
		st.wbase = nb;
		return;
	}
	u32 delta = nb - st.wbase;        /* multiple of 16, > 1024 whenever a slide is needed */
	u32 keep = st.opos - nb;          /* bytes to move */
	wv_sync();
	for (u32 o = 0; o < keep; o += 1024) {
		u32 i = o + 16 * (u32)lane;
		u64 a = 0, b = 0;
		if (i < keep) {
			a = *(const u64 *)(win + delta + i);
			b = *(const u64 *)(win + delta + i + 8);
		}
		wv_sync(); /* all lanes have read this slice before anyone overwrites it */
		if (i < keep) {
			*(u64 *)(win + i) = a;
			*(u64 *)(win + i + 8) = b;
		}
	}
	wv_sync();
	st.wbase = nb;
}

/*
 * Cooperative path: decode exactly one sequence at src[ip...] straight into global memory
 * (same arithmetic as decode_block_serial).  Returns new ip, 0xFFFFFFFF on malformed input,
 * or sets *done when the block's final (literal-only) sequence was consumed.
 */
static __device__ u32 one_sequence_global(const u8 *src, u32 slen, u32 ip, u8 *out, DecState &st,
					  const u8 *ring, u32 low, u32 limit, bool *done, int lane)
{
	u32 tok, lit, ml, off, opos;
	flush_to(st, ring, out, st.opos, lane);
	opos = st.opos;
	*done = false;
	if (ip >= slen)
		return 0xFFFFFFFFu;
	tok = uld8(src + ip++);
	lit = tok >> 4;
	if (lit == 15) {
		u32 b;
		do {
			if (ip >= slen)
				return 0xFFFFFFFFu;
			b = uld8(src + ip++);
			lit += b;
		} while (b == 255);
	}
	if (slen - ip < lit || limit - opos < lit)
		return 0xFFFFFFFFu;
	wave_copy(out + opos, src + ip, lit, lane);
	ip += lit;
	opos += lit;
	if (ip == slen) {
		*done = true;
	} else {
		if (slen - ip < 2)
			return 0xFFFFFFFFu;
		off = uld16(src + ip);
		ip += 2;
		ml = tok & 15;
		if (ml == 15) {
			u32 b;
			do {
				if (ip >= slen)
					return 0xFFFFFFFFu;
				b = uld8(src + ip++);
				ml += b;
			} while (b == 255);
		}
		ml += 4;
		if (off == 0 || off > opos - low || limit - opos < ml)
			return 0xFFFFFFFFu;
		wave_mem_fence();
		{
			const u8 *m = out + opos - off;
			u8 *d = out + opos;
			if (off >= ml) {
				if (ml >= 512) {
					u32 n4 = ml & ~255u;
					for (u32 i = (u32)lane * 4; i < n4; i += 256)
						st32u(d + i, ld32u(m + i));
					for (u32 i = n4 + (u32)lane; i < ml; i += 64)
						d[i] = m[i];
				} else {
					for (u32 i = (u32)lane; i < ml; i += 64)
						d[i] = m[i];
				}
			} else {
				for (u32 i = (u32)lane; i < ml; i += 64)
					d[i] = m[i % off];
			}
		}
		opos += ml;
	}
	wave_mem_fence();
	st.opos = opos;
	st.flushed = opos;
	st.fenced = opos;
	st.valid_from = opos;
	return ip;
}

/*
 * Speculative "next token position" of staged position q (see file header).  Branch-free: two
 * LDS dword reads per position.  Lengths above CAP_LEN (hence any 255-continuation) yield NXT_FAR.
 */
static __device__ __forceinline__ u32 spec_next(const u8 *cb, u32 q, u32 nst, bool block_ends)
{
	const u32 w = ld32u(cb + (q < CB_STAGE ? q : 0)); /* token, then a possible length byte */
	const u32 tok = w & 255;
	const bool lx = (tok >> 4) == 15;
	const u32 lit = (tok >> 4) + (lx ? (w >> 8) & 255 : 0);
	const u32 lend = q + 1 + (lx ? 1 : 0) + lit;      /* offset field position */
	const u32 w2 = ld32u(cb + (lend < CB_STAGE ? lend : 0)); /* offset(2), possible length byte */
	const bool mx = (tok & 15) == 15;
	const u32 ml = (tok & 15) + (mx ? (w2 >> 16) & 255 : 0);
	const u32 m = lend + 2 + (mx ? 1 : 0);
	u32 r = m;
	r = (m > nst || ml + 4 > CAP_LEN) ? NXT_FAR : r;
	r = (lend >= nst) ? ((block_ends && lend == nst) ? NXT_LAST : NXT_FAR) : r;
	r = (q >= nst || lit > CAP_LEN) ? NXT_FAR : r;
	return r;
}

/* one register's worth (64 staged positions) of the token-chain walk; everything is wave-uniform */
#define WALK_STEP(J)                                                                              \
	if (!stop && k < 64 && (p >> 6) == (J)) {                                                  \
		u64 mk = 0;                                                                        \
		do {                                                                               \
			u32 n_ = wv_readlane(nxtv[J], (int)(p & 63));                              \
			if (n_ >= NXT_LAST) {                                                      \
				stop = true;                                                       \
				break;                                                             \
			}                                                                          \
			mk |= 1ull << (p & 63);                                                    \
			k++;                                                                       \
			p = n_;                                                                    \
		} while (k < 64 && (p >> 6) == (J));                                               \
		mask[J] = mk;                                                                      \
	}

/* Decode one LZ4 block; returns ST_OK or ST_BAD_BLOCK.  st.opos advances. */
template <bool PROF>
static __device__ u32 decode_block_batch(const u8 *src, u32 slen, u8 *out, DecState &st, u32 low,
					 u32 limit, u8 *cb, u16 *tokpos, u8 *win, int lane)
{
	u32 ip = 0;
	if (slen == 0)
		return ST_BAD_BLOCK;
	for (;;) {
		u64 tp = PROF_T();
		/* ---------------- stage ---------------- */
		const u32 avail = slen - ip;
		const u32 nst = avail < CB_STAGE ? avail : CB_STAGE;
		const bool block_ends = avail <= CB_STAGE;
		wv_sync(); /* earlier reads of cb are done */
		{
			u32 o = 16u * (u32)lane;
			if (o + 16 <= nst) {
				u64 a = ld64u(src + ip + o), b = ld64u(src + ip + o + 8);
				*(u64 *)(cb + o) = a;
				*(u64 *)(cb + o + 8) = b;
			} else if (o < nst) {
				for (u32 i = o; i < nst; i++)
					cb[i] = src[ip + i];
			}
		}
		wv_sync();
		PROF_ADD(0, tp);

		/* ---------------- spec ---------------- */
		u32 nxtv[8];
		ZMT_UNROLL
		for (int j = 0; j < 8; j++)
			nxtv[j] = spec_next(cb, (u32)j * 64 + (u32)lane, nst, block_ends);
		PROF_ADD(1, tp);

		/* ---------------- walk: mark the real tokens ---------------- */
		u32 p = 0, k = 0;
		bool stop = false;
		u64 mask[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		WALK_STEP(0)
		WALK_STEP(1)
		WALK_STEP(2)
		WALK_STEP(3)
		WALK_STEP(4)
		WALK_STEP(5)
		WALK_STEP(6)
		WALK_STEP(7)
		PROF_ADD(2, tp);

		if (k == 0) {
			bool done;
			ip = one_sequence_global(src, slen, ip, out, st, win, low, limit, &done, lane);
			PROF_ADD(8, tp);
			if (ip == 0xFFFFFFFFu)
				return ST_BAD_BLOCK;
			if (done)
				return ST_OK;
			continue;
		}
		if (PROF) {
			st.prof[9] += 1;
			st.prof[10] += k;
		}

		/* ---------------- compact: sequence i's token position -> lane i ---------------- */
		{
			u32 base = 0;
			ZMT_UNROLL
			for (int j = 0; j < 8; j++) {
				if ((mask[j] >> lane) & 1)
					tokpos[base + wv_mbcnt(mask[j])] = (u16)(j * 64 + lane);
				base += (u32)wv_popc(mask[j]);
			}
		}
		wv_sync();

		/* ---------------- decode own sequence ---------------- */
		const bool act0 = (u32)lane < k;
		u32 lit = 0, ml = 0, off = 1, lsrc = 0;
		{
			const u32 q = act0 ? (u32)tokpos[lane] : 0;
			const u32 w = ld32u(cb + q);
			const u32 tok = w & 255;
			const bool lx = (tok >> 4) == 15;
			const u32 l_ = (tok >> 4) + (lx ? (w >> 8) & 255 : 0);
			const u32 ls_ = q + 1 + (lx ? 1 : 0);
			const u32 lend = ls_ + l_;
			const u32 w2 = ld32u(cb + (lend < CB_STAGE ? lend : 0));
			const bool mx = (tok & 15) == 15;
			if (act0) {
				lit = l_;
				lsrc = ls_;
				off = w2 & 0xFFFF;
				ml = (tok & 15) + (mx ? (w2 >> 16) & 255 : 0) + 4;
			}
		}
		const u32 len = lit + ml;
		const u32 incl = wv_scan_incl(act0 ? len : 0);

		/* a batch never produces more than SPAN_MAX bytes: cut it where it would */
		{
			u64 over = wv_ballot(act0 && incl > SPAN_MAX);
			if (over) {
				u32 cut = (u32)wv_ffs(over) - 1; /* >= 1: a single sequence is <= 128 bytes */
				k = cut;
				p = (u32)tokpos[cut];
				p = wv_readfirst(p);
			}
		}
		const u32 op = st.opos + incl - len; /* start of this sequence's literals */
		const u32 mpos = op + lit;           /* start of its match */
		const u32 src_pos = mpos - off;
		const u32 eff = ml < off ? ml : off; /* bytes of source actually read */

		/* validity: offsets inside the window, output inside the block */
		{
			bool bad = (u32)lane < k && (off == 0 || off > mpos - low);
			u32 total = wv_readlane(incl, (int)(k - 1));
			if (wv_any(bad) || limit - st.opos < total)
				return ST_BAD_BLOCK;
		}

		/* classify sources: window, global, or straddling the frontier (cuts the batch) */
		u32 batch_end = st.opos + wv_readlane(incl, (int)(k - 1));
		win_reserve(st, win, batch_end, lane);
		u32 near_lo = st.valid_from > st.wbase ? st.valid_from : st.wbase;
		bool is_far = (u32)lane < k && src_pos < near_lo;
		{
			bool straddle = is_far && src_pos + eff > st.flushed;
			u64 sm = wv_ballot(straddle);
			if (sm) {
				u32 cut = (u32)wv_ffs(sm) - 1;
				if (cut == 0) {
					bool done;
					ip = one_sequence_global(src, slen, ip, out, st, win, low, limit,
								 &done, lane);
					if (ip == 0xFFFFFFFFu)
						return ST_BAD_BLOCK;
					if (done)
						return ST_OK;
					continue;
				}
				k = cut;
				p = wv_readfirst((u32)tokpos[cut]);
				batch_end = st.opos + wv_readlane(incl, (int)(cut - 1));
			}
		}
		const bool act = (u32)lane < k;
		is_far = is_far && act;
		u8 *const w0 = win - st.wbase; /* w0[pos] */
		PROF_ADD(3, tp);

		/* ---------------- far matches: loads first (one round trip), stores later -------- */
		u64 fv[8];
		const bool far_plain = is_far && off >= ml;
		u32 far_trips = 0;
		{
			/* out[] below st.fenced is known complete; fence only when a source is newer */
			u32 need = is_far ? src_pos + eff : 0;
			if (wv_any(need > st.fenced)) {
				wave_mem_fence();
				st.fenced = st.flushed;
			}
			u64 fm = wv_ballot(far_plain);
			if (fm) {
				/* widest far match of the batch, in 8-byte units (wave-uniform) */
				u32 mx = 0;
				for (u32 t = 1; t <= 8; t++)
					if (wv_any(far_plain && ml > 8 * (t - 1)))
						mx = t;
				far_trips = mx;
				const u8 *g = out + src_pos;
				ZMT_UNROLL
				for (u32 t = 0; t < 8; t++) {
					fv[t] = 0;
					if (t < far_trips && far_plain && ml > 8 * t) {
						u32 o = 8 * t + 8 <= ml ? 8 * t : (ml >= 8 ? ml - 8 : 0);
						fv[t] = ld64u(g + o);
					}
				}
			}
		}

		/* ---------------- literals ---------------- */
		if (act)
			copy_units(w0 + op, cb + lsrc, lit, true);
		PROF_ADD(4, tp);

		/* ---------------- far matches: stores (after the literals: a 1..3 byte literal run is
		 * stored as one dword that spills into the lane's own match region) ---------------- */
		if (far_trips) {
			ZMT_UNROLL
			for (u32 t = 0; t < 8; t++) {
				if (t < far_trips && far_plain && ml > 8 * t) {
					u8 *d = w0 + mpos;
					if (ml >= 8) {
						u32 o = 8 * t + 8 <= ml ? 8 * t : ml - 8;
						st64u(d + o, fv[t]);
					} else {
						/* 4..7 bytes: the 8 bytes loaded at offset 0 cover them */
						st32u(d, (u32)fv[0]);
						st32u(d + ml - 4, (u32)(fv[0] >> (8 * (ml - 4))));
					}
				}
			}
		}
		if (is_far && !far_plain) {
			/* self-overlapping match whose first period lies in global memory: fetch the
			 * period, then replicate it inside the window */
			copy_units(w0 + mpos, out + src_pos, off, false);
			win_match(w0 + mpos + off, off, ml - off);
		}
		wv_sync();
		PROF_ADD(5, tp);

		/* ---------------- near matches, watermark rounds ---------------- */
		{
			bool fin = !(act && !is_far);
			for (;;) {
				u64 unf = wv_ballot(!fin);
				if (!unf)
					break;
				u32 first = (u32)wv_ffs(unf) - 1;
				u32 W = wv_readlane(mpos, (int)first);
				bool ready = !fin && src_pos + eff <= W;
				if (ready) {
					win_match(w0 + mpos, off, ml);
					fin = true;
				}
				wv_sync();
			}
		}
		PROF_ADD(6, tp);

		/* ---------------- flush + advance ---------------- */
		st.opos = batch_end;
		flush_aligned(st, win, out, batch_end, lane);
		ip += p;
		PROF_ADD(7, tp);
	}
}

template <bool PROF>
static __device__ __forceinline__ void
dec_batch_body(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
	       const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
	       const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
	       u32 *__restrict__ status, u32 *__restrict__ chk_expect, u32 *__restrict__ chk_valid,
	       unsigned long long *prof_out)
{
	__shared__ __attribute__((aligned(16))) u8 lds[4 * WAVE_LDS];
	const int lane = wv_lane();
	const u32 wave = threadIdx.x >> 6;
	const u32 rec = blockIdx.x * 4 + wave;
	if (rec >= nrec)
		return;
	u8 *cb = lds + wave * WAVE_LDS;
	u16 *tokpos = (u16 *)(cb + CB_STAGE + CB_SLACK);
	u8 *ring = cb + CB_STAGE + CB_SLACK + TOKPOS_BYTES; /* the output window */
	const u8 *r = stream + rec_off[rec];
	const u32 rlen = rec_len[rec];
	u8 *out = out_base + out_off[rec];
	const u32 cap = out_len[rec];
	u32 st = ST_OK, ip, flen;
	FrameInfo fi;
	DecState ds;
	ds.opos = 0;
	ds.flushed = 0;
	ds.valid_from = 0;
	ds.wbase = 0;
	ds.fenced = 0;
	for (int i = 0; i < 12; i++)
		ds.prof[i] = 0;
	u64 t_all = PROF ? PROF_T() : 0;

	if (lane == 0) {
		chk_valid[rec] = 0;
		chk_expect[rec] = 0;
	}
	if (rlen < 12 || uld32(r) != ZMT_SKIP_MAGIC || uld32(r + 4) != 4 ||
	    uld32(r + 8) != rlen - 12) {
		st = ST_BAD_RECORD;
		goto done;
	}
	flen = rlen - 12;
	r += 12;
	st = parse_frame_header(r, flen, fi);
	if (st != ST_OK)
		goto done;
	ip = fi.hdr;
	for (;;) {
		u32 bh, bsz;
		if (flen - ip < 4) {
			st = ST_BAD_BLOCK;
			goto done;
		}
		bh = uld32(r + ip);
		ip += 4;
		if (bh == 0)
			break;
		bsz = bh & 0x7FFFFFFFu;
		if (bsz > fi.blkmax || flen - ip < bsz) {
			st = ST_BAD_BLOCK;
			goto done;
		}
		if (bh & 0x80000000u) {
			if (cap - ds.opos < bsz) {
				st = ST_BAD_BLOCK;
				goto done;
			}
			flush_to(ds, ring, out, ds.opos, lane);
			wave_copy(out + ds.opos, r + ip, bsz, lane);
			ds.opos += bsz;
			ds.flushed = ds.opos;
			ds.valid_from = ds.opos;
		} else {
			u32 room = cap - ds.opos < fi.blkmax ? cap - ds.opos : fi.blkmax;
			st = decode_block_batch<PROF>(r + ip, bsz, out, ds, fi.indep ? ds.opos : 0,
						ds.opos + room, cb, tokpos, ring, lane);
			if (st != ST_OK)
				goto done;
		}
		ip += bsz;
	}
	flush_to(ds, ring, out, ds.opos, lane);
	if ((fi.has_csize && fi.csize != (u64)ds.opos) || ds.opos != cap) {
		st = ST_SIZE_MISMATCH;
		goto done;
	}
	if (fi.has_ccheck) {
		if (flen - ip < 4) {
			st = ST_BAD_BLOCK;
			goto done;
		}
		if (lane == 0) {
			chk_expect[rec] = ld32u(r + ip);
			chk_valid[rec] = 1;
		}
		ip += 4;
	}
	if (ip != flen)
		st = ST_TRAILING;
done:
	if (lane == 0)
		status[rec] = st;
#ifndef ZMT_EMU
	if (PROF && lane == 0) {
		ds.prof[11] = PROF_T() - t_all;
		for (int i = 0; i < 12; i++)
			atomicAdd(prof_out + i, (unsigned long long)ds.prof[i]);
	}
#endif
}

extern "C" __global__ void __launch_bounds__(256)
zmt_lz4_dec_batch(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
		  const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
		  const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
		  u32 *__restrict__ status, u32 *__restrict__ chk_expect, u32 *__restrict__ chk_valid)
{
	dec_batch_body<false>(stream, rec_off, rec_len, nrec, out_base, out_off, out_len, status,
			      chk_expect, chk_valid, nullptr);
}

#ifndef ZMT_EMU
/* same kernel with per-phase cycle counters (gpumt_set_variant("lz4_dec", 2)) */
extern "C" __global__ void __launch_bounds__(256)
zmt_lz4_dec_batch_prof(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
		       const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
		       const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
		       u32 *__restrict__ status, u32 *__restrict__ chk_expect,
		       u32 *__restrict__ chk_valid, unsigned long long *prof_out)
{
	dec_batch_body<true>(stream, rec_off, rec_len, nrec, out_base, out_off, out_len, status,
			     chk_expect, chk_valid, prof_out);
}
#endif
