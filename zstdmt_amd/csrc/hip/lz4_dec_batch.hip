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
 *            (v_readlane per sequence, no memory access); sequence k's position goes to lane k.
 *   decode   lane k reads its own token / lengths / offset from LDS; a wave prefix sum gives
 *            every sequence its output position.
 *   literals lane k copies its literals LDS -> LDS output ring with unaligned 8-byte accesses.
 *   matches  sources older than the 8 KiB LDS ring come from HBM/L2 (already flushed), sources in
 *            the ring are copied LDS -> LDS.  85 % of matches do not depend on the same batch; the
 *            rest resolve in a few rounds under a watermark rule (a match is ready once every
 *            byte before the first unfinished match is final).
 *   flush    the batch's bytes leave the ring as coalesced 16-byte stores.
 *
 * Anything unusual (literal-only last sequence of a block, lengths > 64, data not staged, stored
 * blocks, sources straddling the ring/HBM frontier) takes a wave-cooperative path that handles one
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
#define RING 8192u
#define RMASK (RING - 1)
#define CAP_LEN 64u
#define NXT_LAST 0xFFFEu
#define NXT_FAR 0xFFFFu
#define WAVE_LDS (CB_STAGE + CB_SLACK + RING)

/* lane `lane` := val (both wave-uniform); a compare + select, no LDS */
static __device__ __forceinline__ u32 wv_writelane(u32 val, int lane, u32 old)
{
	return wv_lane() == lane ? val : old;
}

static __device__ __forceinline__ void st64u(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }

/* ---- ring accessors (positions are chunk-absolute, ring index = pos mod RING) ---- */
static __device__ __forceinline__ u64 ring_ld8(const u8 *ring, u32 pos)
{
	u32 i = pos & RMASK;
	if (i <= RING - 8)
		return ld64u(ring + i);
	u64 v = 0;
	for (int b = 0; b < 8; b++)
		v |= (u64)ring[(pos + b) & RMASK] << (8 * b);
	return v;
}
static __device__ __forceinline__ void ring_st8(u8 *ring, u32 pos, u64 v)
{
	u32 i = pos & RMASK;
	if (i <= RING - 8) {
		st64u(ring + i, v);
		return;
	}
	for (int b = 0; b < 8; b++)
		ring[(pos + b) & RMASK] = (u8)(v >> (8 * b));
}
static __device__ __forceinline__ u32 ring_ld4(const u8 *ring, u32 pos)
{
	u32 i = pos & RMASK;
	if (i <= RING - 4)
		return ld32u(ring + i);
	u32 v = 0;
	for (int b = 0; b < 4; b++)
		v |= (u32)ring[(pos + b) & RMASK] << (8 * b);
	return v;
}
static __device__ __forceinline__ void ring_st4(u8 *ring, u32 pos, u32 v)
{
	u32 i = pos & RMASK;
	if (i <= RING - 4) {
		st32u(ring + i, v);
		return;
	}
	for (int b = 0; b < 4; b++)
		ring[(pos + b) & RMASK] = (u8)(v >> (8 * b));
}

/* copy len bytes from a linear source to ring[dpos...]; exact except that 1..3 byte runs are
 * written as one dword when `slop` (the caller owns the 3 bytes that follow) */
static __device__ __forceinline__ void lin_to_ring(u8 *ring, u32 dpos, const u8 *s, u32 len, bool slop)
{
	if (len >= 8) {
		for (u32 i = 0; i + 8 < len; i += 8)
			ring_st8(ring, dpos + i, ld64u(s + i));
		ring_st8(ring, dpos + len - 8, ld64u(s + len - 8));
	} else if (len >= 4) {
		ring_st4(ring, dpos, ld32u(s));
		ring_st4(ring, dpos + len - 4, ld32u(s + len - 4));
	} else if (len) {
		if (slop) {
			ring_st4(ring, dpos, ld32u(s));
		} else {
			for (u32 i = 0; i < len; i++)
				ring[(dpos + i) & RMASK] = s[i];
		}
	}
}

/* LZ4 match inside the ring: ring[dpos..dpos+ml) = ring[dpos-off...], byte-serial semantics */
static __device__ __forceinline__ void ring_match(u8 *ring, u32 dpos, u32 off, u32 ml)
{
	const u32 spos = dpos - off;
	if (ml >= 4 && (off >= 8 || off >= ml)) {
		if (ml >= 8) {
			/* units run in order, so with off >= 8 a unit only reads bytes already final */
			for (u32 i = 0; i + 8 < ml; i += 8)
				ring_st8(ring, dpos + i, ring_ld8(ring, spos + i));
			ring_st8(ring, dpos + ml - 8, ring_ld8(ring, spos + ml - 8));
		} else {
			u32 a = ring_ld4(ring, spos), b = ring_ld4(ring, spos + ml - 4);
			ring_st4(ring, dpos, a);
			ring_st4(ring, dpos + ml - 4, b);
		}
	} else {
		/* short period (< 8) overlapping itself: replicate the first period */
		u32 j = 0;
		for (u32 i = 0; i < ml; i++) {
			ring[(dpos + i) & RMASK] = ring[(spos + j) & RMASK];
			if (++j == off)
				j = 0;
		}
	}
}

struct DecState {
	u32 opos;       /* next output position (chunk-absolute) */
	u32 flushed;    /* out[0..flushed) is in global memory */
	u32 valid_from; /* ring holds valid bytes for positions >= valid_from (and > opos - RING) */
};

/* ring -> global for [st.flushed, upto); exact */
static __device__ __forceinline__ void flush_to(DecState &st, const u8 *ring, u8 *out, u32 upto, int lane)
{
	u32 f = st.flushed;
	if (upto <= f)
		return;
	/* head: up to the next multiple of 16 */
	u32 head = (16 - (f & 15)) & 15;
	if (head > upto - f)
		head = upto - f;
	if ((u32)lane < head)
		out[f + lane] = ring[(f + lane) & RMASK];
	f += head;
	u32 body_end = f + ((upto - f) & ~15u);
	for (u32 pos = f + 16 * (u32)lane; pos < body_end; pos += 1024) {
		const u8 *r = ring + (pos & RMASK); /* 16-aligned, never wraps */
		u64 a = *(const u64 *)r, b = *(const u64 *)(r + 8);
		st64u(out + pos, a);
		st64u(out + pos + 8, b);
	}
	if ((u32)lane < upto - body_end)
		out[body_end + lane] = ring[(body_end + lane) & RMASK];
	st.flushed = upto;
}

/* ring -> global, full 16-byte units only; the partial tail stays pending */
static __device__ __forceinline__ void flush_aligned(DecState &st, const u8 *ring, u8 *out, u32 upto, int lane)
{
	u32 end = upto & ~15u;
	if (end > st.flushed)
		flush_to(st, ring, out, end, lane);
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
	st.valid_from = opos;
	return ip;
}

/* speculative "next token position" of staged position q (see file header) */
static __device__ __forceinline__ u32 spec_next(const u8 *cb, u32 q, u32 nst, bool block_ends)
{
	if (q >= nst)
		return NXT_FAR;
	u32 tok = cb[q], lit = tok >> 4, h = q + 1;
	if (lit == 15) {
		for (;;) {
			if (h >= nst)
				return NXT_FAR;
			u32 b = cb[h++];
			lit += b;
			if (b != 255)
				break;
			if (lit > CAP_LEN)
				return NXT_FAR;
		}
	}
	if (lit > CAP_LEN)
		return NXT_FAR;
	u32 lend = h + lit;
	if (lend >= nst)
		return (block_ends && lend == nst) ? NXT_LAST : NXT_FAR;
	if (lend + 2 > nst)
		return NXT_FAR;
	u32 ml = tok & 15, m = lend + 2;
	if (ml == 15) {
		for (;;) {
			if (m >= nst)
				return NXT_FAR;
			u32 b = cb[m++];
			ml += b;
			if (b != 255)
				break;
			if (ml > CAP_LEN)
				return NXT_FAR;
		}
	}
	if (ml + 4 > CAP_LEN)
		return NXT_FAR;
	return m;
}

/* Decode one LZ4 block; returns ST_OK or ST_BAD_BLOCK.  st.opos advances. */
static __device__ u32 decode_block_batch(const u8 *src, u32 slen, u8 *out, DecState &st, u32 low,
					 u32 limit, u8 *cb, u8 *ring, int lane)
{
	u32 ip = 0;
	if (slen == 0)
		return ST_BAD_BLOCK;
	for (;;) {
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

		/* ---------------- spec ---------------- */
		u32 nxtv[8];
		ZMT_UNROLL
		for (int j = 0; j < 8; j++)
			nxtv[j] = spec_next(cb, (u32)j * 64 + (u32)lane, nst, block_ends);

		/* ---------------- walk ---------------- */
		u32 p = 0, k = 0, posv = 0;
		bool stop = false;
		ZMT_UNROLL
		for (int j = 0; j < 8; j++) {
			while (!stop && k < 64 && (p >> 6) == (u32)j) {
				u32 n = wv_readlane(nxtv[j], (int)(p & 63));
				if (n >= NXT_LAST) {
					stop = true;
				} else {
					posv = wv_writelane(p, (int)k, posv);
					k++;
					p = n;
				}
			}
		}

		if (k == 0) {
			bool done;
			ip = one_sequence_global(src, slen, ip, out, st, ring, low, limit, &done, lane);
			if (ip == 0xFFFFFFFFu)
				return ST_BAD_BLOCK;
			if (done)
				return ST_OK;
			continue;
		}

		/* ---------------- decode own sequence ---------------- */
		const bool act0 = (u32)lane < k;
		u32 lit = 0, ml = 0, off = 1, lsrc = 0;
		if (act0) {
			u32 q = posv, tok = cb[q], h = q + 1;
			lit = tok >> 4;
			if (lit == 15) {
				u32 b;
				do {
					b = cb[h++];
					lit += b;
				} while (b == 255);
			}
			lsrc = h;
			u32 lend = h + lit;
			off = (u32)cb[lend] | (u32)cb[lend + 1] << 8;
			ml = tok & 15;
			if (ml == 15) {
				u32 m = lend + 2, b;
				do {
					b = cb[m++];
					ml += b;
				} while (b == 255);
			}
			ml += 4;
		}
		const u32 len = act0 ? lit + ml : 0;
		const u32 incl = wv_scan_incl(len);
		const u32 op = st.opos + incl - len; /* start of this sequence's literals */
		const u32 mpos = op + lit;           /* start of its match */
		const u32 src_pos = mpos - off;
		const u32 eff = ml < off ? ml : off; /* bytes of source actually read */

		/* validity: offsets inside the window, output inside the block */
		{
			bool bad = act0 && (off == 0 || off > mpos - low);
			u32 total = wv_readlane(incl, 63);
			if (wv_any(bad) || limit - st.opos < total)
				return ST_BAD_BLOCK;
		}

		/* classify sources: ring, global, or straddling the frontier (cuts the batch) */
		const u32 base_total = wv_readlane(incl, (int)(k - 1));
		u32 batch_end = st.opos + base_total;
		u32 ring_lo = batch_end > RING ? batch_end - RING : 0;
		if (st.valid_from > ring_lo)
			ring_lo = st.valid_from;
		bool is_far = act0 && src_pos < ring_lo;
		{
			bool straddle = is_far && src_pos + eff > st.flushed;
			u64 sm = wv_ballot(straddle);
			if (sm) {
				u32 cut = (u32)wv_ffs(sm) - 1;
				if (cut == 0) {
					bool done;
					ip = one_sequence_global(src, slen, ip, out, st, ring, low, limit,
								 &done, lane);
					if (ip == 0xFFFFFFFFu)
						return ST_BAD_BLOCK;
					if (done)
						return ST_OK;
					continue;
				}
				k = cut;
				p = wv_readlane(posv, (int)cut);
				batch_end = st.opos + wv_readlane(incl, (int)(cut - 1));
			}
		}
		const bool act = (u32)lane < k;
		is_far = is_far && act;

		/* ---------------- literals ---------------- */
		if (act)
			lin_to_ring(ring, op, cb + lsrc, lit, true);

		/* ---------------- far matches (after the literals: a 1..3 byte literal run is stored as
		 * one dword that spills into the lane's own match region) ---------------- */
		wave_mem_fence(); /* flushed bytes of earlier batches are visible */
		if (is_far && off < ml) {
			/* self-overlapping match whose first period lies in global memory: fetch the
			 * period, then replicate it inside the ring */
			lin_to_ring(ring, mpos, out + src_pos, off, false);
			ring_match(ring, mpos + off, off, ml - off);
		} else if (is_far) {
			const u8 *g = out + src_pos;
			if (ml >= 8) {
				for (u32 i = 0; i + 8 < ml; i += 8)
					ring_st8(ring, mpos + i, ld64u(g + i));
				ring_st8(ring, mpos + ml - 8, ld64u(g + ml - 8));
			} else {
				u32 a = ld32u(g), b = ld32u(g + ml - 4);
				ring_st4(ring, mpos, a);
				ring_st4(ring, mpos + ml - 4, b);
			}
		}
		wv_sync();

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
					ring_match(ring, mpos, off, ml);
					fin = true;
				}
				wv_sync();
			}
		}

		/* ---------------- flush + advance ---------------- */
		st.opos = batch_end;
		flush_aligned(st, ring, out, batch_end, lane);
		ip += p;
	}
}

extern "C" __global__ void __launch_bounds__(256)
zmt_lz4_dec_batch(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
		  const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
		  const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
		  u32 *__restrict__ status, u32 *__restrict__ chk_expect, u32 *__restrict__ chk_valid)
{
	__shared__ __attribute__((aligned(16))) u8 lds[4 * WAVE_LDS];
	const int lane = wv_lane();
	const u32 wave = threadIdx.x >> 6;
	const u32 rec = blockIdx.x * 4 + wave;
	if (rec >= nrec)
		return;
	u8 *cb = lds + wave * WAVE_LDS;
	u8 *ring = cb + CB_STAGE + CB_SLACK;
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
			st = decode_block_batch(r + ip, bsz, out, ds, fi.indep ? ds.opos : 0,
						ds.opos + room, cb, ring, lane);
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
}
