/*
 * lz4_enc.hip -- LZ4 frame encoder, one wave per chunk, bit-exact with the reference's output.
 *
 * Replaces, per chunk, LZ4F_compressFrame (call site /root/reference/lib/lz4-mt_compress.c:281,
 * prefs :141-146) and the skippable-header emit of lib/lz4-mt_compress.c:294-298.  The parse is
 * the LZ4 "fast" greedy parser at acceleration 1: its hash-table state is
 * sequential across the whole chunk (carried over the chunk's linked 64 KiB blocks), so the unit
 * of parallelism is the chunk -- 65 536 of them at BASELINE config 2.
 *
 * Layout: the 16 KiB hash table lives in LDS (one table per wave => up to 10 chunk-waves per CU);
 * the parse state is wave-uniform (kept in SGPRs via readfirstlane), the 64 lanes cooperate on
 * the byte-moving parts: match-length counting (64 B per ballot step) and literal copies
 * (coalesced).  HBM traffic = chunk read once + record written once; table traffic stays in LDS.
 */
#include "lz4_common.h"

#define MINMATCH 4u
#define MFLIMIT 12u
#define LASTLITERALS 5u
#define DIST_MAX 65535u

template <bool U16> static __device__ __forceinline__ u32 hash_at(const u8 *p)
{
	if (U16) {
		return (uld32(p) * 2654435761u) >> 19;
	} else {
		u64 v = (u64)uld32(p) | (u64)uld32(p + 4) << 32;
		return (u32)(((v << 24) * 889523592379ULL) >> 52);
	}
}
/*
 * The table is written by lane 0 only and read wave-uniformly.  wv_sync() orders lane 0's store
 * before every lane's later load and every lane's load before lane 0's next store; on hardware it
 * costs no instruction (a wave issues its DS operations in order).
 */
template <bool U16> static __device__ __forceinline__ u32 tab_get(const u32 *tab, u32 h)
{
	wv_sync();
	u32 v = U16 ? (u32)((const u16 *)tab)[h] : tab[h];
	wv_sync();
	return wv_readfirst(v);
}
template <bool U16> static __device__ __forceinline__ void tab_put(u32 *tab, u32 h, u32 v, int lane)
{
	if (lane == 0) {
		if (U16)
			((u16 *)tab)[h] = (u16)v;
		else
			tab[h] = v;
	}
}

/* number of equal bytes a[i] == b[i], i < limit (64 bytes per step) */
static __device__ __forceinline__ u32 wave_count(const u8 *a, const u8 *b, u32 limit, int lane)
{
	u32 base = 0;
	for (;;) {
		u32 i = base + (u32)lane;
		bool stop = true;
		if (i < limit)
			stop = a[i] != b[i];
		u64 m = wv_ballot(stop);
		if (m)
			return base + (u32)wv_ffs(m) - 1;
		base += 64;
	}
}

/* emit a length >= 15 continuation: (len-15) as 255-run bytes; returns bytes written */
static __device__ __forceinline__ u32 put_len_ext(u8 *op, u32 rem, int lane)
{
	u32 n255 = rem / 255;
	for (u32 i = (u32)lane; i < n255; i += 64)
		op[i] = 255;
	if (lane == 0)
		op[n255] = (u8)(rem - n255 * 255);
	return n255 + 1;
}

/*
 * One block.  chunk = start of the chunk, [pos, pos+len) the block.  Returns the compressed size
 * or 0 if it does not fit in cap (then stored raw by the caller; insertions made so far stay).
 */
template <bool U16>
static __device__ u32 encode_block(u32 *tab, const u8 *chunk, u32 pos, u32 len, u8 *dst, u32 cap,
				   int lane)
{
	const u32 iend = pos + len;
	const u32 mflimit_p1 = iend - MFLIMIT + 1;
	const u32 matchlimit = iend - LASTLITERALS;
	const u32 low = U16 ? pos : 0;
	u32 ip = pos, anchor = pos, op = 0;
	u32 fwd_h, match = 0, token = 0, tokhi = 0;

	if (len < MFLIMIT + 1)
		goto last_literals;

	tab_put<U16>(tab, hash_at<U16>(chunk + ip), ip, lane);
	ip++;
	fwd_h = hash_at<U16>(chunk + ip);

	for (;;) {
		{
			u32 fwd = ip, step = 1, nb = 1u << 6;
			for (;;) {
				u32 h = fwd_h, cur = fwd;
				u32 midx = tab_get<U16>(tab, h);
				ip = fwd;
				fwd += step;
				step = nb++ >> 6;
				if (fwd > mflimit_p1)
					goto last_literals;
				match = midx;
				fwd_h = hash_at<U16>(chunk + fwd);
				tab_put<U16>(tab, h, cur, lane);
				if (!U16 && midx + DIST_MAX < cur)
					continue;
				if (uld32(chunk + match) == uld32(chunk + ip))
					break;
			}
		}
		/* catch up */
		while (ip > anchor && match > low && uld8(chunk + ip - 1) == uld8(chunk + match - 1)) {
			ip--;
			match--;
		}
		{
			u32 lit = ip - anchor;
			token = op++;
			if (op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > cap)
				return 0;
			/* the token byte is written once, when the match length is known */
			tokhi = (lit >= 15 ? 15u : lit) << 4;
			if (lit >= 15)
				op += put_len_ext(dst + op, lit - 15, lane);
			wave_copy(dst + op, chunk + anchor, lit, lane);
			op += lit;
		}
		for (;;) { /* next_match */
			u32 mc;
			if (lane == 0)
				st16u(dst + op, ip - match);
			op += 2;
			mc = wave_count(chunk + ip + MINMATCH, chunk + match + MINMATCH,
					matchlimit - (ip + MINMATCH), lane);
			ip += mc + MINMATCH;
			if (op + (1 + LASTLITERALS) + (mc + 240) / 255 > cap)
				return 0;
			if (lane == 0)
				dst[token] = (u8)(tokhi | (mc >= 15 ? 15u : mc));
			if (mc >= 15)
				op += put_len_ext(dst + op, mc - 15, lane);
			anchor = ip;
			if (ip >= mflimit_p1)
				goto block_done;
			tab_put<U16>(tab, hash_at<U16>(chunk + ip - 2), ip - 2, lane);
			{
				u32 h = hash_at<U16>(chunk + ip);
				u32 midx = tab_get<U16>(tab, h);
				match = midx;
				tab_put<U16>(tab, h, ip, lane);
				if ((U16 || midx + DIST_MAX >= ip) &&
				    uld32(chunk + match) == uld32(chunk + ip)) {
					token = op++;
					tokhi = 0;
					continue;
				}
			}
			break;
		}
		fwd_h = hash_at<U16>(chunk + ++ip);
	}
block_done:
last_literals:
	{
		u32 run = iend - anchor;
		if (op + run + 1 + (run + 255 - 15) / 255 > cap)
			return 0;
		if (run >= 15) {
			if (lane == 0)
				dst[op] = 15 << 4;
			op++;
			op += put_len_ext(dst + op, run - 15, lane);
		} else {
			if (lane == 0)
				dst[op] = (u8)(run << 4);
			op++;
		}
		wave_copy(dst + op, chunk + anchor, run, lane);
		op += run;
	}
	return op;
}

extern "C" __global__ void __launch_bounds__(64)
zmt_lz4_enc_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nrec, u8 *__restrict__ slots,
		   u64 slot_stride, u32 *__restrict__ rec_len, const u32 *__restrict__ chk)
{
	__shared__ u32 tab[4096];
	const u32 rec = blockIdx.x;
	const int lane = wv_lane();
	if (rec >= nrec)
		return;

	const u64 start = (u64)rec * chunk;
	const u32 len = (u32)((n - start) < (u64)chunk ? (n - start) : (u64)chunk);
	const u8 *src = in + start;
	u8 *dst = slots + (u64)rec * slot_stride;
	const bool single = len <= ZMT_BLOCK;
	const u32 hdr = len ? 15 : 7;
	u32 op = 12 + hdr;

	if (lane == 0) {
		u8 d[10];
		st32u(dst, ZMT_SKIP_MAGIC);
		st32u(dst + 4, 4);
		st32u(dst + 12, ZMT_LZ4F_MAGIC);
		d[0] = (u8)(0x40 | (single ? 0x20 : 0) | (len ? 0x08 : 0) | 0x04);
		d[1] = 0x40;
		for (int i = 0; i < 8; i++)
			d[2 + i] = (i < 4) ? (u8)(len >> (8 * i)) : 0;
		for (u32 i = 0; i < hdr - 5; i++)
			dst[16 + i] = d[i];
		dst[12 + hdr - 1] = (u8)(xxh32_short(d, hdr - 5) >> 8);
	}
	for (u32 i = (u32)lane; i < 4096; i += 64)
		tab[i] = 0;
	wv_sync();

	for (u32 pos = 0; pos < len; pos += ZMT_BLOCK) {
		u32 blen = len - pos < ZMT_BLOCK ? len - pos : ZMT_BLOCK;
		u32 c = single ? encode_block<true>(tab, src, pos, blen, dst + op + 4, blen - 1, lane)
			       : encode_block<false>(tab, src, pos, blen, dst + op + 4, blen - 1, lane);
		u32 bh = c;
		if (c == 0) {
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
}
