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


/* ============================================================================================
 * Encoder v2: the search loop evaluates 64 consecutive probes of the reference's probe sequence
 * at once (lane j = probe 64*batch + j).  Exactly the reference's decisions are reproduced:
 *   - probe k sits at ip + (k <= 65 ? k : 65 + 64*q(q+3)/2 + (q+2)*r), q = (k-65)>>6, r = (k-65)&63
 *     (steps 1 for the first 65 gaps, then 2, 3, ... per 64 probes: searchMatchNb++ >> skipTrigger);
 *   - a probe is processed only if its successor position does not pass mflimit+1;
 *   - probe j's candidate is the table entry as left by all earlier probes: the old table value,
 *     unless an earlier lane of the same batch has the same hash (detected exactly with an LDS
 *     atomic-or bitmap, resolved by a 64-step readlane loop only when it happens);
 *   - the first matching lane wins; only lanes up to it insert, the latest duplicate last.
 * One candidate gather (global) per batch instead of two dependent loads per probe.
 * ============================================================================================ */
#ifdef ZMT_EMU
static inline u32 lds_atomic_or(u32 *p, u32 v)
{
	u32 o = *p;
	*p = o | v;
	return o;
}
#else
static __device__ __forceinline__ u32 lds_atomic_or(u32 *p, u32 v) { return atomicOr(p, v); }
#endif

template <bool U16> static __device__ __forceinline__ u32 hash_val(u64 x)
{
	if (U16)
		return ((u32)x * 2654435761u) >> 19;
	return (u32)(((x << 24) * 889523592379ULL) >> 52);
}

/* probe k of a search that starts at ip: position and the gap to probe k+1 */
static __device__ __forceinline__ u32 probe_pos(u32 ip, u32 k, u32 *gap)
{
	if (k <= 64) {
		*gap = 1;
		return ip + k;
	}
	const u32 t = k - 65, q = t >> 6, r = t & 63;
	*gap = q + 2;
	return ip + 65 + 64 * (q * (q + 3) / 2) + (q + 2) * r;
}

template <bool U16>
static __device__ u32 encode_block2(u32 *tab, u32 *bitmap, const u8 *chunk, u32 pos, u32 len, u8 *dst,
				    u32 cap, int lane)
{
	const u32 iend = pos + len;
	const u32 mflimit_p1 = iend - MFLIMIT + 1;
	const u32 matchlimit = iend - LASTLITERALS;
	const u32 low = U16 ? pos : 0;
	u32 ip = pos, anchor = pos, op = 0;
	u32 match = 0, token = 0, tokhi = 0;

	if (len < MFLIMIT + 1)
		goto last_literals;

	tab_put<U16>(tab, hash_at<U16>(chunk + ip), ip, lane);
	ip++;

	for (;;) {
		/* ---------------- search: batches of 64 probes ---------------- */
		{
			const u32 ip0 = ip;
			bool found = false;
			for (u32 batch = 0;; batch++) {
				u32 gap;
				const u32 cur = probe_pos(ip0, batch * 64 + (u32)lane, &gap);
				const bool valid = cur + gap <= mflimit_p1 && cur < mflimit_p1;
				const u64 vm = wv_ballot(valid);
				if (vm == 0)
					goto last_literals; /* the very first probe of the batch already ends it */
				const u64 x = valid ? ld64u(chunk + cur) : 0;
				const u32 h = hash_val<U16>(U16 ? (u64)(u32)x : x);
				wv_sync();
				u32 cand = valid ? (U16 ? (u32)((const u16 *)tab)[h] : tab[h]) : 0;
				/* exact in-batch duplicate detection */
				u32 prev_dup = 64, next_dup = 64; /* lane indices, 64 = none */
				{
					bool d = false;
					if (valid)
						d = (lds_atomic_or(&bitmap[h >> 5], 1u << (h & 31)) >> (h & 31)) & 1;
					const bool any_dup = wv_any(d);
					wv_sync();
					if (valid)
						bitmap[h >> 5] = 0;
					if (any_dup) {
						for (int i = 0; i < 64; i++) {
							const u32 hi = wv_readlane(h, i);
							const bool vi = (vm >> i) & 1;
							if (vi && valid && hi == h) {
								if (i < lane)
									prev_dup = (u32)i;
								else if (i > lane && next_dup == 64)
									next_dup = (u32)i;
							}
						}
					}
				}
				u32 c4;
				{
					/* candidate bytes: an earlier probe of this batch, or memory */
					const u32 pc = wv_shfl(cur, (int)(prev_dup & 63));
					const u32 p4 = wv_shfl((u32)x, (int)(prev_dup & 63));
					if (prev_dup < 64)
						cand = pc;
					const bool dist_ok = U16 || cand + DIST_MAX >= cur;
					c4 = (valid && dist_ok && prev_dup == 64) ? ld32u(chunk + cand) : p4;
					const bool m = valid && dist_ok && c4 == (u32)x;
					const u64 mm = wv_ballot(m);
					const u32 nvalid = (u32)wv_popc(vm);
					const u32 jstar = mm ? (u32)wv_ffs(mm) - 1 : 64;
					const u32 ninsert = mm ? jstar + 1 : nvalid;
					/* inserts: probes 0..ninsert-1, the latest duplicate wins */
					if ((u32)lane < ninsert && !(next_dup < ninsert)) {
						if (U16)
							((u16 *)tab)[h] = (u16)cur;
						else
							tab[h] = cur;
					}
					wv_sync();
					if (mm) {
						ip = wv_readlane(cur, (int)jstar);
						match = wv_readlane(cand, (int)jstar);
						found = true;
						break;
					}
					if (nvalid < 64)
						goto last_literals; /* ran into the end of the block without a match */
				}
			}
			if (!found)
				goto last_literals;
		}
		/* ---------------- catch up (cooperative backward compare) ---------------- */
		for (;;) {
			u32 room = ip - anchor;
			if (match - low < room)
				room = match - low;
			if (room == 0)
				break;
			const u32 n = room < 64 ? room : 64;
			bool eq = false;
			if ((u32)lane < n)
				eq = chunk[ip - 1 - (u32)lane] == chunk[match - 1 - (u32)lane];
			const u64 ne = ~wv_ballot(eq);
			const u32 run = ne ? (u32)wv_ffs(ne) - 1 : 64; /* leading equal bytes */
			const u32 take = run < n ? run : n;
			ip -= take;
			match -= take;
			if (take < 64)
				break;
		}
		{
			u32 lit = ip - anchor;
			token = op++;
			if (op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > cap)
				return 0;
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
			/* T[h(ip-2)] = ip-2, then the immediate re-match test at ip: both hashes come
			 * from one 16-byte load at ip-2 */
			{
				u64 a = 0, b = 0;
				a = wv_readfirst((u32)ld32u(chunk + ip - 2)) | (u64)wv_readfirst(ld32u(chunk + ip + 2)) << 32;
				b = wv_readfirst((u32)ld32u(chunk + ip + 6));
				const u64 x2 = a;                       /* bytes ip-2 .. ip+5 */
				const u64 x0 = (a >> 16) | (b << 48);   /* bytes ip .. ip+7   */
				tab_put<U16>(tab, hash_val<U16>(U16 ? (u64)(u32)x2 : x2), ip - 2, lane);
				const u32 h = hash_val<U16>(U16 ? (u64)(u32)x0 : x0);
				const u32 midx = tab_get<U16>(tab, h);
				match = midx;
				tab_put<U16>(tab, h, ip, lane);
				if ((U16 || midx + DIST_MAX >= ip) && uld32(chunk + match) == (u32)x0) {
					token = op++;
					tokhi = 0;
					continue;
				}
			}
			break;
		}
		ip++;
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

template <bool V2>
static __device__ __forceinline__ void enc_body(u32 *tab, u32 *bitmap, const u8 *__restrict__ in, u64 n, u32 chunk, u32 nrec, u8 *__restrict__ slots,
		   u64 slot_stride, u32 *__restrict__ rec_len, const u32 *__restrict__ chk)
{
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
	for (u32 i = (u32)lane; i < 256; i += 64)
		bitmap[i] = 0;
	wv_sync();

	for (u32 pos = 0; pos < len; pos += ZMT_BLOCK) {
		u32 blen = len - pos < ZMT_BLOCK ? len - pos : ZMT_BLOCK;
		u32 c;
		if (V2)
			c = single ? encode_block2<true>(tab, bitmap, src, pos, blen, dst + op + 4, blen - 1, lane)
				   : encode_block2<false>(tab, bitmap, src, pos, blen, dst + op + 4, blen - 1, lane);
		else
			c = single ? encode_block<true>(tab, src, pos, blen, dst + op + 4, blen - 1, lane)
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

extern "C" __global__ void __launch_bounds__(64)
zmt_lz4_enc_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nrec, u8 *__restrict__ slots,
		   u64 slot_stride, u32 *__restrict__ rec_len, const u32 *__restrict__ chk)
{
	__shared__ u32 tab[4096];
	__shared__ u32 bitmap[256];
	enc_body<true>(tab, bitmap, in, n, chunk, nrec, slots, slot_stride, rec_len, chk);
}

/* v1: one probe at a time (kept for A/B measurements: gpumt_set_variant("lz4_enc", 1)) */
extern "C" __global__ void __launch_bounds__(64)
zmt_lz4_enc_v1_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nrec, u8 *__restrict__ slots,
		      u64 slot_stride, u32 *__restrict__ rec_len, const u32 *__restrict__ chk)
{
	__shared__ u32 tab[4096];
	__shared__ u32 bitmap[256];
	enc_body<false>(tab, bitmap, in, n, chunk, nrec, slots, slot_stride, rec_len, chk);
}
