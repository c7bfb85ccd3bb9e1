/*
 * zstd_dec_common.h -- what the zstd decode kernels share (zstd_dec.hip: one wave per frame; zstd_dec_seq.hip: the
 * sequence pre-pass): constants of RFC 8878, the LDS staging helper, FSE table descriptions and decoding cells, bit
 * extraction from a 128-bit window, Huffman weights and table fill.  Device code only; see zstd_dec.hip for the format notes.
 */
#pragma once
#include <cstddef>
#include "lz4_common.h"
#include "lz4_frame.h"
#include "match_copy.h"

#define ZMT_ZSTD_MAGIC 0xFD2FB528u
#define Z_BLOCK_MAX 131072u
#define Z_SEQCAP 24576u /* sequences decoded ahead of their blocks, 8 bytes each (see `unit` below) */
#define Z_LITSLOT (Z_BLOCK_MAX + 256u + 8u * Z_SEQCAP) /* scratch per record = GPUMT_ZSTD_DEC_SCRATCH */
static_assert(Z_LITSLOT == 327936u, "GPUMT_ZSTD_DEC_SCRATCH in include/gpumt.h");
#define Z_STAGE 1024u
#define Z_CAP 64u /* longer literal runs / matches are copied by the whole wave */

#define ST_NEEDS_GENERAL 101u /* internal: the record needs the full-size tables (second kernel) */

static __device__ __forceinline__ int hb32(u32 v) { return 31 - __builtin_clz(v); }

/* ------------------------------------------------------------------ staging */
/* stage[0..n) <- src[rel .. rel+n), n multiple of 16, n <= 1024.  Bytes below `floor_rel`
 * (relative to src) are stored as zeros; [lo_ok, hi_ok) is what may be read from memory at all. */
static __device__ __forceinline__ void stage_load(u8 *stage, const u8 *src, long rel, u32 n, long floor_rel,
						  const u8 *lo_ok, const u8 *hi_ok, int lane)
{
	const u32 o = 16u * (u32)lane;
	if (o < n) {
		const long r = rel + (long)o;
		const u8 *p = src + r;
		u64 a = 0, b = 0;
		if (r >= floor_rel && p >= lo_ok && p + 16 <= hi_ok) {
			a = ld64u(p);
			b = ld64u(p + 8);
		} else {
			for (int k = 0; k < 8; k++) {
				if (r + k >= floor_rel && p + k >= lo_ok && p + k < hi_ok)
					a |= (u64)p[k] << (8 * k);
				if (r + 8 + k >= floor_rel && p + 8 + k >= lo_ok && p + 8 + k < hi_ok)
					b |= (u64)p[8 + k] << (8 * k);
			}
		}
		*(u64 *)(stage + o) = a;
		*(u64 *)(stage + o + 8) = b;
	}
}

/* ------------------------------------------------------------------ FSE (one lane) */
/* forward bit reader over LDS bytes (table descriptions) */
struct FwdBits {
	const u8 *p;
	u32 len, bit;
};
static __device__ __forceinline__ u32 fwd_peek(const FwdBits &b, int n)
{
	const u32 byte = b.bit >> 3;
	u64 v = 0;
	for (u32 i = 0; i < 5; i++)
		if (byte + i < b.len)
			v |= (u64)b.p[byte + i] << (8 * i);
	return (u32)(v >> (b.bit & 7)) & ((1u << n) - 1);
}

/* RFC 8878 4.1.1 table description -> norm[]; returns bytes used or -1 */
static __device__ int fse_read_ncount(const u8 *p, u32 len, short *norm, int max_sym, int max_log,
				      int *nsym_out, int *log_out)
{
	FwdBits b = {p, len, 0};
	if (len < 1)
		return -1;
	const int log = (int)fwd_peek(b, 4) + 5;
	b.bit += 4;
	if (log > max_log)
		return -1;
	int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
	bool prev0 = false;
	for (int i = 0; i < max_sym; i++)
		norm[i] = 0;
	while (remaining > 1 && sym < max_sym) {
		if (prev0) {
			for (;;) {
				const int r = (int)fwd_peek(b, 2);
				b.bit += 2;
				sym += r;
				if (r != 3)
					break;
				if (b.bit > 8 * len)
					return -1;
			}
			if (sym >= max_sym)
				return -1;
			prev0 = false;
			continue;
		}
		const int max = (2 * threshold - 1) - remaining;
		const u32 v = fwd_peek(b, nbits);
		int count;
		if ((int)(v & (u32)(threshold - 1)) < max) {
			count = (int)(v & (u32)(threshold - 1));
			b.bit += (u32)(nbits - 1);
		} else {
			count = (int)(v & (u32)(2 * threshold - 1));
			if (count >= threshold)
				count -= max;
			b.bit += (u32)nbits;
		}
		count--;
		remaining -= count < 0 ? -count : count;
		norm[sym++] = (short)count;
		prev0 = (count == 0);
		while (remaining < threshold) {
			nbits--;
			threshold >>= 1;
		}
	}
	if (remaining != 1 || (b.bit + 7) / 8 > len)
		return -1;
	*nsym_out = sym;
	*log_out = log;
	return (int)((b.bit + 7) / 8);
}

/* decoding cell: sym (6) | nbits (4) << 6 | extra bits of the code (5) << 10 | next-state base << 16 */
#define ZC_SYM(c) ((c) & 63u)
#define ZC_NB(c) (((c) >> 6) & 15u)
#define ZC_AB(c) (((c) >> 10) & 31u)
#define ZC_BASE(c) ((c) >> 16)

/* spread + number the cells (serial, one lane per table); returns 0 or -1.
 * xb: per-code "value base | extra bits << 24" table (LL / ML), NULL for offset codes (extra bits =
 * the code itself) -- kind 3 = no extra bits at all (Huffman weights) */
static __device__ int fse_build(u32 *cell, const short *norm, int nsym, int log, u16 *next, const u32 *xb,
				int kind)
{
	const u32 size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
	u32 high = size - 1, pos = 0;
	for (int s = 0; s < nsym; s++) {
		if (norm[s] == -1) {
			cell[high--] = (u32)s;
			next[s] = 1;
		} else {
			next[s] = (u16)norm[s];
		}
	}
	for (int s = 0; s < nsym; s++) {
		const int c = norm[s];
		for (int i = 0; i < c; i++) {
			cell[pos] = (u32)s;
			do
				pos = (pos + step) & mask;
			while (pos > high);
		}
	}
	if (pos != 0)
		return -1;
	for (u32 u = 0; u < size; u++) {
		const u32 s = cell[u];
		const u32 x = next[s]++;
		const u32 nb = (u32)(log - hb32(x));
		const u32 ab = kind == 3 ? 0 : xb ? xb[s] >> 24 : s;
		cell[u] = s | nb << 6 | ab << 10 | (((x << nb) - size) & 0xFFFFu) << 16;
	}
	return 0;
}

/* `width` (<= 32) bits that start `topoff` bits below the top of the 128-bit window w0:w1 */
/* 16 LDS bytes at any address as w1 (bytes 0..7) and w0 (bytes 8..15) from five ALIGNED dword reads + funnel shifts: a
 * misaligned 8-byte LDS read costs the LDS pipe one cycle per active lane (tools/ubench/lds_cost.hip), and the unit
 * decoder has 48 of them active */
static __device__ __forceinline__ void lds_ld128(const u8 *p, u64 &w1, u64 &w0)
{
#ifdef ZMT_EMU
	w1 = ld64u(p);
	w0 = ld64u(p + 8);
#else
	const u32 a = (u32)(size_t)(const __attribute__((address_space(3))) u8 *)p;
	const __attribute__((address_space(3))) u32 *d = (const __attribute__((address_space(3))) u32 *)(size_t)(a & ~3u);
	const u32 d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
	w1 = (u64)wv_alignbyte(d1, d0, a) | (u64)wv_alignbyte(d2, d1, a) << 32;
	w0 = (u64)wv_alignbyte(d3, d2, a) | (u64)wv_alignbyte(d4, d3, a) << 32;
#endif
}

static __device__ __forceinline__ u32 xbits(u64 w0, u64 w1, u32 topoff, u32 width)
{
	const u32 sh = 128u - topoff - width;
	const u64 hi = w0 >> ((sh - 64u) & 63u);
	const u64 lo = (w1 >> (sh & 63u)) | ((w0 << 1) << (63u - (sh & 63u)));
	const u64 v = sh >= 64u ? hi : lo;
	return (u32)(v & ((1ull << width) - 1));
}

/* ------------------------------------------------------------------ constants */
__device__ static const short Z_LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
					       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ static const short Z_OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
					       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__device__ static const short Z_ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
					       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
					       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__device__ static const u32 Z_LL_BASE[36] = {0,  1,  2,   3,   4,   5,    6,    7,    8,    9,     10,    11,
					      12, 13, 14,  15,  16,  18,   20,   22,   24,   28,    32,    40,
					      48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ static const u8 Z_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,  0,  0,  1,  1,
					     1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ static const u32 Z_ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16,
					      17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30,
					      31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83,
					      99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ static const u8 Z_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,  0,  0,
					     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  1,  1,  1,  1,
					     2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* ------------------------------------------------------------------ Huffman */
/* Tree description at d[0..len) (LDS) -> lds.w[0..nw] incl. the implied last weight; sets *log.
 * One lane.  Returns bytes consumed or -1.  `cells` is 64 words of scratch. */
static __device__ int huf_read_weights(const u8 *d, u32 len, u8 *w, int *nw_out, int *log_out, u32 *cells,
				       short *norm, u16 *next)
{
	int nw = 0, used;
	if (len < 1)
		return -1;
	const int hb = d[0];
	if (hb >= 128) {
		nw = hb - 127;
		const int bytes = (nw + 1) / 2;
		if ((u32)(1 + bytes) > len)
			return -1;
		for (int i = 0; i < nw; i++)
			w[i] = (i & 1) ? (d[1 + i / 2] & 15) : (d[1 + i / 2] >> 4);
		used = 1 + bytes;
	} else {
		if ((u32)(1 + hb) > len || hb < 1)
			return -1;
		int nsym, log;
		const int u = fse_read_ncount(d + 1, (u32)hb, norm, 13, 6, &nsym, &log);
		if (u < 0 || u >= hb || fse_build(cells, norm, nsym, log, next, nullptr, 3))
			return -1;
		/* backward bitstream, two interleaved states (RFC 8878 4.2.1.2) */
		const u8 *bs = d + 1 + u;
		const int blen = hb - u;
		if (bs[blen - 1] == 0)
			return -1;
		int pos = 8 * (blen - 1) + hb32(bs[blen - 1]);
#define WBITS(n, out)                                                                              \
	do {                                                                                       \
		pos -= (n);                                                                        \
		u32 v_ = 0;                                                                        \
		for (int k_ = 0; k_ < (n); k_++) {                                                 \
			const int bp_ = pos + k_;                                                  \
			if (bp_ >= 0)                                                              \
				v_ |= (u32)((bs[bp_ >> 3] >> (bp_ & 7)) & 1) << k_;                \
		}                                                                                  \
		(out) = v_;                                                                        \
	} while (0)
		u32 s1, s2, t;
		WBITS(log, s1);
		WBITS(log, s2);
		if (pos < 0)
			return -1;
		for (;;) {
			if (nw > 253)
				return -1;
			u32 c = cells[s1];
			w[nw++] = (u8)ZC_SYM(c);
			WBITS((int)ZC_NB(c), t);
			s1 = ZC_BASE(c) + t;
			if (pos < 0) {
				w[nw++] = (u8)ZC_SYM(cells[s2]);
				break;
			}
			if (nw > 253)
				return -1;
			c = cells[s2];
			w[nw++] = (u8)ZC_SYM(c);
			WBITS((int)ZC_NB(c), t);
			s2 = ZC_BASE(c) + t;
			if (pos < 0) {
				w[nw++] = (u8)ZC_SYM(cells[s1]);
				break;
			}
		}
#undef WBITS
		used = 1 + hb;
	}
	u32 total = 0;
	for (int i = 0; i < nw; i++) {
		if (w[i] > 11)
			return -1;
		total += w[i] ? (1u << (w[i] - 1)) : 0;
	}
	if (total == 0)
		return -1;
	const int log = hb32(total) + 1;
	if (log > 11)
		return -1;
	const u32 rest = (1u << log) - total;
	if (rest & (rest - 1))
		return -1;
	w[nw++] = (u8)(hb32(rest) + 1);
	u32 c1 = 0;
	for (int i = 0; i < nw; i++)
		c1 += (w[i] == 1);
	if (c1 < 2 || (c1 & 1))
		return -1;
	*nw_out = nw;
	*log_out = log;
	return used;
}

/* wave: weights -> decoding table.  Symbols are laid out by ascending weight, then symbol value;
 * a symbol of weight r owns 2^(r-1) consecutive cells. */
static __device__ void huf_fill(u16 *huf, const u8 *w, int nw, int log, int lane)
{
	/* start offset of every rank */
	u32 start = 0;
	for (int r = 1; r <= log; r++) {
		u32 cnt = 0;
		for (int g = 0; g < 256; g += 64) {
			const int s = g + lane;
			cnt += (u32)wv_popc(wv_ballot(s < nw && w[s] == r));
		}
		/* symbols of this rank, in symbol order */
		u32 at = start;
		for (int g = 0; g < 256 && g < nw; g += 64) {
			const int s = g + lane;
			const bool mine = s < nw && w[s] == r;
			const u64 m = wv_ballot(mine);
			const u32 n = 1u << (r - 1);
			if (mine) {
				const u32 base = at + wv_mbcnt(m) * n;
				const u16 e = (u16)((u32)s | (u32)(log + 1 - r) << 8);
				for (u32 i = 0; i < n; i++)
					huf[base + i] = e;
			}
			at += (u32)wv_popc(m) * n;
		}
		start += cnt << (r - 1);
	}
}
