/*
 * brotli_dec_common.h -- what the two brotli decode kernels share (brotli_dec.hip: one record per wave, every stream
 * RFC 7932 allows; brotli_dec4.hip: four records per wave, the streams without context modelling and block
 * switching): LDS layout, the wave-cooperative bit reader, prefix-code reading and canonical-table building, context
 * maps, dictionary words.  Replaces BrotliDecoderDecompress as called per record by the reference
 * (/root/reference/lib/brotli-mt_decompress.c:344-346).
 */
#ifndef ZMT_BROTLI_DEC_COMMON_H
#define ZMT_BROTLI_DEC_COMMON_H
#include "lz4_common.h"
#include "lz4_frame.h"
#define ZMT_GCOPY_ONE_TRIP /* the copies of a batch in one memory round trip (match_copy.h) */
#include "match_copy.h"

#define BR_NLIT_LDS 8u
#define BR_LIT_STRIDE 384u   /* 16 x u64 vector + 256 x u8 sorted symbols */
#define BR_CMD_STRIDE 1536u  /* vector + 704 x u16 */
#define BR_DIST_STRIDE 1216u /* vector + 544 x u16 */
#define BR_BT_STRIDE 704u    /* block type (258) / block count (26) / context map (272) codes */
#define BR_NCMD_LDS 1u
#define BR_NDIST_LDS 1u
#define BR_CAP 64u /* longer copies are moved by the whole wave */
/* uniform branches are what a single wave pays most for: keep the common path falling through */
#define BR_RARE(c) __builtin_expect(!!(c), 0)
#define BR_OFTEN(c) __builtin_expect(!!(c), 1)

/* scratch of one wave in HBM (include/gpumt.h GPUMT_BROTLI_SCRATCH) */
#define BR_G_LIT 0u
#define BR_G_CMD (BR_G_LIT + 256u * BR_LIT_STRIDE)
#define BR_G_DIST (BR_G_CMD + 256u * BR_CMD_STRIDE)
#define BR_G_BT (BR_G_DIST + 256u * BR_DIST_STRIDE)
#define BR_G_LCMAP (BR_G_BT + 8u * BR_BT_STRIDE)
#define BR_G_DCMAP (BR_G_LCMAP + 64u * 256u)
#define BR_G_END (BR_G_DCMAP + 4u * 256u)
#define BR_WSCRATCH ((BR_G_END + 255u) & ~255u)
static_assert(BR_WSCRATCH == 825856u, "keep GPUMT_BROTLI_SCRATCH (include/gpumt.h) in step");

#ifdef ZMT_EMU
#define ZMT_NOUNROLL
#define ZMT_NOINLINE inline
#define BR_SETTLE(v) ((void)(v))
#else
#define ZMT_NOINLINE __forceinline__
#define ZMT_NOUNROLL _Pragma("nounroll")
/* make a value loaded from HBM arrive inside the (rare) branch that loaded it: left pending, every
 * later use on the common path would wait for ALL memory operations in flight, stores included */
#define BR_SETTLE(v) asm volatile("" ::"v"(v))
#endif

struct BrLds {
	u8 lut[2048];
	__attribute__((aligned(8))) u8 lit[BR_NLIT_LDS * BR_LIT_STRIDE];
	__attribute__((aligned(8))) u8 cmd[BR_NCMD_LDS * BR_CMD_STRIDE];
	__attribute__((aligned(8))) u8 dist[BR_NDIST_LDS * BR_DIST_STRIDE];
	__attribute__((aligned(8))) u8 clrec[128 + 32];
	u8 lens[704];
	u8 cmap_cur[64];
	u8 cmode[256];
	u8 mtf[256 + 8];
	u8 tmp[64];
	u32 kins[24], kcopy[24]; /* insert / copy length codes: base | extra bits << 24 */
};

#ifdef ZMT_EMU
static inline u32 brbad_(int line)
{
	if (getenv("ZMT_EMU_DEBUG") && wv_lane() == 0)
		fprintf(stderr, "brotli_dec: malformed stream flagged at line %d\n", line);
	return ST_BAD_BLOCK;
}
#define BRBAD() brbad_(__LINE__)
static inline u32 br_rev15(u32 v)
{
	u32 r = 0;
	for (int i = 0; i < 15; i++)
		r |= ((v >> i) & 1) << (14 - i);
	return r;
}
#else
#define BRBAD() ST_BAD_BLOCK
static __device__ __forceinline__ u32 br_rev15(u32 v) { return __brev(v) >> 17; }
#endif

/* ------------------------------------------------------------------ bit reader */
struct BrBits {
	const u8 *p;
	u32 n;
	u32 win; /* lane i: stream bytes [wbyte + 4 i, +4) */
	u32 wbyte, widx;
	u64 acc;
	u32 navail;
};

static __device__ __forceinline__ u32 br_load_win(const u8 *p, u32 n, u32 wbyte, int lane)
{
	const u64 off = (u64)wbyte + 4u * (u32)lane;
	u32 v = 0;
	if (off + 4 <= n) {
		v = ld32u(p + off);
	} else {
		for (u32 k = 0; k < 4; k++)
			if (off + k < n)
				v |= (u32)p[off + k] << (8 * k);
	}
	return v;
}

static __device__ __forceinline__ void br_seek(BrBits &b, u32 byte, int lane)
{
	b.wbyte = byte;
	b.win = br_load_win(b.p, b.n, byte, lane);
	BR_SETTLE(b.win);
	b.widx = 0;
	b.acc = 0;
	b.navail = 0;
}

static __device__ __forceinline__ void br_refill(BrBits &b, int lane)
{
	if (b.navail <= 32) {
		if (BR_RARE(b.widx == 64)) {
			/* no prefetch of the next window: a register with a load in flight cannot be
			 * carried through the branches of the decode loop (every copy of it would wait
			 * for all memory operations, stores included); one wait per 256 stream bytes */
			b.wbyte += 256u;
			b.win = br_load_win(b.p, b.n, b.wbyte, lane);
			BR_SETTLE(b.win);
			b.widx = 0;
		}
		const u32 d = wv_readlane(b.win, (int)b.widx);
		b.acc |= (u64)d << b.navail;
		b.navail += 32;
		b.widx++;
	}
}

/* bits consumed since the start of the stream: everything fetched into the accumulator minus what
 * is still in it */
static __device__ __forceinline__ u64 br_used(const BrBits &b)
{
	return 8ull * b.wbyte + 32ull * b.widx - b.navail;
}
/* the stream ended before the bits consumed so far (windows past the end read as zeros) */
static __device__ __forceinline__ bool br_over(const BrBits &b)
{
	return br_used(b) > 8ull * b.n;
}

/* The header parsers take and return the reader by value and pin its wave-uniform fields to SGPRs
 * afterwards: should the compiler ever keep one of them out of line, the reader must not end up
 * living in memory (every value derived from it would count as lane-varying). */
static __device__ __forceinline__ BrBits br_pin(const BrBits &r)
{
	BrBits b;
	b.p = r.p;
	b.n = wv_readfirst(r.n);
	b.win = r.win;
	b.wbyte = wv_readfirst(r.wbyte);
	b.widx = wv_readfirst(r.widx);
	b.acc = (u64)wv_readfirst((u32)r.acc) | (u64)wv_readfirst((u32)(r.acc >> 32)) << 32;
	b.navail = wv_readfirst(r.navail);
	return b;
}

/* n <= 24 */
static __device__ __forceinline__ u32 br_get(BrBits &b, u32 n, int lane)
{
	br_refill(b, lane);
	const u32 v = (u32)b.acc & ((1u << n) - 1u);
	b.acc >>= n;
	b.navail -= n;
	return v;
}

__device__ static const u8 BR_CL_ORDER[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
__device__ static const u16 BR_BLEN_BASE[26] = {1, 5, 9, 13, 17, 25, 33, 41, 49, 65, 81, 97, 113, 145, 177, 209, 241,
						305, 369, 497, 753, 1265, 2289, 4337, 8433, 16625};
__device__ static const u8 BR_BLEN_BITS[26] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5,
					       5, 5, 5, 6, 6, 7, 8, 9, 10, 11, 12, 13, 24};
__device__ static const u16 BR_INS_BASE[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26,
					       34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
__device__ static const u8 BR_INS_BITS[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
__device__ static const u16 BR_COPY_BASE[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18,
						22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
__device__ static const u8 BR_COPY_BITS[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};
__device__ static const u8 BR_DICT_BITS[25] = {0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10,
					       9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5};
__device__ static const u32 BR_DICT_OFFS[25] = {0, 0, 0, 0, 0, 4096, 9216, 21504, 35840, 44032, 53248, 63488, 74752,
						87040, 93696, 100864, 104704, 106752, 108928, 113536, 115968, 118528,
						119872, 121280, 122016};

/* wave-uniform reads of the constant tables */
static __device__ __forceinline__ u32 ut8(const u8 *t, u32 i) { return wv_readfirst((u32)t[i]); }
static __device__ __forceinline__ u32 ut16(const u16 *t, u32 i) { return wv_readfirst((u32)t[i]); }

/* ------------------------------------------------------------------ prefix codes */
/* A tree record: 16 x u64 vector (lane l: low word = hi | lo << 16, high word = first index of
 * length l in the sorted array), then the symbols sorted by (length, value).  hi / lo are the
 * bounds of the code range of length l, left-aligned to 15 bits; lane 0 is the zero-bit code of a
 * one-symbol tree (hi = 32768). */
struct BrTree {
	u32 a, i;      /* this lane's vector entry (lanes >= 16: 0) */
	const u8 *sym; /* sorted symbols */
};

static __device__ __forceinline__ BrTree br_tree_load(const u8 *rec, int lane)
{
	BrTree t;
	u64 e = 0;
	if (lane < 16)
		e = *(const u64 *)(rec + 8 * lane);
	t.a = (u32)e;
	t.i = (u32)(e >> 32);
	t.sym = rec + 128;
	return t;
}

/* index of the next symbol in the tree's sorted array; consumes its bits.  `grp` selects the
 * 16-lane group the vector sits in (distance trees: one group per context). */
static __device__ __forceinline__ u32 br_sym_index(BrBits &b, u32 va, u32 vi, u32 grp, bool &bad, int lane)
{
	br_refill(b, lane);
	const u32 c = br_rev15((u32)b.acc);
	u64 m = wv_ballot(c < (va & 0xFFFFu));
	m = (m >> (16 * grp)) & 0xFFFFull;
	if (BR_RARE(!m)) {
		bad = true;
		return 0;
	}
	const int l = wv_ffs(m) - 1;
	const int sel = (int)(16 * grp) + l;
	const u32 a = wv_readlane(va, sel), i0 = wv_readlane(vi, sel);
	b.acc >>= l;
	b.navail -= (u32)l;
	return i0 + ((c - (a >> 16)) >> (15 - l));
}

static __device__ __forceinline__ u32 br_sym8(BrBits &b, const BrTree &t, bool &bad, int lane)
{
	const u32 k = br_sym_index(b, t.a, t.i, 0, bad, lane);
	return wv_readfirst((u32)t.sym[k]);
}
static __device__ __forceinline__ u32 br_sym16(BrBits &b, const BrTree &t, bool &bad, int lane)
{
	const u32 k = br_sym_index(b, t.a, t.i, 0, bad, lane);
	return wv_readfirst((u32) * (const u16 *)(t.sym + 2 * k));
}

/* lens[0, A) (LDS) -> tree record at rec.  The caller has checked that the code is complete or
 * has exactly one symbol. */
static __device__ ZMT_NOINLINE void br_build(u8 *rec, const u8 *lens, u32 A, bool sym16, int lane)
{
	u32 cnt = 0, one = 0;
	ZMT_NOUNROLL
	for (u32 s0 = 0; s0 < A; s0 += 64) {
		const u32 s = s0 + (u32)lane;
		const u32 ln = s < A ? lens[s] : 0;
		ZMT_NOUNROLL
		for (u32 l = 1; l <= 15; l++) {
			const u64 m = wv_ballot(ln == l);
			if ((u32)lane == l)
				cnt += (u32)wv_popc(m);
		}
		const u64 any = wv_ballot(ln != 0);
		if (any)
			one = s0 + (u32)wv_ffs(any) - 1;
	}
	u32 used = cnt;
	for (int d = 32; d; d >>= 1)
		used += wv_shfl(used, lane ^ d);
	u32 my_a = 0, my_i = 0;
	if (used == 1) {
		if (lane == 0)
			my_a = 32768u;
		if (lane == 0) {
			if (sym16)
				*(u16 *)(rec + 128) = (u16)one;
			else
				rec[128] = (u8)one;
		}
	} else {
		u32 code = 0, index = 0;
		ZMT_NOUNROLL
		for (u32 l = 1; l <= 15; l++) {
			const u32 c = wv_readlane(cnt, (int)l);
			const u32 lo = code << (15 - l), hi = (code + c) << (15 - l);
			if ((u32)lane == l) {
				my_a = hi | lo << 16;
				my_i = index;
			}
			if (c) {
				u32 run = index;
				ZMT_NOUNROLL
				for (u32 s0 = 0; s0 < A; s0 += 64) {
					const u32 s = s0 + (u32)lane;
					const bool mine = s < A && lens[s] == l;
					const u64 m = wv_ballot(mine);
					if (mine) {
						const u32 at = run + wv_mbcnt(m);
						if (sym16)
							*(u16 *)(rec + 128 + 2 * at) = (u16)s;
						else
							rec[128 + at] = (u8)s;
					}
					run += (u32)wv_popc(m);
				}
			}
			index += c;
			code = (code + c) << 1;
		}
	}
	if (lane < 16)
		*(u64 *)(rec + 8 * lane) = (u64)my_a | (u64)my_i << 32;
	wv_sync();
	wave_mem_fence();
}

/* RFC 7932 3.4 / 3.5: read one prefix code over `A` symbols and build its record at rec */
/* (LT: an LDS layout with lens[704], tmp[64] and clrec[160] -- BrLds, or the leaner one of brotli_dec4.hip) */
template <typename LT>
static __device__ ZMT_NOINLINE BrBits br_read_code_core(BrBits b, LT &L, u8 *rec, u32 A, bool sym16, int lane, u32 *ok)
{
	*ok = 0;
	for (u32 i = (u32)lane; i < 704; i += 64)
		L.lens[i] = 0;
	wv_sync();
	const u32 hskip = br_get(b, 2, lane);
	if (hskip == 1) {
		u32 max_bits = 0;
		for (u32 a = A - 1; a; a >>= 1)
			max_bits++;
		const u32 nsym = br_get(b, 2, lane) + 1;
		u32 s[4] = {0, 0, 0, 0};
		for (u32 i = 0; i < nsym; i++) {
			s[i] = br_get(b, max_bits, lane);
			if (s[i] >= A)
				return b;
		}
		for (u32 i = 1; i < nsym; i++)
			for (u32 j = 0; j < i; j++)
				if (s[i] == s[j])
					return b;
		u32 l0 = 1, l1 = 1, l2 = 2, l3 = 0;
		if (nsym == 3) {
			l1 = 2;
		} else if (nsym == 4) {
			if (br_get(b, 1, lane)) {
				l1 = 2;
				l2 = l3 = 3;
			} else {
				l0 = l1 = l3 = 2;
			}
		}
		if (lane == 0) {
			L.lens[s[0]] = (u8)l0;
			if (nsym > 1)
				L.lens[s[1]] = (u8)l1;
			if (nsym > 2)
				L.lens[s[2]] = (u8)l2;
			if (nsym > 3)
				L.lens[s[3]] = (u8)l3;
		}
		wv_sync();
		br_build(rec, L.lens, A, sym16, lane);
		*ok = 1;
		return b;
	}
	/* complex: the code-length code first (fixed variable-length code, values 0..5) */
	{
		u32 cl_len = 0; /* lane i: length of code-length symbol i */
		int space = 32;
		u32 ncodes = 0;
		ZMT_NOUNROLL
		for (u32 i = hskip; i < 18; i++) {
			br_refill(b, lane);
			const u32 pk = (u32)b.acc & 15u;
			u32 v, nb;
			if ((pk & 3) == 0) {
				v = 0;
				nb = 2;
			} else if ((pk & 3) == 1) {
				v = 4;
				nb = 2;
			} else if ((pk & 3) == 2) {
				v = 3;
				nb = 2;
			} else if ((pk & 7) == 3) {
				v = 2;
				nb = 3;
			} else {
				v = pk == 7 ? 1 : 5;
				nb = 4;
			}
			b.acc >>= nb;
			b.navail -= nb;
			const u32 sym = ut8(BR_CL_ORDER, i);
			if ((u32)lane == sym)
				cl_len = v;
			if (v) {
				space -= 32 >> v;
				ncodes++;
				if (space <= 0)
					break;
			}
		}
		if (ncodes != 1 && space != 0)
			return b;
		if (lane < 18)
			L.tmp[lane] = (u8)cl_len;
		wv_sync();
		br_build(L.clrec, L.tmp, 18, false, lane);
	}
	const BrTree cl = br_tree_load(L.clrec, lane);
	u32 sym = 0, prev = 8, repeat = 0, repeat_len = 0;
	int sp = 32768;
	bool bad = false;
	while (sym < A && sp > 0) {
		const u32 v = br_sym8(b, cl, bad, lane);
		if (bad)
			return b;
		if (v < 16) {
			repeat = 0;
			if (lane == 0)
				L.lens[sym] = (u8)v;
			sym++;
			if (v) {
				prev = v;
				sp -= 32768 >> v;
			}
		} else {
			const u32 xb = v == 16 ? 2 : 3, nl = v == 16 ? prev : 0;
			if (repeat_len != nl) {
				repeat = 0;
				repeat_len = nl;
			}
			const u32 old = repeat;
			if (repeat > 0)
				repeat = (repeat - 2) << xb;
			repeat += br_get(b, xb, lane) + 3;
			const u32 delta = repeat - old;
			if (sym + delta > A)
				return b;
			for (u32 i = (u32)lane; i < delta; i += 64)
				L.lens[sym + i] = (u8)nl;
			sym += delta;
			if (nl)
				sp -= (int)(delta << (15 - nl));
		}
	}
	if (sp != 0)
		return b;
	wv_sync();
	br_build(rec, L.lens, A, sym16, lane);
	*ok = 1;
	return b;
}

template <typename LT>
static __device__ __forceinline__ bool br_read_code(BrBits &b, LT &L, u8 *rec, u32 A, bool sym16, int lane)
{
	u32 ok;
	b = br_pin(br_read_code_core(b, L, rec, A, sym16, lane, &ok));
	return wv_readfirst(ok) != 0;
}

/* ------------------------------------------------------------------ header pieces */
static __device__ __forceinline__ u32 br_varlen8(BrBits &b, int lane)
{
	if (!br_get(b, 1, lane))
		return 0;
	const u32 n = br_get(b, 3, lane);
	if (!n)
		return 1;
	return (1u << n) + br_get(b, n, lane);
}

/* per-category block-switch state: the two codes sit in the wave's HBM scratch (slots 2k, 2k+1) */
struct BrCat {
	u32 ntypes, type, prev, left;
};

static __device__ __forceinline__ u32 br_block_len(BrBits &b, const u8 *rec, bool &bad, int lane)
{
	const BrTree t = br_tree_load(rec, lane);
	const u32 s = br_sym16(b, t, bad, lane);
	if (bad || s >= 26)
		return 0;
	return ut16(BR_BLEN_BASE, s) + br_get(b, ut8(BR_BLEN_BITS, s), lane);
}

static __device__ __forceinline__ void br_switch(BrBits &b, BrCat &k, const u8 *bt, u32 cat, bool &bad, int lane)
{
	const BrTree t = br_tree_load(bt + (2 * cat) * BR_BT_STRIDE, lane);
	u32 ty = br_sym16(b, t, bad, lane);
	if (ty == 0)
		ty = k.prev;
	else if (ty == 1)
		ty = k.type + 1;
	else
		ty -= 2;
	if (ty >= k.ntypes)
		ty -= k.ntypes;
	k.prev = k.type;
	k.type = ty;
	k.left = br_block_len(b, bt + (2 * cat + 1) * BR_BT_STRIDE, bad, lane);
}

/* RFC 7932 7.3: context map of `size` entries into HBM at map; returns false on malformed input */
static __device__ ZMT_NOINLINE BrBits br_context_map_core(BrBits b, BrLds &L, u8 *map, u32 size, u8 *code_rec, u32 *res, int lane)
{
	res[0] = 0; /* ok */
	const u32 ntrees = br_varlen8(b, lane) + 1;
	res[1] = ntrees;
	for (u32 i = (u32)lane; i < size; i += 64)
		map[i] = 0;
	wave_mem_fence();
	if (ntrees == 1) {
		res[0] = 1;
		return b;
	}
	const u32 rlemax = br_get(b, 1, lane) ? br_get(b, 4, lane) + 1 : 0;
	if (!br_read_code(b, L, code_rec, ntrees + rlemax, true, lane))
		return b;
	const BrTree t = br_tree_load(code_rec, lane);
	bool bad = false;
	u32 pend = 0; /* lane j: entry (i & ~63) + j, flushed as whole groups of 64 */
	u32 i = 0;
	while (i < size) {
		const u32 s = br_sym16(b, t, bad, lane);
		if (bad || br_over(b))
			return b;
		u32 v = 0, reps = 1;
		if (s && s <= rlemax) {
			reps = (1u << s) + br_get(b, s, lane);
			if (reps > size - i)
				return b;
		} else if (s) {
			v = s - rlemax;
		}
		for (; reps; reps--) {
			if ((u32)lane == (i & 63))
				pend = v;
			i++;
			if ((i & 63) == 0 || i == size) {
				const u32 base = (i - 1) & ~63u;
				if (base + (u32)lane < i)
					map[base + (u32)lane] = (u8)pend;
			}
		}
	}
	wave_mem_fence();
	if (br_get(b, 1, lane)) {
		/* inverse move-to-front, 64 entries per round trip to the map */
		for (u32 j = (u32)lane; j < 256; j += 64)
			L.mtf[j] = (u8)j;
		wv_sync();
		for (u32 base = 0; base < size; base += 64) {
			const u32 mine = base + (u32)lane < size ? map[base + (u32)lane] : 0;
			u32 res = 0;
			const u32 n = size - base < 64 ? size - base : 64;
			for (u32 j = 0; j < n; j++) {
				const u32 idx = wv_readlane(mine, (int)j);
				const u32 v = wv_readfirst((u32)L.mtf[idx]);
				if ((u32)lane == j)
					res = v;
				if (idx) {
					/* mtf[1..idx] <- mtf[0..idx-1], then the value to the front */
					u32 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
					if ((u32)lane < idx)
						t0 = L.mtf[lane];
					if (64u + (u32)lane < idx)
						t1 = L.mtf[64 + lane];
					if (128u + (u32)lane < idx)
						t2 = L.mtf[128 + lane];
					if (192u + (u32)lane < idx)
						t3 = L.mtf[192 + lane];
					wv_sync();
					if ((u32)lane < idx)
						L.mtf[lane + 1] = (u8)t0;
					if (64u + (u32)lane < idx)
						L.mtf[65 + lane] = (u8)t1;
					if (128u + (u32)lane < idx)
						L.mtf[129 + lane] = (u8)t2;
					if (192u + (u32)lane < idx)
						L.mtf[193 + lane] = (u8)t3;
					if (lane == 0)
						L.mtf[0] = (u8)v;
					wv_sync();
				}
			}
			if (base + (u32)lane < size)
				map[base + (u32)lane] = (u8)res;
		}
		wave_mem_fence();
	}
	res[0] = 1;
	return b;
}

static __device__ __forceinline__ bool br_context_map(BrBits &b, BrLds &L, u8 *map, u32 size, u8 *code_rec, u32 &ntrees, int lane)
{
	u32 res[2];
	b = br_pin(br_context_map_core(b, L, map, size, code_rec, res, lane));
	ntrees = wv_readfirst(res[1]);
	return wv_readfirst(res[0]) != 0;
}

/* ------------------------------------------------------------------ dictionary words */
/* RFC 7932 section 8 + Appendix B: transform `tidx` of the `copy`-byte word `widx`, written at d;
 * returns the number of bytes, 0xFFFFFFFF when they do not fit `room`.  Lane 0 assembles the
 * (at most 24 + 2 x 8 byte) string in LDS, the wave stores it. */
static __device__ __forceinline__ u32 br_dict_word(BrLds &L, const u8 *blob, u32 copy, u32 widx, u32 tidx, u8 *d, u32 room, int lane)
{
	const u8 *dict = blob + uld32(blob + 8), *tr = blob + uld32(blob + 16) + 3 * tidx;
	const u8 *psmap = blob + uld32(blob + 20), *ps = blob + uld32(blob + 24);
	const u32 pre = uld8(tr), type = uld8(tr + 1), suf = uld8(tr + 2);
	const u8 *pp = ps + uld16(psmap + 2 * pre), *sp = ps + uld16(psmap + 2 * suf);
	const u32 pn = uld8(pp), sn = uld8(sp);
	const u8 *w = dict + wv_readfirst(BR_DICT_OFFS[copy]) + widx * copy;
	u32 wn = copy, skip = 0;
	if (type >= 12 && type <= 20) {
		skip = type - 11 < wn ? type - 11 : wn;
		wn -= skip;
	} else if (type >= 1 && type <= 9) {
		wn = type >= wn ? 0 : wn - type;
	}
	const u32 total = pn + wn + sn;
	if (total > room)
		return 0xFFFFFFFFu;
	u32 c = 0;
	if ((u32)lane < pn)
		c = pp[1 + lane];
	else if ((u32)lane < pn + wn)
		c = w[skip + (u32)lane - pn];
	else if ((u32)lane < total)
		c = sp[1 + (u32)lane - pn - wn];
	if ((u32)lane < total)
		L.tmp[lane] = (u8)c;
	wv_sync();
	if ((type == 10 || type == 11) && lane == 0) {
		u8 *q = L.tmp + pn;
		u32 left = wn;
		while (left) {
			u32 st;
			if (q[0] < 0xC0) {
				if (q[0] >= 'a' && q[0] <= 'z')
					q[0] ^= 32;
				st = 1;
			} else if (q[0] < 0xE0) {
				if (left > 1)
					q[1] ^= 32;
				st = 2;
			} else {
				if (left > 2)
					q[2] ^= 5;
				st = 3;
			}
			if (type == 10 || st >= left)
				break;
			q += st;
			left -= st;
		}
	}
	wv_sync();
	if ((u32)lane < total)
		d[lane] = L.tmp[lane];
	return total;
}

/* whole-wave copy of a match (plain or overlapping); same as wave_match of match_copy.h, inlined: the
 * kernel makes no calls, so no value has to sit in the sparse callee-saved register ranges */
static __device__ __forceinline__ void br_wave_match(u8 *d, u32 off, u32 ml, int lane)
{
	const u8 *s = d - off;
	if (off >= ml) {
		wave_copy(d, s, ml, lane);
	} else if (off >= 64) {
		for (u32 done = 0; done < ml; done += off) {
			const u32 n = ml - done < off ? ml - done : off;
			wave_copy(d + done, s + done, n, lane);
			wave_mem_fence();
		}
	} else {
		for (u32 i = (u32)lane; i < ml; i += 64)
			d[i] = s[i % off];
	}
}

#endif
