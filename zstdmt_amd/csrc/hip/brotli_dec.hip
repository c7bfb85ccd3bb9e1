/*
 * brotli_dec.hip -- brotli stream decoder for gfx950, one wave per record.
 *
 * Replaces BrotliDecoderDecompress as called per record by the reference
 * (/root/reference/lib/brotli-mt_decompress.c:344-346; one raw brotli stream per 16-byte-header
 * record, output capacity hint << 16, :236-239).  Format: RFC 7932, complete: uncompressed and
 * metadata meta-blocks, simple and complex prefix codes, block switching for the three
 * categories, literal and distance context maps (RLE + inverse move-to-front), the distance ring
 * buffer with its 16 short codes, static dictionary words with the 121 transforms.
 *
 * A brotli stream is ONE serial bitstream per record: every symbol's position depends on the
 * length of the symbol before it, so the wave decodes in wave-uniform control flow and uses its
 * lanes where the format has width:
 *   - the input is held in a register window (lane i = dword i of 256 stream bytes); the bit accumulator is refilled with v_readlane, no memory
 *     access sits on the decode chain;
 *   - prefix codes are kept in canonical form: lane l of a tree's vector holds the left-aligned
 *     end of the code range of length l, so ONE compare + ballot finds a symbol's length; the
 *     symbol itself comes from the tree's sorted-symbol array.  The trees in use (insert&copy
 *     tree of the block type, the four distance trees of the block type on four 16-lane groups,
 *     the literal tree when the meta-block has a single one) stay in registers; further trees
 *     live in LDS (first few) or in the wave's scratch in HBM;
 *   - literals collect in a register (lane = output position mod 64) and leave as coalesced
 *     stores; copies, uncompressed meta-blocks and dictionary words are moved by all 64 lanes;
 *   - tables are built lane-parallel (ballot counting sort of the code lengths).
 *
 * Constant data of the format (dictionary, transforms, context lookup): csrc/data/brotli_static.bin.
 */
#include "lz4_common.h"
#include "lz4_frame.h"
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
static __device__ ZMT_NOINLINE BrBits br_read_code_core(BrBits b, BrLds &L, u8 *rec, u32 A, bool sym16, int lane, u32 *ok)
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

static __device__ __forceinline__ bool br_read_code(BrBits &b, BrLds &L, u8 *rec, u32 A, bool sym16, int lane)
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

/* ------------------------------------------------------------------ the kernel
 * grid = min(nrec, resident waves); wave w decodes records w, w + grid, ...  out_len[r] receives the
 * decoded size, status[r] GPUMT_ST_OK / BAD_BLOCK (malformed or truncated) / SIZE_MISMATCH (does
 * not fit the record's capacity). */
#ifndef ZMT_EMU
#define BRT() (PROF ? (u64)clock64() : 0ull)
#else
#define BRT() 0ull
#endif
#define BRP(i)                                                                                     \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = BRT();                                                      \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

template <bool PROF>
static __device__ __forceinline__ void
brotli_dec_body(BrLds &L, const u8 *__restrict__ stream, const u64 *__restrict__ rec_off, const u32 *__restrict__ rec_len,
		u32 nrec, u8 *out_base, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap,
		u32 *__restrict__ out_len, u32 *__restrict__ status, u8 *__restrict__ scratch,
		const u8 *__restrict__ blob, unsigned long long *prof)
{
	const int lane = wv_lane();
	u64 pc[PROF ? 8 : 1] = {0}, tq = BRT();
	const u64 t_begin = tq;
	(void)t_begin;
	u8 *const G = scratch + (u64)blockIdx.x * BR_WSCRATCH;
	{
		const u8 *ctx = blob + uld32(blob + 12);
		for (u32 i = (u32)lane * 4; i < 2048; i += 256)
			*(u32 *)(L.lut + i) = ld32u(ctx + i);
		if (lane < 24) {
			L.kins[lane] = (u32)BR_INS_BASE[lane] | (u32)BR_INS_BITS[lane] << 24;
			L.kcopy[lane] = (u32)BR_COPY_BASE[lane] | (u32)BR_COPY_BITS[lane] << 24;
		}
	}
	wv_sync();

	for (u32 rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
		u8 *const out = out_base + out_off[rec];
		const u32 cap = wv_readfirst(out_cap[rec]);
		u32 stc = ST_OK;
		BrBits b;
		b.p = stream + rec_off[rec];
		b.n = wv_readfirst(rec_len[rec]);
		br_seek(b, 0, lane);
		u32 pos = 0;

		/* 9.1 window bits */
		u32 wbits = 16;
		if (br_get(b, 1, lane)) {
			u32 v = br_get(b, 3, lane);
			if (v) {
				wbits = 17 + v;
			} else {
				v = br_get(b, 3, lane);
				if (v == 1)
					stc = BRBAD(); /* large-window streams are not brotli-mt's */
				wbits = v ? 8 + v : 17;
			}
		}
		const u32 max_backward = (1u << wbits) - 16u;
		u32 rb0 = 16, rb1 = 15, rb2 = 11, rb3 = 4; /* rb3 = last distance */
		u32 litv = 0, lit_lo = 0;                  /* literals not stored yet: [lit_lo, pos) */
#define BR_FLUSH()                                                                                 \
	do {                                                                                       \
		if (lit_lo < pos) {                                                                \
			const u32 a_ = ((pos - 1) & ~63u) + (u32)lane;                             \
			if (a_ >= lit_lo && a_ < pos)                                              \
				out[a_] = (u8)litv;                                                \
			lit_lo = pos;                                                              \
		}                                                                                  \
	} while (0)

		/* Copies are not executed one by one: a copy costs a load -> store round trip to HBM, which
		 * would stall the serial decode for thousands of cycles per command.  Meta-blocks without
		 * literal context modelling never look at the output while decoding (positions follow
		 * from the lengths alone), so up to 64 copies wait in lane slots -- literals already go to
		 * their final positions -- and are then executed side by side: every lane performs its
		 * own copy once its source lies below the watermark W (the destination of the first
		 * unfinished copy; everything below W is complete). */
		u32 bm_pos = 0, bm_dist = 0, bm_len = 0, nbatch = 0;
#define BR_EXEC()                                                                                  \
	do {                                                                                       \
		if (nbatch) {                                                                      \
			wave_mem_fence();                                                          \
			const bool act_ = (u32)lane < nbatch;                                      \
			const u32 ml_ = act_ ? bm_len : 0;                                         \
			const u32 sp_ = bm_pos - bm_dist, eff_ = ml_ < bm_dist ? ml_ : bm_dist;    \
			bool fin_ = !act_;                                                         \
			for (;;) {                                                                 \
				const u64 unf_ = wv_ballot(!fin_);                                 \
				if (!unf_)                                                         \
					break;                                                     \
				const int fst_ = wv_ffs(unf_) - 1;                                 \
				const u32 W_ = wv_readlane(bm_pos, fst_);                          \
				const u32 hl_ = wv_readlane(ml_, fst_);                            \
				if (hl_ > BR_CAP) {                                                \
					br_wave_match(out + W_, wv_readlane(bm_dist, fst_), hl_, lane); \
					if (lane == fst_)                                          \
						fin_ = true;                                       \
				} else {                                                           \
					const bool rdy_ = !fin_ && ml_ <= BR_CAP && sp_ + eff_ <= W_; \
					if (rdy_) {                                                \
						g_match(out + bm_pos, bm_dist, ml_);               \
						fin_ = true;                                       \
					}                                                          \
				}                                                                  \
				wave_mem_fence();                                                  \
			}                                                                          \
			nbatch = 0;                                                                \
		}                                                                                  \
	} while (0)

		while (stc == ST_OK) {
			if (br_over(b)) {
				stc = BRBAD();
				break;
			}
			const u32 is_last = br_get(b, 1, lane);
			if (is_last && br_get(b, 1, lane))
				break;
			const u32 nib = br_get(b, 2, lane);
			if (nib == 3) {
				if (br_get(b, 1, lane)) {
					stc = BRBAD();
					break;
				}
				const u32 nb = br_get(b, 2, lane);
				u32 skip = 0;
				bool e = false;
				for (u32 i = 0; i < nb; i++) {
					const u32 v = br_get(b, 8, lane);
					if (i + 1 == nb && nb > 1 && v == 0)
						e = true;
					skip |= v << (8 * i);
				}
				if (nb)
					skip++;
				if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane))
					e = true;
				const u64 at = br_used(b) >> 3;
				if (e || br_over(b) || at + skip > b.n) {
					stc = BRBAD();
					break;
				}
				br_seek(b, (u32)at + skip, lane);
				if (is_last)
					break;
				continue;
			}
			u32 mlen = 0;
			{
				bool e = false;
				for (u32 i = 0; i < nib + 4; i++) {
					const u32 v = br_get(b, 4, lane);
					if (i + 1 == nib + 4 && nib && v == 0)
						e = true;
					mlen |= v << (4 * i);
				}
				if (e) {
					stc = BRBAD();
					break;
				}
			}
			mlen++;
			if (!is_last && br_get(b, 1, lane)) {
				/* uncompressed meta-block */
				if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane)) {
					stc = BRBAD();
					break;
				}
				const u64 at = br_used(b) >> 3;
				if (br_over(b) || at + mlen > b.n) {
					stc = BRBAD();
					break;
				}
				if (mlen > cap - pos) {
					stc = ST_SIZE_MISMATCH;
					break;
				}
				BR_FLUSH();
				BR_EXEC();
				wave_copy(out + pos, b.p + at, mlen, lane);
				wave_mem_fence();
				pos += mlen;
				lit_lo = pos;
				br_seek(b, (u32)at + mlen, lane);
				continue;
			}
			/* ---------------- compressed meta-block header (9.2) ---------------- */
			BrCat c0, c1, c2;
			bool bad = false;
#define BR_CAT_HDR(c, k)                                                                           \
	do {                                                                                       \
		c.ntypes = br_varlen8(b, lane) + 1;                                                \
		c.type = 0;                                                                        \
		c.prev = 1;                                                                        \
		c.left = 1u << 24;                                                                 \
		if (!bad && c.ntypes >= 2) {                                                       \
			if (!br_read_code(b, L, G + BR_G_BT + (2 * k) * BR_BT_STRIDE, c.ntypes + 2, true, lane) || \
			    !br_read_code(b, L, G + BR_G_BT + (2 * k + 1) * BR_BT_STRIDE, 26, true, lane))         \
				bad = true;                                                        \
			else                                                                       \
				c.left = br_block_len(b, G + BR_G_BT + (2 * k + 1) * BR_BT_STRIDE, bad, lane);  \
		}                                                                                  \
	} while (0)
			BR_CAT_HDR(c0, 0);
			BR_CAT_HDR(c1, 1);
			BR_CAT_HDR(c2, 2);
			if (bad || br_over(b)) {
				stc = BRBAD();
				break;
			}
			const u32 npostfix = br_get(b, 2, lane);
			const u32 ndirect = br_get(b, 4, lane) << npostfix;
			for (u32 i = 0; i < c0.ntypes; i++) {
				const u32 m = br_get(b, 2, lane);
				if (lane == 0)
					L.cmode[i] = (u8)m;
			}
			u32 ntl = 1, ntd = 1;
			if (!br_context_map(b, L, G + BR_G_LCMAP, 64 * c0.ntypes, G + BR_G_BT + 6 * BR_BT_STRIDE, ntl, lane) ||
			    !br_context_map(b, L, G + BR_G_DCMAP, 4 * c2.ntypes, G + BR_G_BT + 6 * BR_BT_STRIDE, ntd, lane)) {
				stc = BRBAD();
				break;
			}
			{
				/* map entries must name existing trees */
				bool e = false;
				for (u32 i = (u32)lane; i < 64 * c0.ntypes; i += 64)
					e |= G[BR_G_LCMAP + i] >= ntl;
				for (u32 i = (u32)lane; i < 4 * c2.ntypes; i += 64)
					e |= G[BR_G_DCMAP + i] >= ntd;
				if (wv_any(e)) {
					stc = BRBAD();
					break;
				}
			}
			const u32 dist_alphabet = 16 + ndirect + (48u << npostfix);
#define BR_LIT_REC(t) ((t) < BR_NLIT_LDS ? L.lit + (t) * BR_LIT_STRIDE : G + BR_G_LIT + (t) * BR_LIT_STRIDE)
#define BR_CMD_REC(t) ((t) < BR_NCMD_LDS ? L.cmd + (t) * BR_CMD_STRIDE : G + BR_G_CMD + (t) * BR_CMD_STRIDE)
#define BR_DIST_REC(t) ((t) < BR_NDIST_LDS ? L.dist + (t) * BR_DIST_STRIDE : G + BR_G_DIST + (t) * BR_DIST_STRIDE)
			for (u32 i = 0; i < ntl && !bad; i++)
				bad = !br_read_code(b, L, BR_LIT_REC(i), 256, false, lane) || br_over(b);
			for (u32 i = 0; i < c1.ntypes && !bad; i++) {
				bad = !br_read_code(b, L, BR_CMD_REC(i), 704, true, lane) || br_over(b);
				/* the decode loop wants the insert / copy length codes of an insert&copy symbol,
				 * not its number: rewrite the tree's symbol array once as
				 * insert code | copy code << 5 | "distance is the last one" << 10 (RFC 7932 section 5:
				 * 11 cells of 8 x 8 codes, code bases in units of 8 packed two bits per cell) */
				u16 *sy = (u16 *)(BR_CMD_REC(i) + 128);
				for (u32 k = (u32)lane; k < 704; k += 64) {
					const u32 v = sy[k];
					if (v < 704) {
						const u32 cell = v >> 6;
						const u32 icode = (((0x298500u >> (2 * cell)) & 3u) << 3) + ((v >> 3) & 7);
						const u32 ccode = (((0x262444u >> (2 * cell)) & 3u) << 3) + (v & 7);
						sy[k] = (u16)(icode | ccode << 5 | (v < 128 ? 1u << 10 : 0u));
					}
				}
				wv_sync();
				wave_mem_fence();
			}
			for (u32 i = 0; i < ntd && !bad; i++)
				bad = !br_read_code(b, L, BR_DIST_REC(i), dist_alphabet, true, lane) || br_over(b);
			if (bad) {
				stc = BRBAD();
				break;
			}
			/* ---------------- commands (section 10) ----------------
			 * The loop keeps its whole state in registers, wave-uniform values in SGPRs.
			 * Nothing inside the loop waits on HBM: symbols come from
			 * LDS (explicit ds reads -- a flat access would wait for every literal store still in
			 * flight, loads and stores share one in-order counter on gfx9), trees beyond the LDS
			 * budget and block switches are the (rare) exceptions. */
			const bool ctx_free = ntl == 1;
			bool hbad = false;
			const u8 *const gbt = G + BR_G_BT;
			/* vector of a tree record: explicit LDS / HBM branches, never a flat access */
#define BR_VEC(va_, vi_, ldsarr, nlds, stride, goff, t)                                            \
	do {                                                                                       \
		u64 e_ = 0;                                                                        \
		if (lane < 16) {                                                                   \
			if ((t) < (nlds))                                                          \
				e_ = *(const u64 *)((ldsarr) + (t) * (stride) + 8 * lane);         \
			else {                                                                     \
				e_ = *(const u64 *)(G + (goff) + (t) * (stride) + 8 * lane);       \
				BR_SETTLE(e_);                                                     \
			}                                                                          \
		}                                                                                  \
		va_ = (u32)e_;                                                                     \
		vi_ = (u32)(e_ >> 32);                                                             \
	} while (0)
			u32 lva, lvi; /* literal tree 0 (the only one when ctx_free) */
			BR_VEC(lva, lvi, L.lit, BR_NLIT_LDS, BR_LIT_STRIDE, BR_G_LIT, 0u);
			u32 cva, cvi; /* insert&copy tree of the block type */
			BR_VEC(cva, cvi, L.cmd, BR_NCMD_LDS, BR_CMD_STRIDE, BR_G_CMD, 0u);
			u32 dva = 0, dvi = 0; /* lane 16 k + l: vector of the tree of distance context k */
			u32 dtree_k = 0;      /* lanes 16 k ..: tree index of context k */
#define BR_LOAD_DIST()                                                                             \
	do {                                                                                       \
		const u32 k_ = (u32)lane >> 4;                                                     \
		const u32 t_ = G[BR_G_DCMAP + 4 * c2.type + k_];                                   \
		u64 e_;                                                                            \
		if (t_ < BR_NDIST_LDS)                                                             \
			e_ = *(const u64 *)(L.dist + t_ * BR_DIST_STRIDE + 8 * ((u32)lane & 15)); \
		else                                                                               \
			e_ = *(const u64 *)(G + BR_G_DIST + t_ * BR_DIST_STRIDE + 8 * ((u32)lane & 15)); \
		BR_SETTLE(e_);                                                                     \
		dva = (u32)e_;                                                                     \
		dvi = (u32)(e_ >> 32);                                                             \
		dtree_k = t_;                                                                      \
	} while (0)
#define BR_LOAD_LCMAP()                                                                            \
	do {                                                                                       \
		L.cmap_cur[lane] = G[BR_G_LCMAP + 64 * c0.type + (u32)lane];                       \
		wv_sync();                                                                         \
	} while (0)
			BR_LOAD_DIST();
			u32 p1 = 0, p2 = 0; /* the two bytes before pos (context modelling only) */
			if (!ctx_free) {
				BR_LOAD_LCMAP();
				BR_FLUSH();
				BR_EXEC();
				wave_mem_fence();
				p1 = pos > 0 ? uld8(out + pos - 1) : 0;
				p2 = pos > 1 ? uld8(out + pos - 2) : 0;
			}
			/* MLEN bytes must fit the record's capacity; the loop then only checks against MLEN */
			if (mlen > cap - pos) {
				stc = ST_SIZE_MISMATCH;
				break;
			}
			u32 left = mlen;
			BRP(0);
			while (left) {
				if (BR_RARE(c1.left == 0)) {
					br_switch(b, c1, gbt, 1, hbad, lane);
					BR_VEC(cva, cvi, L.cmd, BR_NCMD_LDS, BR_CMD_STRIDE, BR_G_CMD, c1.type);
				}
				c1.left--;
				u32 cs;
				{
					const u32 k = br_sym_index(b, cva, cvi, 0, hbad, lane);
					if (c1.type < BR_NCMD_LDS)
						cs = wv_readfirst(*(const u16 *)(L.cmd + c1.type * BR_CMD_STRIDE + 128 + 2 * k));
					else
						cs = wv_readfirst(*(const u16 *)(G + BR_G_CMD + c1.type * BR_CMD_STRIDE + 128 + 2 * k));
				}
				if (BR_RARE(hbad))
					break;
				BRP(1);
				const u32 icode = cs & 31, ccode = (cs >> 5) & 31; /* packed when the tree was read */
				const bool last_dist = (cs >> 10) & 1;
				const u32 ki = wv_readfirst(L.kins[icode]), kc = wv_readfirst(L.kcopy[ccode]);
				u32 ins = ki & 0xFFFFFFu, copy = kc & 0xFFFFFFu;
				{
					/* extra bits of both lengths, in one read when they fit (nearly always) */
					const u32 ib = ki >> 24, cb = kc >> 24;
					if (BR_OFTEN(ib + cb <= 24)) {
						const u32 x = br_get(b, ib + cb, lane);
						ins += x & ((1u << ib) - 1u);
						copy += x >> ib;
					} else {
						ins += br_get(b, ib, lane);
						copy += br_get(b, cb, lane);
					}
				}
				const u32 ins0 = ins;
				if (BR_RARE(ins > left)) {
					hbad = true;
					break;
				}
				left -= ins;
				if (ctx_free && c0.ntypes == 1) {
					/* one literal tree, one block type (what levels 0..4 write): no block switch
					 * can fall inside the run, and the run is cut where the pending-literal
					 * register fills up, so the inner loop carries no checks.  The symbol of
					 * literal i is fetched from LDS while literal i + 1 is being located. */
					u32 todo = ins;
					while (todo) {
						const u32 room = 64u - (pos & 63u);
						const u32 seg = todo < room ? todo : room;
						todo -= seg;
						u32 kprev = br_sym_index(b, lva, lvi, 0, hbad, lane);
						for (u32 i = 1; i < seg; i++) {
							const u32 syp = L.lit[128 + kprev];
							const u32 k = br_sym_index(b, lva, lvi, 0, hbad, lane);
							litv = (u32)lane == (pos & 63) ? syp : litv;
							pos++;
							kprev = k;
						}
						const u32 syl = L.lit[128 + kprev];
						litv = (u32)lane == (pos & 63) ? syl : litv;
						pos++;
						if (BR_RARE((pos & 63) == 0))
							BR_FLUSH();
					}
				} else if (ctx_free) {
					/* the symbol of literal i is fetched from LDS while literal i + 1 is being
					 * located in the bitstream */
					u32 kprev = 0xFFFFFFFFu;
					for (; ins; ins--) {
						if (c0.left == 0)
							br_switch(b, c0, gbt, 0, hbad, lane);
						c0.left--;
						u32 syp = 0;
						if (kprev != 0xFFFFFFFFu)
							syp = L.lit[128 + kprev];
						const u32 k = br_sym_index(b, lva, lvi, 0, hbad, lane);
						if (kprev != 0xFFFFFFFFu) {
							litv = (u32)lane == (pos & 63) ? syp : litv;
							pos++;
							if ((pos & 63) == 0)
								BR_FLUSH();
						}
						kprev = k;
					}
					if (kprev != 0xFFFFFFFFu) {
						const u32 syp = L.lit[128 + kprev];
						litv = (u32)lane == (pos & 63) ? syp : litv;
						pos++;
						if ((pos & 63) == 0)
							BR_FLUSH();
					}
				} else {
					for (; ins; ins--) {
						if (c0.left == 0) {
							br_switch(b, c0, gbt, 0, hbad, lane);
							if (hbad)
								break;
							BR_LOAD_LCMAP();
						}
						c0.left--;
						const u8 *lut = L.lut + ((u32)L.cmode[c0.type] << 9);
						const u32 cid = wv_readfirst((u32)lut[p1] | (u32)lut[256 + p2]);
						const u32 tr = wv_readfirst((u32)L.cmap_cur[cid]);
						u32 va, vi, sy;
						BR_VEC(va, vi, L.lit, BR_NLIT_LDS, BR_LIT_STRIDE, BR_G_LIT, tr);
						const u32 k = br_sym_index(b, va, vi, 0, hbad, lane);
						if (tr < BR_NLIT_LDS)
							sy = wv_readfirst(L.lit[tr * BR_LIT_STRIDE + 128 + k]);
						else
							sy = wv_readfirst(G[BR_G_LIT + tr * BR_LIT_STRIDE + 128 + k]);
						p2 = p1;
						p1 = sy;
						litv = (u32)lane == (pos & 63) ? sy : litv;
						pos++;
						if ((pos & 63) == 0)
							BR_FLUSH();
					}
				}
				if (hbad)
					break; /* a truncated stream shows at the end of the meta-block: windows past
						* the end read as zeros, the loop is bounded by MLEN */
				BRP(2);
				if (!left)
					break;
				/* ---- distance (section 4) ---- */
				u32 dist;
				bool push = true;
				if (last_dist) {
					dist = rb3;
					push = false;
				} else {
					if (BR_RARE(c2.left == 0)) {
						br_switch(b, c2, gbt, 2, hbad, lane);
						if (hbad)
							break;
						BR_LOAD_DIST();
					}
					c2.left--;
					const u32 dctx = copy > 4 ? 3 : copy - 2;
					const u32 k = br_sym_index(b, dva, dvi, dctx, hbad, lane);
					if (hbad)
						break;
					const u32 tr = wv_readlane(dtree_k, (int)(16 * dctx));
					u32 dc;
					if (tr < BR_NDIST_LDS)
						dc = wv_readfirst(*(const u16 *)(L.dist + tr * BR_DIST_STRIDE + 128 + 2 * k));
					else
						dc = wv_readfirst(*(const u16 *)(G + BR_G_DIST + tr * BR_DIST_STRIDE + 128 + 2 * k));
					if (dc < 16) {
						const u32 which = dc < 4 ? dc : dc < 10 ? 0 : 1;
						const u32 r = which == 0 ? rb3 : which == 1 ? rb2 : which == 2 ? rb1 : rb0;
						int del = 0;
						if (dc >= 4) {
							const u32 q = (dc - 4) % 6; /* -1 +1 -2 +2 -3 +3 */
							del = (int)(q / 2 + 1);
							if (!(q & 1))
								del = -del;
						}
						const long dd = (long)r + del;
						if (dd <= 0) {
							hbad = true;
							break;
						}
						dist = (u32)dd;
						push = dc != 0;
					} else if (dc < 16 + ndirect) {
						dist = dc - 15;
					} else {
						const u32 d = dc - ndirect - 16;
						const u32 hcode = d >> npostfix, lcode = d & ((1u << npostfix) - 1);
						const u32 nbits = 1 + (hcode >> 1);
						const u64 offset = ((2ull + (hcode & 1)) << nbits) - 4;
						const u64 dd = ((offset + br_get(b, nbits, lane)) << npostfix) + lcode + ndirect + 1;
						if (dd > 0x7FFFFFFCull) {
							hbad = true;
							break;
						}
						dist = (u32)dd;
					}
				}
				const u32 max_dist = pos < max_backward ? pos : max_backward;
				BR_FLUSH();
				BRP(3);
				if (BR_RARE(dist > max_dist)) {
					if (copy < 4 || copy > 24) {
						hbad = true;
						break;
					}
					const u32 id = dist - max_dist - 1;
					const u32 shift = ut8(BR_DICT_BITS, copy);
					const u32 widx = id & ((1u << shift) - 1), tidx = id >> shift;
					if (tidx >= 121) {
						hbad = true;
						break;
					}
					BR_EXEC();
					const u32 wn = wv_readfirst(br_dict_word(L, blob, copy, widx, tidx, out + pos, left, lane));
					if (wn == 0xFFFFFFFFu) {
						hbad = true;
						break;
					}
					if (wn == 0 && ins0 == 0) {
						hbad = true; /* a command without output: never written by an encoder */
						break;
					}
					if (!ctx_free && wn) {
						const u32 np1 = L.tmp[wn - 1], np2 = wn >= 2 ? (u32)L.tmp[wn - 2] : p1;
						p2 = wv_readfirst(np2);
						p1 = wv_readfirst(np1);
					}
					wave_mem_fence();
					pos += wn;
					lit_lo = pos;
					left -= wn;
					BRP(5);
				} else {
					if (BR_RARE(copy > left)) {
						hbad = true;
						break;
					}
					if (push) {
						rb0 = rb1;
						rb1 = rb2;
						rb2 = rb3;
						rb3 = dist;
					}
					if (ctx_free) {
						if ((u32)lane == nbatch) {
							bm_pos = pos;
							bm_dist = dist;
							bm_len = copy;
						}
						nbatch++;
						pos += copy;
						lit_lo = pos;
						left -= copy;
						if (BR_RARE(nbatch == 64)) {
							BR_EXEC();
							BRP(4);
						}
					} else {
						wave_mem_fence();
						const u8 *s = out + pos - dist;
						p1 = uld8(s + (copy - 1) % dist);
						p2 = uld8(s + (copy - 2) % dist);
						br_wave_match(out + pos, dist, copy, lane);
						wave_mem_fence();
						pos += copy;
						lit_lo = pos;
						left -= copy;
					}
				}
			}
			bad = hbad || br_over(b);
			if (bad && stc == ST_OK)
				stc = BRBAD();
			if (stc != ST_OK || is_last)
				break;
		}
		if (stc == ST_OK) {
			/* zero padding up to the byte boundary; bytes after it are ignored */
			if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane))
				stc = BRBAD();
			if (br_over(b))
				stc = BRBAD();
		}
		BR_FLUSH();
		BR_EXEC();
		wave_mem_fence();
		if (lane == 0) {
			status[rec] = stc;
			out_len[rec] = stc == ST_OK ? pos : 0;
		}
		BRP(6);
	}
#ifndef ZMT_EMU
	if (PROF && prof && lane == 0) {
		for (int i = 0; i < (PROF ? 7 : 1); i++)
			atomicAdd(prof + i, (unsigned long long)pc[i]);
		atomicAdd(prof + 8, (unsigned long long)(BRT() - t_begin));
		atomicAdd(prof + 9, 1ull);
	}
#endif
}

/* 4 waves per SIMD (128 VGPRs) and < 10 KiB of LDS: 16 records per CU.  One record keeps a wave busy
 * with a chain of dependent scalar and cross-lane operations, so the number of records in flight is
 * what sets the throughput. */
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
zmt_brotli_dec_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off, const u32 *__restrict__ rec_len,
		      u32 nrec, u8 *out_base, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap,
		      u32 *__restrict__ out_len, u32 *__restrict__ status, u8 *__restrict__ scratch,
		      const u8 *__restrict__ blob)
{
	__shared__ __attribute__((aligned(16))) BrLds L;
	brotli_dec_body<false>(L, stream, rec_off, rec_len, nrec, out_base, out_off, out_cap, out_len, status, scratch,
			       blob, nullptr);
}

#ifndef ZMT_EMU
/* same kernel with per-phase cycle counters (developer tool): 0 headers + tables, 1 insert&copy
 * symbols, 2 literals, 3 distances, 4 copy batches, 5 dictionary words / uncompressed, 6 rest */
extern "C" __global__ void __launch_bounds__(64)
zmt_brotli_dec_kernel_prof(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
			   const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base, const u64 *__restrict__ out_off,
			   const u32 *__restrict__ out_cap, u32 *__restrict__ out_len, u32 *__restrict__ status,
			   u8 *__restrict__ scratch, const u8 *__restrict__ blob, unsigned long long *prof)
{
	__shared__ __attribute__((aligned(16))) BrLds L;
	brotli_dec_body<true>(L, stream, rec_off, rec_len, nrec, out_base, out_off, out_cap, out_len, status, scratch,
			      blob, prof);
}
#endif
