/*
 * brotli_enc.hip -- brotli stream encoder for gfx950 (replaces BrotliEncoderCompress as called per
 * chunk by the reference, /root/reference/lib/brotli-mt_compress.c:269-272).  The bar for brotli is
 * decompress-identical (brotli's bytes differ between versions), so this is not a
 * restatement of libbrotli's match finder but an encoder built for the wave, valid RFC 7932:
 *
 *   zmt_brotli_enc_kernel       persistent waves, one 128 KiB block at a time.  Match finding and the
 *                               greedy parse are those of the zstd encoder (zstd_enc.hip: 64 positions
 *                               per step, LDS hash table, matches stay inside the block).  Every block
 *                               becomes one meta-block with its own three prefix codes (literals,
 *                               insert&copy lengths, distances; one block type each, no context
 *                               modelling, NPOSTFIX = NDIRECT = 0): histograms with LDS atomics,
 *                               code lengths <= 15 by the Kraft repair of the zstd literal coder,
 *                               canonical codes by ballots, the lengths written as a complex prefix
 *                               code whose code-length code is the flat 4-bit one.  A match with the
 *                               distance of the one before it uses the implicit / zero distance code.
 *                               The command stream is laid down 64 commands at a time: bit offsets by
 *                               prefix sums, every lane ORs its command and the lanes together the
 *                               batch's literal codes into an LDS stage that leaves as dword stores
 *                               (a bit-serial writer covers headers and oversize batches).  An empty metadata meta-block pads
 *                               every meta-block to a byte boundary, so the blocks of a chunk are
 *                               coded independently and concatenated; a block that does not shrink
 *                               becomes an uncompressed meta-block.
 *   zmt_brotli_assemble_kernel  one workgroup per chunk: 16-byte record header (hint = 64 KiB units the
 *                               decoder must provide, lib/brotli-mt_compress.c:294-304), the chunk's
 *                               blocks moved together, the final empty meta-block.
 */
#include <cstddef>
#include "lz4_common.h"
#include "lz4_frame.h"

#define BE_BLOCK 131072u
#define BE_BSTRIDE (BE_BLOCK + 16u) /* area of one block inside a record slot (same as zstd's) */
#define BE_HDR 32u
#ifndef BE_HLOG
#define BE_HLOG 12
#endif
#ifndef BE_LAZYW
#define BE_LAZYW 4u /* positions the greedy parse looks ahead for a longer match (0 = none) */
#endif
#ifndef BE_MINMATCH
#define BE_MINMATCH 6u /* (7 until round 5: with the look-ahead 6 gives 2.512 against 2.463 on the bench text) */
#endif
#define BE_MAXSEQ (BE_BLOCK / 4u)
#define BE_WSCRATCH (3u * BE_MAXSEQ * 4u) /* per persistent wave: the three sequence arrays */
/* 6 bytes hashed; quality tiers (the reference hands `level` to BrotliEncoderCompress, /root/reference/lib/brotli-mt_compress.c:269-272):
 * as in zstd_enc.hip what the wave-parallel match finder can trade is table size (LDS, waves per CU) and minimum
 * match against ratio -- bench text, 1 MiB chunks (emulator), round 5 with the look-ahead [rounds 3-4]: 4 Ki entries /
 * minimum match 6: 2.512 [4 Ki / 7: 2.450]; 8 Ki / 6: 2.616 [2.584]; 16 Ki / 6: 2.694 [2.659] (libbrotli quality 1: 2.81) */
template <int HLOG> static __device__ __forceinline__ u32 be_hash(u64 v)
{
	return (u32)(((v << 16) * 0x9E3779B185EBCA87ull) >> (64 - HLOG));
}
#ifndef BE_WAVES_PER_EU
#define BE_WAVES_PER_EU 4
#endif
#define BE_MAXLEN 15u
/* symbol space of the three alphabets inside the shared LDS arrays */
#define BE_LIT 0u
#define BE_CMD 256u
#define BE_DIST 960u
#define BE_NSYM 1024u

struct BEncLds {
	u32 kins[24], kcopy[24]; /* insert / copy length codes: base | extra bits << 24 */
	u32 misc[8];
	/* LAST member: the kernels of the higher quality tiers declare the rest of a larger hash table right behind
	 * the struct (BEncLdsExt), the table simply runs on */
	union {
		u16 table[1u << BE_HLOG]; /* match finding */
		struct {                  /* block assembly (the table is rebuilt for the next block) */
			u32 hist[BE_NSYM];
			u16 code[BE_NSYM]; /* canonical code, bits reversed (the stream is LSB first) */
			u8 len[BE_NSYM];
		};
	};
};
template <int HLOG> struct BEncLdsExt {
	BEncLds L;
	u16 more[(1u << HLOG) - (1u << BE_HLOG) + 8];
};
static_assert(sizeof(((BEncLds *)0)->table) >= BE_NSYM * 7, "entropy-phase arrays must fit the idle hash table");

static __device__ __forceinline__ void st64g(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }
static __device__ __forceinline__ int hb32(u32 v) { return 31 - __builtin_clz(v); }

__device__ static const u16 BE_INS_BASE[24] = {0, 1, 2, 3, 4, 5, 6, 8, 10, 14, 18, 26,
					       34, 50, 66, 98, 130, 194, 322, 578, 1090, 2114, 6210, 22594};
__device__ static const u8 BE_INS_BITS[24] = {0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 24};
__device__ static const u16 BE_COPY_BASE[24] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 18,
						22, 30, 38, 54, 70, 102, 134, 198, 326, 582, 1094, 2118};
__device__ static const u8 BE_COPY_BITS[24] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 24};

/* ------------------------------------------------------------------ bit writer (wave-uniform) */
struct BeBits {
	u8 *p;     /* next dword goes here */
	u64 acc;
	u32 n;     /* bits in acc, < 32 between calls */
};
static __device__ __forceinline__ void be_put(BeBits &w, u32 v, u32 n, int lane)
{
	w.acc |= (u64)(v & ((1u << n) - 1u)) << w.n; /* n <= 24; nothing of v beyond its n bits */
	w.n += n;
	if (w.n >= 32) {
		if (lane == 0)
			st32u(w.p, (u32)w.acc);
		w.p += 4;
		w.acc >>= 32;
		w.n -= 32;
	}
}
/* pad with zero bits to the next byte boundary and store what is left; returns the end of the stream */
static __device__ __forceinline__ u8 *be_finish(BeBits &w, int lane)
{
	const u32 nb = (w.n + 7) >> 3;
	if (lane == 0)
		for (u32 i = 0; i < nb; i++)
			w.p[i] = (u8)(w.acc >> (8 * i));
	w.p += nb;
	w.acc = 0;
	w.n = 0;
	return w.p;
}

/* ------------------------------------------------------------------ prefix codes
 * Code lengths (<= BE_MAXLEN) for the A symbols at hist[base..], complete by construction: Shannon
 * lengths ceil(log2(N / count)) first, then the Kraft sum is repaired to exactly 1 -- the rarest
 * symbols get longer while it is above, the most frequent whose step fits get shorter while it is
 * below (zstd_enc.hip ze_huf_build, generalised to any alphabet).  Returns the number of used symbols;
 * with fewer than two there is nothing to build (the caller writes a one-symbol code). */
static __device__ u32 be_code_lengths(BEncLds &L, u32 base, u32 A, u32 *one_sym, int lane)
{
	const u32 KF = 1u << BE_MAXLEN;
	u32 total = 0, used = 0, last = 0;
	for (u32 s = (u32)lane; s < A; s += 64) {
		const u32 c = L.hist[base + s];
		total += c;
		if (c) {
			used++;
			last = s;
		}
	}
	for (int d = 32; d; d >>= 1) {
		total += wv_shfl(total, lane ^ d);
		used += wv_shfl(used, lane ^ d);
		const u32 o = wv_shfl(last, lane ^ d);
		last = o > last ? o : last;
	}
	*one_sym = last;
	if (used < 2) {
		for (u32 s = (u32)lane; s < A; s += 64)
			L.len[base + s] = 0;
		wv_sync();
		return used;
	}
	u32 K = 0;
	for (u32 s = (u32)lane; s < A; s += 64) {
		const u32 c = L.hist[base + s];
		u32 l = 0;
		if (c) {
			l = 1;
			while (l < BE_MAXLEN && ((u64)c << l) < total)
				l++;
			K += KF >> l;
		}
		L.len[base + s] = (u8)l;
	}
	for (int d = 32; d; d >>= 1)
		K += wv_shfl(K, lane ^ d);
	wv_sync();
	for (u32 guard = 0; K != KF && guard < 16384; guard++) {
		const bool over = K > KF;
		const u32 deficit = over ? 0 : KF - K;
		u64 best = over ? ~0ull : 0ull; /* count << 16 | symbol */
		for (u32 s = (u32)lane; s < A; s += 64) {
			const u32 c = L.hist[base + s], l = L.len[base + s];
			if (!c)
				continue;
			const u64 key = (u64)c << 16 | s;
			if (over) {
				if (l < BE_MAXLEN && key < best)
					best = key;
			} else if (l > 1 && (KF >> l) <= deficit && key > best) {
				best = key;
			}
		}
		for (int d = 32; d; d >>= 1) {
			const u64 o = (u64)wv_shfl((u32)best, lane ^ d) | (u64)wv_shfl((u32)(best >> 32), lane ^ d) << 32;
			best = over ? (o < best ? o : best) : (o > best ? o : best);
		}
		if (best == (over ? ~0ull : 0ull))
			break; /* cannot happen for a histogram with two or more symbols */
		const u32 sym = (u32)best & 0xFFFFu;
		const u32 l = L.len[base + sym];
		wv_sync();
		if (lane == 0)
			L.len[base + sym] = (u8)(over ? l + 1 : l - 1);
		wv_sync();
		/* l -> l+1 gives back 2^(MAX-l-1); l -> l-1 takes 2^(MAX-l) more */
		K = over ? K - (KF >> (l + 1)) : K + (KF >> l);
	}
	return K == KF ? used : 0xFFFFFFFFu;
}

/* canonical codes of the lengths at len[base..base+A), bit-reversed for the LSB-first stream */
static __device__ void be_assign_codes(BEncLds &L, u32 base, u32 A, int lane)
{
	u32 cnt = 0; /* lane l: number of symbols of length l */
	for (u32 s = (u32)lane; s < A; s += 64)
		L.code[base + s] = 0; /* symbols without a code (and the one symbol of a zero-bit code) */
	for (u32 s0 = 0; s0 < A; s0 += 64) {
		const u32 s = s0 + (u32)lane;
		const u32 ln = s < A ? L.len[base + s] : 0;
		for (u32 l = 1; l <= BE_MAXLEN; l++) {
			const u64 m = wv_ballot(ln == l);
			if ((u32)lane == l)
				cnt += (u32)wv_popc(m);
		}
	}
	u32 code = 0;
	for (u32 l = 1; l <= BE_MAXLEN; l++) {
		const u32 c = wv_readlane(cnt, (int)l);
		if (c) {
			u32 run = code;
			for (u32 s0 = 0; s0 < A; s0 += 64) {
				const u32 s = s0 + (u32)lane;
				const bool mine = s < A && L.len[base + s] == l;
				const u64 m = wv_ballot(mine);
				if (mine) {
					const u32 v = run + wv_mbcnt(m);
					u32 r = 0;
					for (u32 k = 0; k < l; k++)
						r |= ((v >> k) & 1) << (l - 1 - k);
					L.code[base + s] = (u16)r;
				}
				run += (u32)wv_popc(m);
			}
		}
		code = (code + c) << 1;
	}
	wv_sync();
}

/* RFC 7932 3.4 / 3.5: the code of one alphabet into the stream */
static __device__ void be_write_code(BeBits &w, BEncLds &L, u32 base, u32 A, u32 used, u32 one_sym, int lane)
{
	if (used < 2) {
		/* simple code, NSYM = 1: the symbol costs zero bits */
		u32 bits = 0;
		for (u32 a = A - 1; a; a >>= 1)
			bits++;
		be_put(w, 1, 2, lane); /* HSKIP = 1 */
		be_put(w, 0, 2, lane); /* NSYM - 1 */
		be_put(w, used ? one_sym : 0, bits, lane);
		return;
	}
	/* complex code; code-length code: the 16 length symbols at 4 bits each, no repeat symbols */
	be_put(w, 0, 2, lane); /* HSKIP = 0 */
	for (u32 i = 0; i < 18; i++) {
		/* order 1 2 3 4 0 5 17 6 16 7 ... 15: value 4 is "01" (first bit 1), value 0 is "00" */
		const bool zero = i == 6 || i == 8;
		be_put(w, zero ? 0u : 1u, 2, lane);
	}
	u32 lastu = 0;
	for (u32 s = (u32)lane; s < A; s += 64)
		if (L.len[base + s])
			lastu = s;
	for (int d = 32; d; d >>= 1) {
		const u32 o = wv_shfl(lastu, lane ^ d);
		lastu = o > lastu ? o : lastu;
	}
	/* the decoder stops reading lengths once the code is complete: nothing after the last used symbol */
	for (u32 s0 = 0; s0 <= lastu; s0 += 64) {
		const u32 s = s0 + (u32)lane;
		const u32 mine = s <= lastu ? L.len[base + s] : 0;
		const u32 n = lastu - s0 < 63 ? lastu - s0 + 1 : 64;
		for (u32 j = 0; j < n; j++) {
			const u32 v = wv_readlane(mine, (int)j);
			/* 4-bit canonical code = the value, first bit of the code first */
			be_put(w, ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3), 4, lane);
		}
	}
}

/* OR the n <= 63 low bits of v into the LDS bit stage at bit offset `at` (lanes in parallel) */
static __device__ __forceinline__ void be_or_bits(u32 *stage, u32 at, u64 v, u32 n)
{
	if (!n)
		return;
	const u32 wi = at >> 5, sh = at & 31;
	const u32 w0 = (u32)(v << sh), w1 = (u32)(sh ? v >> (32 - sh) : v >> 32), w2 = sh ? (u32)(v >> (64 - sh)) : 0u;
	if (w0)
		atomicOr(&stage[wi], w0);
	if (w1)
		atomicOr(&stage[wi + 1], w1);
	if (w2)
		atomicOr(&stage[wi + 2], w2);
}
#ifdef ZMT_EMU
static inline bool getenv_serial() { return getenv("BE_SERIAL") != nullptr; }
#else
static __device__ __forceinline__ bool getenv_serial() { return false; }
#endif
#define BE_STAGE_WORDS 640u /* 20 480 bits: a batch of 64 commands is ~5 Kbit on text */

/* ------------------------------------------------------------------ commands
 * One sequence (insert literals, copy length, distance) -> insert&copy symbol, distance symbol and
 * the extra bits, per lane.  `last` = the distance equals the one of the previous command. */
struct BeCmd {
	u32 sym;          /* insert&copy symbol 0..703 */
	u32 ix, ib;       /* insert extra value / bits */
	u32 cx, cb;       /* copy extra value / bits */
	u32 dsym;         /* distance symbol, 0xFFFFFFFF = none (implicit) */
	u32 dx, db;
};
static __device__ __forceinline__ BeCmd be_command(const BEncLds &L, u32 ins, u32 copy, u32 dist, bool last, bool has_copy)
{
	BeCmd c;
	u32 ic = 0, cc = 0;
	for (u32 k = 1; k < 24; k++) {
		ic += (L.kins[k] & 0xFFFFFFu) <= ins;
		cc += (L.kcopy[k] & 0xFFFFFFu) <= copy;
	}
	c.ix = ins - (L.kins[ic] & 0xFFFFFFu);
	c.ib = L.kins[ic] >> 24;
	c.cx = copy - (L.kcopy[cc] & 0xFFFFFFu);
	c.cb = L.kcopy[cc] >> 24;
	const bool implicit = has_copy ? (last && ic < 8 && cc < 16) : (ic < 8);
	u32 cell;
	if (implicit) {
		cell = cc >> 3; /* 0, 1 */
	} else {
		const u32 r = ic >> 3, q = cc >> 3;
		cell = r == 0 ? (q == 0 ? 2u : q == 1 ? 3u : 6u) : r == 1 ? (q == 0 ? 4u : q == 1 ? 5u : 8u)
									   : (q == 0 ? 7u : q == 1 ? 9u : 10u);
	}
	c.sym = cell * 64 + ((ic & 7) << 3) + (cc & 7);
	c.dsym = 0xFFFFFFFFu;
	c.dx = c.db = 0;
	if (has_copy && !implicit) {
		if (last) {
			c.dsym = 0; /* "same as the last distance" */
		} else {
			/* NPOSTFIX = NDIRECT = 0: distance = ((2 + (h & 1)) << nbits) - 4 + extra + 1 */
			const u32 v = dist + 3;
			const u32 nbits = (u32)hb32(v) - 1;
			const u32 h = ((nbits - 1) << 1) | ((v >> nbits) & 1);
			c.dsym = 16 + h;
			c.dx = v & ((1u << nbits) - 1);
			c.db = nbits;
		}
	}
	return c;
}

template <int HLOG, u32 MM>
static __device__ __forceinline__ void
brotli_enc_body(BEncLds &L, const u8 *__restrict__ in, u64 n, u32 chunk, u32 nblk_total, u32 blk_per_rec,
		u8 *__restrict__ slots, u64 stride, u32 *__restrict__ blk_len, u8 *__restrict__ scratch)
{
	const int lane = wv_lane();
	/* 2^HLOG entries: runs on behind the struct for the larger tiers -- derived from the byte address of the enclosing
	 * object, not from the member array (zstd_enc.hip has the reason) */
	u16 *const tab = (u16 *)((u8 *)&L + offsetof(BEncLds, table));
	u8 *const wscr = scratch + (u64)blockIdx.x * BE_WSCRATCH;
	u32 *const sq_ll = (u32 *)wscr;
	u32 *const sq_ml = sq_ll + BE_MAXSEQ, *const sq_of = sq_ml + BE_MAXSEQ;
	if (lane < 24) {
		L.kins[lane] = (u32)BE_INS_BASE[lane] | (u32)BE_INS_BITS[lane] << 24;
		L.kcopy[lane] = (u32)BE_COPY_BASE[lane] | (u32)BE_COPY_BITS[lane] << 24;
	}
	wv_sync();

	for (u32 g = blockIdx.x; g < nblk_total; g += gridDim.x) {
		const u32 rec = g / blk_per_rec, bi = g % blk_per_rec;
		const u64 cstart = (u64)rec * chunk;
		const u32 clen = (u32)(n - cstart < chunk ? n - cstart : chunk);
		const u32 bstart = bi * BE_BLOCK;
		if (bstart >= clen) {
			if (lane == 0)
				blk_len[g] = 0; /* no such block (short last chunk, or an empty input) */
			continue;
		}
		const u32 bsize = clen - bstart < BE_BLOCK ? clen - bstart : BE_BLOCK;
		const u8 *src = in + cstart + bstart;
		u8 *out = slots + (u64)rec * stride + BE_HDR + (u64)bi * BE_BSTRIDE;

		/* ------------------------------------------------ match finding + greedy parse (zstd_enc.hip) */
		for (u32 i = (u32)lane; i < (1u << HLOG); i += 64)
			tab[i] = 0;
		wv_sync();
		u32 ns = 0, anchor = 0, cursor = 0;
		u32 r_ll = 0, r_ml = 0, r_of = 0;
		const u32 steps = bsize >= MM ? (bsize - MM) / 64 + 1 : 0;
/* All loads of the pipeline are unconditional (addresses clamped, results of invalid lanes ignored):
 * a load under an exec mask needs its destination initialised first, and that write would have to
 * wait for every load still in flight. */
#define BE_LOADV(t, V)                                                                             \
	do {                                                                                       \
		const u32 p_ = (t) * 64u + (u32)lane;                                              \
		(V) = ld64u(src + (p_ < bsize ? p_ : bsize - 1));                                  \
	} while (0)
#define BE_LOOKUP(t, V, Cc, M)                                                                     \
	do {                                                                                       \
		const u32 p_ = (t) * 64u + (u32)lane;                                              \
		const bool ok_ = (t) < steps && p_ + MM <= bsize;                                  \
		const u32 h_ = be_hash<HLOG>(V);                                                   \
		const u32 e_ = ok_ ? tab[h_] : 0;                                                  \
		wv_sync();                                                                         \
		if (ok_)                                                                           \
			tab[h_] = (u16)p_;                                                         \
		wv_sync();                                                                         \
		/* equal hashes inside one step: the highest position must stay, whatever order the   \
		 * LDS served the conflicting lanes in (a step never straddles a 64 Ki boundary) */    \
		while (wv_any(ok_ && tab[h_] < (u16)p_)) {                                         \
			if (ok_ && tab[h_] < (u16)p_)                                              \
				tab[h_] = (u16)p_;                                                 \
			wv_sync();                                                                 \
		}                                                                                  \
		u32 c_ = (p_ & ~0xFFFFu) | e_;                                                     \
		if (c_ >= p_)                                                                      \
			c_ -= 65536u;                                                              \
		(Cc) = (ok_ && c_ < p_) ? c_ : 0xFFFFFFFFu;                                        \
		/* 24 bytes of the candidate and of the input beyond the hashed 8: most matches are   \
		 * measured right here, without the wave-wide extension below */                      \
		{                                                                                  \
			const u8 *cp_ = src + ((Cc) != 0xFFFFFFFFu ? (Cc) : 0u);                   \
			const u8 *ip_ = src + (ok_ ? p_ : 0u);                                     \
			(M).v = (V);                                                               \
			(M).a = ld64u(cp_);                                                        \
			(M).b = ld64u(cp_ + 8);                                                    \
			(M).c = ld64u(cp_ + 16);                                                   \
			(M).d = ld64u(ip_ + 8);                                                    \
			(M).e = ld64u(ip_ + 16);                                                   \
		}                                                                                  \
	} while (0)
		struct Cmp {
			u64 v, a, b, c, d, e; /* input bytes 0..7, candidate bytes 0..23, input bytes 8..23 */
		};
		/* three register sets rotate by unrolling (copying a set would wait for its loads) */
		Cmp M[3]; /* compare data of steps t, t+1, t+2 at index step % 3 */
		u32 Cn[3];
		u64 V[3]; /* hashed input word of the step that is looked up next, same indexing */
		BE_LOADV(0u, V[0]);
		BE_LOADV(1u, V[1]);
		BE_LOADV(2u, V[2]);
		BE_LOOKUP(0u, V[0], Cn[0], M[0]);
		BE_LOOKUP(1u, V[1], Cn[1], M[1]);
		for (u32 t0 = 0; t0 < steps; t0 += 3) {
		ZMT_UNROLL
		for (int k = 0; k < 3; k++) {
			const u32 t = t0 + (u32)k;
			if (t >= steps)
				break;
			BE_LOADV(t + 3, V[k]);
			BE_LOOKUP(t + 2, V[(k + 2) % 3], Cn[(k + 2) % 3], M[(k + 2) % 3]);
			const Cmp &m0 = M[k];
			const u32 c0 = Cn[k];
			const u64 v0 = m0.v;
			const u32 p0 = t * 64u, p = p0 + (u32)lane;
			if (p0 + 64 > cursor) { /* else the whole step lies inside the previous match */
				const u64 x0 = v0 ^ m0.a, x1 = m0.d ^ m0.b, x2 = m0.e ^ m0.c;
				u32 m = x0   ? (u32)__builtin_ctzll(x0) >> 3
					: x1 ? 8u + ((u32)__builtin_ctzll(x1) >> 3)
					: x2 ? 16u + ((u32)__builtin_ctzll(x2) >> 3)
					     : 24u;
				const bool cand = c0 != 0xFFFFFFFFu && p >= cursor;
				if (cand && m > bsize - p)
					m = bsize - p;
				u64 mask = wv_ballot(cand && m >= MM);
				/* look-ahead, as in zstd_enc.hip (round 5): a match that starts d <= BE_LAZYW bytes further on wins when
				 * it is longer by more than the d literals it adds; settled for all positions of the step at once -- lane
				 * j reads the lengths of lanes j + 1 .. j + BE_LAZYW through DPP wave_shl:1 steps -- and the skipped
				 * positions leave the candidate mask */
				if (BE_LAZYW) {
					u32 w = (cand && m >= MM) ? m : 0u;
					bool lz = false;
					ZMT_UNROLL
					for (u32 d = 1; d <= BE_LAZYW; d++) {
						w = wv_shl1(w, 0u);
						lz = lz || (w != 0u && w >= m + d + 1u);
					}
					mask &= ~wv_ballot(lz && m < 24u);
				}
				while (mask) {
					const int j = wv_ffs(mask) - 1;
					mask &= mask - 1;
					const u32 pj = p0 + (u32)j;
					if (__builtin_expect(pj < cursor, 0))
						continue;
					const u32 cj = wv_readlane(c0, j);
					u32 ml = wv_readlane(m, j);
					if (__builtin_expect(ml == 24, 0)) {
						/* extend: 64 lanes x 8 bytes per step */
						for (u32 base = 24;; base += 512) {
							const u32 o = base + 8u * (u32)lane;
							u32 k = 0;
							bool stop = true;
							if (pj + o < bsize) {
								const u64 y = ld64u(src + pj + o) ^ ld64u(src + cj + o);
								k = y ? (u32)__builtin_ctzll(y) >> 3 : 8u;
								stop = k < 8;
							}
							const u64 sm = wv_ballot(stop);
							if (sm) {
								const int f = wv_ffs(sm) - 1;
								ml = base + 8u * (u32)f + wv_readlane(k, f);
								break;
							}
						}
						if (ml > bsize - pj)
							ml = bsize - pj;
					}
					/* sequences collect in registers (lane = index mod 64) and leave 64 at a time */
					{
						const bool me = (u32)lane == (ns & 63);
						r_ll = me ? pj - anchor : r_ll;
						r_ml = me ? ml : r_ml;
						r_of = me ? pj - cj : r_of;
					}
					ns++;
					if ((ns & 63) == 0) {
						sq_ll[ns - 64 + (u32)lane] = r_ll;
						sq_ml[ns - 64 + (u32)lane] = r_ml;
						sq_of[ns - 64 + (u32)lane] = r_of;
					}
					anchor = cursor = pj + ml;
					/* drop every candidate the match covers in one go */
					mask = cursor - p0 >= 64 ? 0 : mask & ~((1ull << (cursor - p0)) - 1);
				}
			}
		}
		}
#undef BE_LOADV
#undef BE_LOOKUP
		if ((u32)lane < (ns & 63)) { /* the sequences still in registers */
			sq_ll[(ns & ~63u) + (u32)lane] = r_ll;
			sq_ml[(ns & ~63u) + (u32)lane] = r_ml;
			sq_of[(ns & ~63u) + (u32)lane] = r_of;
		}
		wave_mem_fence();
		const u32 tail_lits = bsize - anchor; /* literals after the last match */
#ifdef ZMT_EMU
		if (getenv("ZMT_EMU_DEBUG") && lane == 0) {
			u32 p = 0;
			for (u32 i = 0; i < ns; i++) {
				fprintf(stderr, "S pos %u ll %u ml %u of %u\n", p + sq_ll[i], sq_ll[i], sq_ml[i], sq_of[i]);
				p += sq_ll[i] + sq_ml[i];
			}
			fprintf(stderr, "tail %u\n", tail_lits);
		}
#endif

		/* ------------------------------------------------ histograms */
		for (u32 i = (u32)lane; i < BE_NSYM; i += 64)
			L.hist[i] = 0;
		wv_sync();
		{
			u32 prev_of = 0, pos = 0; /* pos: block position of the first sequence of the batch */
			for (u32 s0 = 0; s0 < ns + 1; s0 += 64) {
				const u32 i = s0 + (u32)lane;
				const bool isseq = i < ns, istail = i == ns && tail_lits;
				const u32 ll = isseq ? sq_ll[i] : istail ? tail_lits : 0;
				const u32 ml = isseq ? sq_ml[i] : 0;
				const u32 of = isseq ? sq_of[i] : 0;
				u32 pof = wv_shfl(of, lane - 1);
				if (lane == 0)
					pof = prev_of;
				if (isseq || istail) {
					const BeCmd c = be_command(L, ll, isseq ? ml : 2, of, isseq && i > 0 && of == pof, isseq);
					atomicAdd(&L.hist[BE_CMD + c.sym], 1u);
					if (c.dsym != 0xFFFFFFFFu)
						atomicAdd(&L.hist[BE_DIST + c.dsym], 1u);
				}
				/* literal bytes of the batch, spread over the lanes (owner of literal t = last lane whose
				 * first literal index is <= t; the code array is not built yet and lends its LDS) */
				const u32 incl = wv_scan_incl(ll + ml);
				const u32 lp = pos + incl - (ll + ml);
				{
					u32 *const a_pre = (u32 *)L.code, *const a_lp = a_pre + 64;
					const u32 incl_ll = wv_scan_incl(ll);
					const u32 T = wv_readlane(incl_ll, 63);
					wv_sync();
					a_pre[lane] = incl_ll - ll;
					a_lp[lane] = lp;
					wv_sync();
					for (u32 t0 = 0; t0 < T; t0 += 64) {
						const u32 t = t0 + (u32)lane;
						u32 j = 0;
						for (u32 st = 32; st; st >>= 1)
							if (j + st < 64 && a_pre[j + st] <= t)
								j += st;
						if (t < T)
							atomicAdd(&L.hist[BE_LIT + src[a_lp[j] + (t - a_pre[j])]], 1u);
					}
				}
				pos += wv_readlane(incl, 63);
				prev_of = wv_readlane(of, 63);
			}
		}
		wv_sync();
		/* ------------------------------------------------ the three codes */
		u32 one_lit, one_cmd, one_dist;
		const u32 u_lit = be_code_lengths(L, BE_LIT, 256, &one_lit, lane);
		const u32 u_cmd = be_code_lengths(L, BE_CMD, 704, &one_cmd, lane);
		const u32 u_dist = be_code_lengths(L, BE_DIST, 64, &one_dist, lane);
		bool codes_ok = u_lit != 0xFFFFFFFFu && u_cmd != 0xFFFFFFFFu && u_dist != 0xFFFFFFFFu;
		{
			/* bits of all symbols (extra bits not counted): a block that cannot shrink is not coded */
			u64 bits = 0;
			for (u32 i = (u32)lane; i < BE_NSYM; i += 64)
				bits += (u64)L.hist[i] * L.len[i];
			for (int d = 32; d; d >>= 1)
				bits += (u64)wv_shfl((u32)bits, lane ^ d) | (u64)wv_shfl((u32)(bits >> 32), lane ^ d) << 32;
			if (bits / 8 + 600 > bsize)
				codes_ok = false;
		}
		be_assign_codes(L, BE_LIT, 256, lane);
		be_assign_codes(L, BE_CMD, 704, lane);
		be_assign_codes(L, BE_DIST, 64, lane);

		/* ------------------------------------------------ meta-block */
		BeBits w;
		w.p = out;
		w.acc = 0;
		w.n = 0;
		if (bi == 0)
			be_put(w, 3, 4, lane); /* WBITS = 18: "1" then 18 - 17 in three bits */
		/* MLEN - 1 in 4 or 5 nibbles: the last nibble must not be zero (9.2) */
		const u32 nib5 = bsize - 1 >= 65536u;
		be_put(w, 0, 1, lane);                          /* ISLAST */
		be_put(w, nib5, 2, lane);                       /* MNIBBLES = 4 + nib5 */
		be_put(w, bsize - 1, nib5 ? 20 : 16, lane);     /* MLEN - 1 */
		be_put(w, 0, 1, lane);                          /* ISUNCOMPRESSED */
		be_put(w, 0, 3, lane);          /* NBLTYPESL = NBLTYPESI = NBLTYPESD = 1 */
		be_put(w, 0, 6, lane);          /* NPOSTFIX = 0, NDIRECT = 0 */
		be_put(w, 0, 2, lane);          /* context mode of the one literal block type */
		be_put(w, 0, 2, lane);          /* NTREESL = NTREESD = 1 */
		be_write_code(w, L, BE_LIT, 256, u_lit, one_lit, lane);
		be_write_code(w, L, BE_CMD, 704, u_cmd, one_cmd, lane);
		be_write_code(w, L, BE_DIST, 64, u_dist, one_dist, lane);
		/* commands, in stream order.  The codes of the literals are looked up lane-parallel per
		 * 64-byte window of the input (lane t = byte wbase + t), the loop then only reads lanes. */
		const u32 room = bsize > 64 ? bsize - 64 : 0; /* stop when the block would not shrink */
		bool fits = codes_ok && bsize > 64;
#ifdef BE_SKIP_EMIT /* developer: time everything but the command emission */
		fits = false;
#endif
		{
			u32 prev_of = 0, pos = 0, wbase = 0x80000000u, wcode = 0, wlen = 0;
			for (u32 s0 = 0; s0 < ns + 1 && fits; s0 += 64) {
				const u32 i = s0 + (u32)lane;
				const bool isseq = i < ns, istail = i == ns && tail_lits;
				const u32 ll = isseq ? sq_ll[i] : istail ? tail_lits : 0;
				const u32 ml = isseq ? sq_ml[i] : 0;
				const u32 of = isseq ? sq_of[i] : 0;
				u32 pof = wv_shfl(of, lane - 1);
				if (lane == 0)
					pof = prev_of;
				BeCmd c = be_command(L, ll, isseq ? ml : 2, of, isseq && i > 0 && of == pof, isseq);
				/* symbol -> code | length << 16 */
				const u32 ccode = (u32)L.code[BE_CMD + c.sym] | (u32)L.len[BE_CMD + c.sym] << 16;
				const u32 dcode = c.dsym != 0xFFFFFFFFu
							  ? (u32)L.code[BE_DIST + c.dsym] | (u32)L.len[BE_DIST + c.dsym] << 16
							  : 0u;
				const u32 k = ns + (tail_lits ? 1u : 0u) - s0 < 64 ? ns + (tail_lits ? 1u : 0u) - s0 : 64;
				/* ---- the batch side by side: every lane places its own command, the literals of the
				 * batch are spread over the lanes, all into an LDS stage (the histograms are dead by
				 * now); a batch whose bits do not fit the stage goes through the serial loop below ---- */
				u32 *const stage = L.hist;
				u32 *const a_pre = L.hist + BE_STAGE_WORDS, *const a_lp = a_pre + 64, *const a_lbits = a_lp + 64;
				u32 *const a_lbase = a_lbits + 64, *const a_sh = a_lbase + 64;
				const u32 seqlen = ll + ml;
				const u32 incl_pos = wv_scan_incl(seqlen);
				const u32 lp = pos + incl_pos - seqlen; /* block position of this lane's first literal */
				const u32 incl_ll = wv_scan_incl(ll);
				const u32 T = wv_readlane(incl_ll, 63);
				const u32 cl = ccode >> 16, dl = dcode >> 16;
				const u32 hdrbits = (u32)lane < k ? cl + c.ib + c.cb : 0;
				const u32 dbits = (u32)lane < k && c.dsym != 0xFFFFFFFFu ? dl + c.db : 0;
				wv_sync();
				a_pre[lane] = incl_ll - ll;
				a_lp[lane] = lp;
				a_lbits[lane] = 0;
				a_lbase[lane] = 0;
				wv_sync();
				/* pass 1: literal bits per command; owner of literal t = last lane whose first literal
				 * index is <= t */
				u32 cum_carry = 0;
				for (u32 t0 = 0; t0 < T; t0 += 64) {
					const u32 t = t0 + (u32)lane;
					const bool v = t < T;
					u32 j = 0;
					for (u32 st = 32; st; st >>= 1)
						if (j + st < 64 && a_pre[j + st] <= t)
							j += st;
					const u32 byte = v ? src[a_lp[j] + (t - a_pre[j])] : 0;
					const u32 ln = v ? L.len[BE_LIT + byte] : 0;
					const u32 inc = wv_scan_incl(ln);
					if (v) {
						atomicAdd(&a_lbits[j], ln);
						if (t == a_pre[j])
							a_lbase[j] = cum_carry + inc - ln;
					}
					cum_carry += wv_readlane(inc, 63);
				}
				wv_sync();
				const u32 lbits = a_lbits[lane];
				const u32 tot = hdrbits + lbits + dbits;
				const u32 incl_s = wv_scan_incl(tot);
				const u32 S = w.n + incl_s - tot;
				const u32 B = wv_readlane(incl_s, 63);
				if (!getenv_serial() && w.n + B + 96 <= BE_STAGE_WORDS * 32) {
					const u32 nw = (w.n + B + 31) / 32 + 3;
					for (u32 q = (u32)lane; q < nw; q += 64)
						stage[q] = 0;
					a_sh[lane] = S + hdrbits - a_lbase[lane];
					wv_sync();
					if (lane == 0)
						stage[0] = (u32)w.acc; /* the bits not yet stored */
					wv_sync();
					if ((u32)lane < k) {
						be_or_bits(stage, S, (u64)(ccode & 0xFFFFu) | (u64)c.ix << cl | (u64)c.cx << (cl + c.ib), hdrbits);
						be_or_bits(stage, S + hdrbits + lbits, (u64)(dcode & 0xFFFFu) | (u64)c.dx << dl, dbits);
					}
					u32 cc2 = 0;
					for (u32 t0 = 0; t0 < T; t0 += 64) {
						const u32 t = t0 + (u32)lane;
						const bool v = t < T;
						u32 j = 0;
						for (u32 st = 32; st; st >>= 1)
							if (j + st < 64 && a_pre[j + st] <= t)
								j += st;
						const u32 byte = v ? src[a_lp[j] + (t - a_pre[j])] : 0;
						const u32 ln = v ? L.len[BE_LIT + byte] : 0;
						const u32 inc = wv_scan_incl(ln);
						if (v)
							be_or_bits(stage, a_sh[j] + cc2 + inc - ln, L.code[BE_LIT + byte], ln);
						cc2 += wv_readlane(inc, 63);
					}
					wv_sync();
					{
						const u32 nbits = w.n + B, nwords = nbits >> 5;
						for (u32 q = (u32)lane; q < nwords; q += 64)
							st32u(w.p + 4 * q, stage[q]);
						w.p += 4 * nwords;
						w.acc = stage[nwords]; /* bits of the partial word; zero above them */
						w.n = nbits & 31;
					}
					pos += wv_readlane(incl_pos, 63);
					wv_sync();
					if ((u32)(w.p - out) > room)
						fits = false;
					prev_of = wv_readlane(of, 63);
					continue;
				}
				for (u32 j = 0; j < k; j++) {
					const u32 cj = wv_readlane(ccode, (int)j);
					be_put(w, cj & 0xFFFFu, cj >> 16, lane);
					be_put(w, wv_readlane(c.ix, (int)j), wv_readlane(c.ib, (int)j), lane);
					be_put(w, wv_readlane(c.cx, (int)j), wv_readlane(c.cb, (int)j), lane);
					u32 lj = wv_readlane(ll, (int)j);
					for (; lj; lj--, pos++) {
						if (pos - wbase >= 64u) {
							if ((u32)(w.p - out) > room) {
								fits = false;
								break;
							}
							wbase = pos & ~63u;
							const u32 p = wbase + (u32)lane;
							const u32 byte = src[p < bsize ? p : bsize - 1];
							wcode = L.code[BE_LIT + byte];
							wlen = L.len[BE_LIT + byte];
						}
						be_put(w, wv_readlane(wcode, (int)(pos - wbase)), wv_readlane(wlen, (int)(pos - wbase)), lane);
					}
					if (!fits)
						break;
					const u32 dj = wv_readlane(dcode, (int)j);
					be_put(w, dj & 0xFFFFu, dj >> 16, lane);
					be_put(w, wv_readlane(c.dx, (int)j), wv_readlane(c.db, (int)j), lane);
					pos += wv_readlane(ml, (int)j);
					if ((u32)(w.p - out) > room) {
						fits = false;
						break;
					}
				}
				prev_of = wv_readlane(of, 63);
			}
		}
		u32 total;
		if (fits) {
			/* empty metadata meta-block: ISLAST 0, MNIBBLES "11", reserved 0, MSKIPBYTES 0, then zero
			 * bits up to the byte boundary -- the next meta-block starts on a byte */
			be_put(w, 0x06, 6, lane);
			total = (u32)(be_finish(w, lane) - out);
		} else {
			/* uncompressed meta-block: header, padding, the bytes */
			w.p = out;
			w.acc = 0;
			w.n = 0;
			if (bi == 0)
				be_put(w, 3, 4, lane);
			be_put(w, 0, 1, lane);
			be_put(w, nib5, 2, lane);
			be_put(w, bsize - 1, nib5 ? 20 : 16, lane);
			be_put(w, 1, 1, lane); /* ISUNCOMPRESSED */
			u8 *d = be_finish(w, lane);
			wave_mem_fence();
			wave_copy(d, src, bsize, lane);
			total = (u32)(d - out) + bsize;
		}
		wave_mem_fence();
		if (lane == 0)
			blk_len[g] = total;
	}
}

#define BE_KERNEL_ARGS                                                                                             \
	const u8 *__restrict__ in, u64 n, u32 chunk, u32 nblk_total, u32 blk_per_rec, u8 *__restrict__ slots,      \
		u64 stride, u32 *__restrict__ blk_len, u8 *__restrict__ scratch
/* qualities 0..3 */
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BE_WAVES_PER_EU, BE_WAVES_PER_EU)))
zmt_brotli_enc_kernel(BE_KERNEL_ARGS)
{
	__shared__ __attribute__((aligned(16))) BEncLds L;
	brotli_enc_body<BE_HLOG, BE_MINMATCH>(L, in, n, chunk, nblk_total, blk_per_rec, slots, stride, blk_len, scratch);
}
/* qualities 4..8 and 9..11: 8 Ki / 16 Ki table entries, minimum match 6 */
extern "C" __global__ void __launch_bounds__(64) zmt_brotli_enc_t2_kernel(BE_KERNEL_ARGS)
{
	__shared__ __attribute__((aligned(16))) BEncLdsExt<13> S;
	brotli_enc_body<13, 6u>(S.L, in, n, chunk, nblk_total, blk_per_rec, slots, stride, blk_len, scratch);
}
extern "C" __global__ void __launch_bounds__(64) zmt_brotli_enc_t3_kernel(BE_KERNEL_ARGS)
{
	__shared__ __attribute__((aligned(16))) BEncLdsExt<14> S;
	brotli_enc_body<14, 6u>(S.L, in, n, chunk, nblk_total, blk_per_rec, slots, stride, blk_len, scratch);
}

/* ---------------------------------------------------------------------------------------------
 * Record assembly: 16-byte brotli-mt header (lib/brotli-mt_compress.c:285-304: skippable magic, 8,
 * compressed size, "BR", hint), the chunk's blocks moved together, the closing empty last meta-block. */
extern "C" __global__ void __launch_bounds__(256)
zmt_brotli_assemble_kernel(u64 n, u32 chunk, u32 nrec, u32 blk_per_rec, u8 *__restrict__ slots, u64 stride,
			   const u32 *__restrict__ blk_len, u32 *__restrict__ rec_len)
{
	const u32 rec = blockIdx.x, t = threadIdx.x;
	if (rec >= nrec)
		return;
	const u64 cstart = (u64)rec * chunk;
	const u32 clen = (u32)(n - cstart < chunk ? n - cstart : chunk);
	u8 *slot = slots + (u64)rec * stride;
	u32 at = 16;
	const u32 nb = clen ? (clen + BE_BLOCK - 1) / BE_BLOCK : 0;
	for (u32 b = 0; b < nb; b++) {
		const u32 len = blk_len[(u64)rec * blk_per_rec + b];
		const u8 *s = slot + BE_HDR + (u64)b * BE_BSTRIDE;
		u8 *d = slot + at;
		/* moving left, regions may overlap: forward order, one 4 KiB piece at a time */
		for (u32 o = 0; o < len; o += 4096) {
			const u32 m = len - o < 4096 ? len - o : 4096;
			u64 a = 0, c = 0;
			u8 tail[16];
			const u32 i = 16 * t;
			if (i + 16 <= m) {
				a = ld64u(s + o + i);
				c = ld64u(s + o + i + 8);
			} else if (i < m) {
				for (u32 k = 0; k < m - i; k++)
					tail[k] = s[o + i + k];
			}
			__syncthreads();
			if (i + 16 <= m) {
				st64g(d + o + i, a);
				st64g(d + o + i + 8, c);
			} else if (i < m) {
				for (u32 k = 0; k < m - i; k++)
					d[o + i + k] = tail[k];
			}
			__syncthreads();
		}
		at += len;
	}
	if (t == 0) {
		if (nb == 0) {
			slot[at++] = 0x33; /* WBITS = 18 (bits 1,1,0,0), ISLAST, ISLASTEMPTY: the empty stream */
		} else {
			slot[at++] = 0x03; /* ISLAST, ISLASTEMPTY */
		}
		const u32 csz = at - 16;
		/* hint: 64 KiB units of output the decoder must provide (:294-304).  The reference writes
		 * inputsize >> 16 for a full chunk, which is one unit short when inputsize is not a multiple
		 * of 64 KiB (its own decoder then rejects the record): round up -- a larger hint stays
		 * decodable by the reference, and for multiples of 64 KiB the value is the reference's. */
		const u32 hint = clen < chunk ? (clen >> 16) + 1 : (chunk + 65535u) >> 16;
		slot[0] = 0x50; slot[1] = 0x2A; slot[2] = 0x4D; slot[3] = 0x18;
		slot[4] = 8; slot[5] = 0; slot[6] = 0; slot[7] = 0;
		slot[8] = (u8)csz; slot[9] = (u8)(csz >> 8); slot[10] = (u8)(csz >> 16); slot[11] = (u8)(csz >> 24);
		slot[12] = 0x42; slot[13] = 0x52; /* "BR" */
		slot[14] = (u8)hint; slot[15] = (u8)(hint >> 8);
		rec_len[rec] = at;
	}
}
