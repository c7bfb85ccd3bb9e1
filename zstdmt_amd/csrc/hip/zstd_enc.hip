/*
 * zstd_enc.hip -- zstd frame encoder for gfx950.
 *
 * Replaces ZSTD_compress(dst, cap, chunk, n, level) as called per chunk by the reference
 * (/root/reference/lib/zstd-mt_compress.c:285) and the record header emit (:296-302).  The bar for
 * this codec is decompress-identical (zstd's bytes are version dependent), so the
 * encoder is free to choose its own parse and entropy stage as long as every frame is valid
 * RFC 8878 and decodes to the chunk; it is designed for the GPU instead of restating zstd's
 * sequential match finder:
 *
 *   zmt_zstd_enc_kernel      one wave per 128 KiB block (persistent waves, blocks round-robin).
 *     match finding          64 positions per step, one per lane: hash of 6 bytes, 4096-entry u16 LDS
 *                            table (the newest position of a step wins, deterministically),
 *                            candidates verified with unaligned 8-byte compares per lane (20 bytes
 *                            forwards, 4 backwards), loads software-pipelined two steps ahead, long
 *                            matches extended 512 bytes per step by the whole wave; while the data shows
 *                            repeating offsets, every position is also compared at the newest offset
 *                            (RFC 8878 repeat offsets: compare data asked for a step ahead);
 *     parse                  greedy, leftmost match first, with a look-ahead of 4 positions (a clearly
 *                            longer match a few bytes on wins), settled per lane for the whole step;
 *                            the serial loop over the chosen matches only reads three values per
 *                            match; a match grows backwards into the literals in front of it;
 *     literals               gathered lane-per-run, one Huffman code per 128 KiB unit (its first block
 *                            carries the tree, the others are treeless), coded in parallel (histogram with LDS
 *                            atomics, repaired ceil(log2) code lengths, canonical codes by ballots,
 *                            bit packing by prefix sum + atomicOr into LDS), raw when that does not pay;
 *     sequences              the list is cut into up to 16 zstd blocks, FSE-coded by 16 lanes side by
 *                            side with tables fitted to the unit (described in its first block,
 *                            repeated by the others; predefined-size tables, see ze_normalize) or,
 *                            for a unit with few sequences, the predefined ones (RFC 8878 3.1.1.3.2.2).
 *     A block that does not shrink is stored raw.
 *   zmt_zstd_assemble_kernel one workgroup per chunk: skippable header + frame header
 *                            (single segment, content size) + the chunk's blocks moved together.
 *
 * Matches stay inside their block (blocks are independent: 65536 of them per 8 GiB).
 */
#include <cstddef>
#include "lz4_common.h"
#include "lz4_frame.h"

#define ZMT_ZSTD_MAGIC 0xFD2FB528u
#define ZE_BLOCK 131072u
#define ZE_BSTRIDE (ZE_BLOCK + 16u) /* area of one block inside a record slot */
#define ZE_HDR 32u                  /* room in front of the blocks for record + frame header */
/* 4 Ki hash entries: with the entropy-phase arrays overlaid on the idle table a wave needs < 10 KiB of
 * LDS, i.e. 16 chunk-waves per CU instead of 7 with 8 Ki entries -- 155 -> 104 ms per 8 GiB for 2.7 %
 * of ratio (2.36 -> 2.30) [MI355X] */
#ifndef ZE_HLOG
#define ZE_HLOG 12
#endif
#ifndef ZE_MINMATCH
#define ZE_MINMATCH 6u /* 7 until round 5: with the look-ahead in the parse 6 / 6 costs 1.3 ms per 8 GiB and gives 1.2 % */
#endif
#define ZE_MAXSEQ (ZE_BLOCK / 4u) /* capacity of the per-wave sequence arrays (>= BLOCK / MINMATCH) */
#define ZE_CAP 64u
#define ZE_FWD 20u /* bytes of a match measured by the lane that found it; longer ones by the whole wave */
/* accuracy logs of the three sequence tables (the predefined tables' sizes: the decoder's small-table variant applies).  The
 * table builder, the merged state-bit field of the sequence coder and its end flush all use these */
#define ZE_LL_LOG 6
#define ZE_OF_LOG 5
#define ZE_ML_LOG 6
static_assert(ZE_LL_LOG + ZE_OF_LOG + ZE_ML_LOG <= 31, "the three state-bit counts of a sequence leave as one bit-writer field of at most 31 bits");
#define ZE_HUF_MAXLOG 10 /* longest literal code: 10 bits keep the decoder's table at 2 KiB (format max 11) */
/* While a block is assembled the hash table is idle: its LDS doubles as the bit-packing stage of the
 * Huffman coder plus the other entropy-phase arrays (ZEncLds), ZE_ENT_BYTES of them */
#define ZE_ENT_BYTES (1024u + 512u + 256u + 136u + 8u + 1536u)
#define ZE_STAGE_WORDS (((2u << ZE_HLOG) - ZE_ENT_BYTES) / 4u)
/* bytes hashed = the shortest match taken.  (Rounds 2-4: 7 / 7 -- hashing 6 with minimum match 7 lets strings that differ
 * in the 7th byte evict each other from the small table, 2.428 vs 2.446 on the bench text, and 6 / 6 made 26 % more
 * sequences for the greedy parse.  With the look-ahead: 7 / 7 2.5590, 6 / 7 2.5426, 8 / 7 2.4924, 5 / 5 2.5849, 6 / 6
 * 2.5877 on the emulator; [MI355X] 6 / 6 96.9 ms and 2.5851 against 95.6 ms and 2.5544 for 7 / 7) */
#ifndef ZE_HBYTES
#define ZE_HBYTES 6
#endif
/* ZE_REP: matches at the previous match's offset are looked for at every position of a step (one coalesced load: the
 * lanes read consecutive addresses) and sequences carry RFC 8878 repeat-offset values.  Emulator, 1 MiB chunks, level 1:
 * JSON lines 9.22 -> 10.78, Python sources 3.33 -> 3.38, the bench text (random words: no structure to repeat) 2.560 ->
 * 2.557.  Shorter repeat matches than the tier's minimum lose on all three (2.515 at 4 bytes on the bench text).
 * ZE_LAZYW: how many positions the greedy parse looks ahead for a better match (0 = none): bench text, 7 bytes hashed /
 * minimum match 7: 2.503 (0), 2.522 (1), 2.550 (3), 2.560 (5, 6), 2.555 (8); 6 / 6: 2.5870 (4), 2.5877 (6), 2.5805 (8).
 * libzstd 1.4.9's `fast` strategy with the same table (hashLog 12, 128 KiB pieces) gets 2.57-2.60, with hashLog 14 and the
 * whole 1 MiB chunk (its level 1) 2.842 */
#ifndef ZE_REP
#define ZE_REP 1
#endif
#ifndef ZE_REPLIVE
#define ZE_REPLIVE 40u /* steps the look-out stays open behind a group of 64 sequences with repeats in it */
#endif
#ifndef ZE_REPBONUS
#define ZE_REPBONUS 1u
#endif
#ifndef ZE_LAZYW
#define ZE_LAZYW 4u
#endif
template <int HB, int HLOG> static __device__ __forceinline__ u32 ze_hash(u64 v)
{
	if (HB == 4)
		return ((u32)v * 2654435761u) >> (32 - HLOG);
	return (u32)(((v << (64 - 8 * HB)) * 0x9E3779B185EBCA87ull) >> (64 - HLOG));
}
/* Level tiers (the reference hands `level` to ZSTD_compress, /root/reference/lib/zstd-mt_compress.c:285): what the
 * wave-parallel match finder can trade is table size (LDS, i.e. waves per CU) and hash width against ratio.
 * Bench text, 1 MiB chunks (emulator, deterministic), round 5 (look-ahead + repeat offsets; rounds 2-4 in brackets):
 * tier 1 = 6 bytes hashed / minimum match 6 / 4 Ki entries: 2.588 [7 / 7 / 4 Ki: 2.503]; tier 2 = 6 / 6 / 8 Ki: 2.696
 * [2.610]; tier 3 = 6 / 6 / 16 Ki: 2.771 [2.672]; libzstd 1.4.9 level 1 (hashLog 14, the whole 1 MiB chunk as window):
 * 2.84, with this tier's table (hashLog 12, 128 KiB pieces): 2.57-2.60 (profiles/r05_sweeps/zstd_enc_steps.txt) */

struct ZEncLds {
	u16 st_ll[64], st_ml[64], st_of[32]; /* FSE state tables: predefined distributions, or fitted to the unit */
	u32 tt_ll[36][2], tt_ml[53][2], tt_of[29][2]; /* per symbol: deltaNbBits, deltaFindState */
	u32 llx[36], mlx[53];            /* value base | extra bits << 24 */
	u8 llcode[64], mlcode[128];      /* code of literal length v / match length v + 3 */
	u32 sb_lo[16], sb_hi[16], sb_bits[16]; /* runs: sequence range left to code, bitstream bytes */
	u32 misc[8];
	/* LAST member: the kernels of the higher level tiers declare the rest of a larger hash table right behind
	 * the struct (ZEncLdsExt), the table simply runs on */
	union {
		u16 table[1u << ZE_HLOG]; /* match finding: low 16 bits of the newest position of a hash */
		struct {                  /* block assembly (the table is rebuilt for the next block) */
			u32 bitstage[ZE_STAGE_WORDS]; /* Huffman bit packing */
			u32 hist[256];                /* literal histogram of the block being assembled */
			u16 hcode[256];               /* Huffman code | length << 11 */
			u8 hlen[256];
			u8 treebuf[136 + 8];          /* Huffman_Tree_Description of the unit's code */
			u32 stage[16][8][3];          /* 8 staged sequences (ll, ml, offset) per run */
		};
	};
};
template <int HLOG> struct ZEncLdsExt {
	ZEncLds L;
	u16 more[(1u << HLOG) - (1u << ZE_HLOG) + 8]; /* entries ZE_HLOG.. of a 2^HLOG-entry table */
};
static_assert(sizeof(((ZEncLds *)0)->table) >= ZE_STAGE_WORDS * 4 + ZE_ENT_BYTES, "entropy-phase arrays must fit the idle hash table");
static_assert(ZE_STAGE_WORDS >= 1024, "one packing round adds up to 176 words and the stage flushes at 3/4");

#ifndef ZE_COLD
#define ZE_COLD __noinline__ /* once-per-unit table code: kept out of the kernel's register allocation */
#endif
static __device__ __forceinline__ void st64g(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }
static __device__ __forceinline__ int hb32(u32 v) { return 31 - __builtin_clz(v); }

__device__ static const short ZE_LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
						2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ static const short ZE_OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
						1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__device__ static const short ZE_ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
						1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
						1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__device__ static const u32 ZE_LL_BASE[36] = {0,  1,  2,   3,   4,   5,    6,    7,    8,    9,     10,    11,
					       12, 13, 14,  15,  16,  18,   20,   22,   24,   28,    32,    40,
					       48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ static const u8 ZE_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,  0,  0,  1,  1,
					      1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ static const u32 ZE_ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16,
					       17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30,
					       31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83,
					       99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ static const u8 ZE_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,  0,  0,
					      0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  1,  1,  1,  1,
					      2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* FSE compression table of a normalized distribution (one lane): state table + per-symbol
 * transform, the encoder-side mirror of RFC 8878 4.1.1's decoding table */
static __device__ ZE_COLD void fse_ctable(u16 *state_tab, u32 (*tt)[2], const short *norm, int nsym, int log, u8 *scratch)
{
	const u32 size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
	u32 cumul[54];
	u32 high = size - 1, pos = 0;
	cumul[0] = 0;
	for (int s = 0; s < nsym; s++) {
		if (norm[s] == -1) {
			cumul[s + 1] = cumul[s] + 1;
			scratch[high--] = (u8)s;
		} else {
			cumul[s + 1] = cumul[s] + (u32)norm[s];
		}
	}
	for (int s = 0; s < nsym; s++) {
		for (int i = 0; i < norm[s]; i++) {
			scratch[pos] = (u8)s;
			do
				pos = (pos + step) & mask;
			while (pos > high);
		}
	}
	for (u32 u = 0; u < size; u++) {
		const u32 s = scratch[u];
		state_tab[cumul[s]++] = (u16)(size + u);
	}
	u32 total = 0;
	for (int s = 0; s < nsym; s++) {
		const int c = norm[s];
		if (c == 0) {
			tt[s][0] = ((u32)(log + 1) << 16) - size;
			tt[s][1] = 0;
		} else if (c == -1 || c == 1) {
			tt[s][0] = ((u32)log << 16) - size;
			tt[s][1] = total - 1;
			total++;
		} else {
			const u32 maxbits = (u32)(log - hb32((u32)c - 1));
			tt[s][0] = (maxbits << 16) - ((u32)c << maxbits);
			tt[s][1] = total - (u32)c;
			total += (u32)c;
		}
	}
}

/* ------------------------------------------------------------------ sequence tables of a unit
 * The three FSE tables are fitted to the unit's own code statistics (Compressed mode in the unit's
 * first block, Repeat mode in the others) when the unit has enough sequences to pay for the three
 * descriptions; the table sizes are those of the predefined tables (2^6 / 2^5 / 2^6 cells), so the
 * LDS arrays, the final-state widths and the decoder's small tables stay as they are.
 * [1 MiB of the bench text: code bits 102.3 -> 76.3 KB, ratio 2.29 -> 2.43; 2^9 / 2^8 / 2^9 cells
 * would give 72.5 KB] */
#ifndef ZE_ADAPT_MIN
#define ZE_ADAPT_MIN 256u /* sequences from which fitted tables pay for their descriptions */
#endif
#define ZE_DESC 192u      /* room in front of run 0's bitstream for the three descriptions */

/* counts -> cells of a 2^log table, every present symbol at least one (one lane).  Returns the
 * number of present symbols. */
static __device__ ZE_COLD u32 ze_normalize(const u32 *hist, int nsym, u32 total, int log, short *norm)
{
	const u32 size = 1u << log;
	u32 sum = 0, present = 0;
	for (int s = 0; s < nsym; s++) {
		const u32 c = hist[s];
		u32 q = c ? (u32)(((u64)c << log) / total) : 0;
		if (c && !q)
			q = 1;
		norm[s] = (short)q;
		sum += q;
		present += c != 0;
	}
	/* hand out what the rounding left to the symbols whose cells are fullest (largest count per
	 * cell), or take cells back from where a cell holds least */
	while (sum < size) {
		int best = -1;
		for (int s = 0; s < nsym; s++)
			if (hist[s] && (best < 0 || (u64)hist[s] * (u32)norm[best] > (u64)hist[best] * (u32)norm[s]))
				best = s;
		norm[best]++;
		sum++;
	}
	while (sum > size) {
		int best = -1;
		for (int s = 0; s < nsym; s++)
			if (norm[s] > 1 &&
			    (best < 0 || (u64)hist[s] * (u32)(norm[best] - 1) < (u64)hist[best] * (u32)(norm[s] - 1)))
				best = s;
		norm[best]--;
		sum--;
	}
	return present;
}

/* RFC 8878 4.1.1 description of a normalized distribution (no "less than one" entries); nsym =
 * last present symbol + 1.  One lane, output into LDS; returns its length. */
static __device__ ZE_COLD u32 ze_write_ncount(u8 *out, const short *norm, int nsym, int log)
{
	u64 acc = (u64)(log - 5);
	u32 nb = 4, len = 0;
	int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
	bool prev0 = false;
	while (sym < nsym && remaining > 1) {
		if (prev0) {
			int start = sym;
			while (sym < nsym && !norm[sym])
				sym++;
			if (sym == nsym)
				break;
			while (sym >= start + 24) {
				start += 24;
				acc |= 0xFFFFull << nb;
				nb += 16;
				while (nb >= 8) {
					out[len++] = (u8)acc;
					acc >>= 8;
					nb -= 8;
				}
			}
			while (sym >= start + 3) {
				start += 3;
				acc |= 3ull << nb;
				nb += 2;
			}
			acc |= (u64)(sym - start) << nb;
			nb += 2;
			while (nb >= 8) {
				out[len++] = (u8)acc;
				acc >>= 8;
				nb -= 8;
			}
		}
		{
			int count = norm[sym++];
			const int max = (2 * threshold - 1) - remaining;
			remaining -= count;
			count++;
			if (count >= threshold)
				count += max;
			acc |= (u64)(u32)count << nb;
			nb += (u32)nbits - (count < max ? 1u : 0u);
			prev0 = count == 1;
			while (remaining < threshold) {
				nbits--;
				threshold >>= 1;
			}
		}
		while (nb >= 8) {
			out[len++] = (u8)acc;
			acc >>= 8;
			nb -= 8;
		}
	}
	if (nb)
		out[len++] = (u8)acc;
	return len;
}

/* lane-private forward bit writer (LSB first), 4-byte flushes */
struct BitW {
	u8 *p, *limit;
	u64 acc;
	u32 nb;
	bool ovf;
};
static __device__ __forceinline__ void bw_add(BitW &w, u32 v, u32 n) /* n <= 31, w.nb <= 32 */
{
	w.acc |= (u64)(v & ((1u << n) - 1)) << w.nb;
	w.nb += n;
	if (w.nb >= 32) {
		if (w.p + 4 > w.limit)
			w.ovf = true;
		else
			st32u(w.p, (u32)w.acc);
		w.p += 4;
		w.acc >>= 32;
		w.nb -= 32;
	}
}

/* The literals of one sequence (len <= ZE_CAP) copied by its lane, 64 sequences side by side, in two halves: all loads of a batch, then all
 * its stores.  Every load of the batch is issued
 * before its first store: the memory counter is in order, so a load behind a store waits for the store, and the three shapes of
 * the old routine (8-byte loop / two 4-byte pieces / bytes) each ran their own load -> store chain, one after the other, for the
 * lanes that took them -- ~8 memory round trips per batch, 11 % of the encoder (profiles/r06_sweeps/zstd_enc_gather.txt).  Reads up
 * to 7 bytes behind a run shorter than 8 (inside the input's slack, include/gpumt.h). */
struct ZeLits {
	u64 head, tail, v[ZE_CAP / 8 - 1];
};
static __device__ __forceinline__ void ze_lits_load(ZeLits &q, const u8 *s, u32 len)
{
	const bool big = len >= 8u;
	q.head = (len != 0 && !big) ? ld64u(s) : 0ull;    /* a run of 1..7 bytes */
	q.tail = big ? ld64u(s + len - 8) : 0ull;         /* the last 8 bytes of a longer one (may overlap) */
	ZMT_UNROLL
	for (u32 k = 0; k < ZE_CAP / 8 - 1; k++) {
		q.v[k] = 0;
		if (wv_any(len > 8u * k + 8u)) { /* qword k lies inside the run and the tail's 8 bytes do not reach back over all of it */
			if (len > 8u * k + 8u)
				q.v[k] = ld64u(s + 8u * k);
		}
	}
}
static __device__ __forceinline__ void ze_lits_store(const ZeLits &q, u8 *d, u32 len)
{
	ZMT_UNROLL
	for (u32 k = 0; k < ZE_CAP / 8 - 1; k++) {
		if (wv_any(len > 8u * k + 8u)) {
			if (len > 8u * k + 8u)
				st64g(d + 8u * k, q.v[k]);
		}
	}
	if (len >= 8u) {
		st64g(d + len - 8, q.tail);
	} else if (len >= 4u) {
		st32u(d, (u32)q.head);
		st32u(d + len - 4, (u32)(q.head >> (8u * (len - 4u))));
	} else if (len != 0) {
		d[0] = (u8)q.head;
		if (len > 1u)
			d[1] = (u8)(q.head >> 8);
		if (len > 2u)
			d[2] = (u8)(q.head >> 16);
	}
}

/* ------------------------------------------------------------------ Huffman literals
 * RFC 8878 3.1.1.3.1 / 4.2: literals sections of type Compressed / Treeless_Literals_Block, the tree
 * described by direct 4-bit weights or FSE-coded weights, four streams.  Everything but the choice of code lengths is
 * data parallel: histogram with LDS atomics, canonical codes with ballots, the bitstreams with a
 * prefix sum of code lengths and atomicOr into an LDS staging area.
 * Returns the size of the section written at dst (header included), 0 = not worth it / not
 * representable this way (caller stores the literals raw).  `stage` = ZE_STAGE_WORDS words of LDS
 * (the hash table is idle while a block is assembled). */
struct ZHuf {
	u32 log, maxsym; /* log == 0: no usable code (the unit's literals stay raw) */
	u32 tree_len;    /* bytes of L.treebuf */
};

/* Huffman weights 0..maxsym-1 FSE-coded (RFC 8878 4.2.1.2), for alphabets that reach beyond byte 128
 * where the direct 4-bit form does not exist.  One lane; `scr` = 512 bytes of idle LDS.  Writes
 * [size byte][table description][two-state bitstream] to out, returns its length or 0 (not
 * representable / not below 128 bytes). */
static __device__ u32 ze_fse_weights(const u8 *hlen, u32 log, u32 n, u8 *out, u8 *scr)
{
	u16 *st = (u16 *)scr;                     /* 64 states */
	u32(*tt)[2] = (u32(*)[2])(scr + 128);     /* 13 symbols */
	u8 *spread = scr + 128 + 104;             /* 64 bytes */
	short norm[13];
	u32 cnt[13];
	for (int i = 0; i < 13; i++)
		cnt[i] = 0;
	for (u32 i = 0; i < n; i++)
		cnt[hlen[i] ? log + 1 - hlen[i] : 0]++;
	u32 present = 0, maxc = 0, maxw = 0;
	for (int i = 0; i < 13; i++) {
		if (cnt[i]) {
			present++;
			maxw = (u32)i;
			if (cnt[i] > maxc)
				maxc = cnt[i];
		}
	}
	if (n < 2 || present < 2)
		return 0;
	const int tl = n <= 32 ? 5 : 6;
	const u32 size = 1u << tl;
	/* normalise to 2^tl: proportional, every present symbol at least 1, the rounding error goes to
	 * (or comes from) the largest entries */
	int sum = 0;
	for (int i = 0; i < 13; i++) {
		int v = 0;
		if (cnt[i]) {
			v = (int)((cnt[i] * size + n / 2) / n);
			if (v < 1)
				v = 1;
		}
		norm[i] = (short)v;
		sum += v;
	}
	while (sum != (int)size) {
		int best = -1;
		for (int i = 0; i < 13; i++)
			if (norm[i] > (sum > (int)size ? 1 : 0) && (best < 0 || norm[i] > norm[best]))
				best = i;
		if (best < 0)
			return 0;
		const int d = sum > (int)size ? -1 : 1;
		norm[best] = (short)(norm[best] + d);
		sum += d;
	}
	/* table description (inverse of the reader: RFC 8878 4.1.1) */
	u64 acc = (u64)(tl - 5);
	u32 nb = 4, o = 1; /* out[0] = total size, filled last */
	{
		int remaining = (int)size + 1, threshold = (int)size, nbits = tl + 1;
		bool prev0 = false;
		u32 sym = 0;
		while (remaining > 1) {
			if (prev0) {
				u32 start = sym;
				while (sym <= maxw && norm[sym] == 0)
					sym++;
				while (sym >= start + 3) {
					acc |= 3ull << nb;
					nb += 2;
					start += 3;
				}
				acc |= (u64)(sym - start) << nb;
				nb += 2;
			}
			const int c0 = norm[sym++];
			const int max = (2 * threshold - 1) - remaining;
			int c = c0 + 1;
			remaining -= c0;
			if (c >= threshold)
				c += max;
			acc |= (u64)(u32)c << nb;
			nb += (u32)nbits - (u32)(c < max);
			prev0 = (c0 == 0);
			while (remaining < threshold) {
				nbits--;
				threshold >>= 1;
			}
			while (nb >= 8) {
				out[o++] = (u8)acc;
				acc >>= 8;
				nb -= 8;
			}
		}
		if (nb)
			out[o++] = (u8)acc;
	}
	/* two interleaved states over the weights, last one first (decoder: state1 yields the even
	 * positions, state2 the odd ones) */
	fse_ctable(st, tt, norm, (int)maxw + 1, tl, spread);
	acc = 0;
	nb = 0;
#define WADD(v, k)                                                                                 \
	do {                                                                                       \
		acc |= (u64)((v) & ((1u << (k)) - 1)) << nb;                                       \
		nb += (k);                                                                         \
		while (nb >= 8) {                                                                  \
			if (o >= 132)                                                              \
				return 0;                                                          \
			out[o++] = (u8)acc;                                                        \
			acc >>= 8;                                                                 \
			nb -= 8;                                                                   \
		}                                                                                  \
	} while (0)
#define WSYM(i) ((u32)(hlen[i] ? log + 1 - hlen[i] : 0))
#define WINIT(S, w)                                                                                \
	do {                                                                                       \
		const u32 nbo_ = (tt[w][0] + (1u << 15)) >> 16;                                    \
		(S) = st[(((nbo_ << 16) - tt[w][0]) >> nbo_) + tt[w][1]];                          \
	} while (0)
#define WENC(S, w)                                                                                 \
	do {                                                                                       \
		const u32 nbo_ = ((S) + tt[w][0]) >> 16;                                           \
		WADD((S), nbo_);                                                                   \
		(S) = st[((S) >> nbo_) + tt[w][1]];                                                \
	} while (0)
	{
		u32 s1, s2, i = n;
		if (n & 1) {
			WINIT(s1, WSYM(i - 1));
			WINIT(s2, WSYM(i - 2));
			WENC(s1, WSYM(i - 3));
			i -= 3;
		} else {
			WINIT(s2, WSYM(i - 1));
			WINIT(s1, WSYM(i - 2));
			i -= 2;
		}
		while (i >= 2) {
			WENC(s2, WSYM(i - 1));
			WENC(s1, WSYM(i - 2));
			i -= 2;
		}
		WADD(s2, (u32)tl);
		WADD(s1, (u32)tl);
		WADD(1u, 1u); /* end mark */
		if (nb) {
			if (o >= 132)
				return 0;
			out[o++] = (u8)acc;
		}
	}
#undef WADD
#undef WSYM
#undef WINIT
#undef WENC
	if (o - 1 >= 128)
		return 0;
	out[0] = (u8)(o - 1);
	return o;
}

/* Code for all literals of one work unit (its blocks share it: the first one carries the tree, the
 * others are Treeless_Literals_Blocks, RFC 8878 3.1.1.3.1.1): histogram -> L.hcode / L.hlen */
static __device__ ZHuf ze_huf_build(ZEncLds &L, const u8 *lit, u32 regen, int lane)
{
	ZHuf hf = {0, 0, 0};
	if (regen < 256)
		return hf;
	/* histogram */
	for (u32 i = (u32)lane; i < 256; i += 64)
		L.hist[i] = 0;
	wv_sync();
	for (u32 i = 8u * (u32)lane; i < regen; i += 512) {
		if (i + 8 <= regen) {
			const u64 v = ld64u(lit + i);
			for (u32 k = 0; k < 8; k++)
				atomicAdd(&L.hist[(u32)(v >> (8 * k)) & 255], 1u);
		} else {
			for (u32 k = i; k < regen; k++)
				atomicAdd(&L.hist[lit[k]], 1u);
		}
	}
	wv_sync();
	/* lane owns symbols lane, lane+64, lane+128, lane+192 */
	u32 cnt[4], len[4];
	u32 present = 0, hi_sym = 0;
	for (u32 q = 0; q < 4; q++) {
		cnt[q] = L.hist[64 * q + (u32)lane];
		if (cnt[q]) {
			present++;
			hi_sym = 64 * q + (u32)lane;
		}
	}
	u32 nsym = present, maxsym = hi_sym;
	for (int d = 32; d; d >>= 1) {
		nsym += wv_shfl(nsym, lane ^ d);
		const u32 o = wv_shfl(maxsym, lane ^ d);
		maxsym = o > maxsym ? o : maxsym;
	}
	if (nsym < 2)
		return hf; /* one symbol: RLE would do, left raw here */
	/* code lengths: ceil(log2(N / count)) capped at ZE_HUF_MAXLOG, then repaired to a
	 * complete code (Kraft sum exactly KF = 2^ZE_HUF_MAXLOG in units of 1/KF) */
	constexpr u32 HM = ZE_HUF_MAXLOG, KF = 1u << HM;
	u32 K = 0;
	for (u32 q = 0; q < 4; q++) {
		u32 l = 0;
		if (cnt[q]) {
			l = 1;
			while (l < HM && (cnt[q] << l) < regen)
				l++;
			K += 1u << (HM - l);
		}
		len[q] = l;
	}
	for (int d = 32; d; d >>= 1)
		K += wv_shfl(K, lane ^ d);
	for (u32 guard = 0; K != KF && guard < 4096; guard++) {
		/* over-subscribed: lengthen the rarest symbol that still can; under-subscribed: shorten
		 * the most frequent symbol whose step fits the deficit */
		const bool over = K > KF;
		const u32 deficit = over ? 0 : KF - K;
		u32 best = over ? 0xFFFFFFFFu : 0;
		for (u32 q = 0; q < 4; q++) {
			if (!cnt[q])
				continue;
			const u32 key = cnt[q] << 8 | (64 * q + (u32)lane);
			if (over) {
				if (len[q] < HM && key < best)
					best = key;
			} else if (len[q] > 1 && (1u << (HM - len[q])) <= deficit && key > best) {
				best = key;
			}
		}
		for (int d = 32; d; d >>= 1) {
			const u32 o = wv_shfl(best, lane ^ d);
			best = over ? (o < best ? o : best) : (o > best ? o : best);
		}
		if (best == (over ? 0xFFFFFFFFu : 0u))
			return hf; /* cannot happen for a valid histogram; stay safe */
		const u32 sym = best & 255, sq_ = sym >> 6;
		if ((sym & 63) == (u32)lane) {
			const u32 d_ = over ? 1u : 0xFFFFFFFFu; /* +1 / -1 */
			len[0] += sq_ == 0 ? d_ : 0;
			len[1] += sq_ == 1 ? d_ : 0;
			len[2] += sq_ == 2 ? d_ : 0;
			len[3] += sq_ == 3 ? d_ : 0;
		}
		const u32 lnew = wv_shfl(sq_ == 0 ? len[0] : sq_ == 1 ? len[1] : sq_ == 2 ? len[2] : len[3], (int)(sym & 63));
		/* l -> l+1 removes 2^(HM-lnew); l -> l-1 adds 2^(HM-1-lnew) */
		K = over ? K - (1u << (HM - lnew)) : K + (1u << (HM - 1 - lnew));
	}
	if (K != KF)
		return hf;
	u32 log = 0;
	for (u32 q = 0; q < 4; q++)
		log = len[q] > log ? len[q] : log;
	for (int d = 32; d; d >>= 1) {
		const u32 o = wv_shfl(log, lane ^ d);
		log = o > log ? o : log;
	}
	/* canonical codes in the decoder's table order: weight ascending, then symbol value; a symbol
	 * of weight w owns 2^(w-1) cells starting at `start`; its code is start >> (w-1) */
	{
		u32 start = 0;
		for (u32 w = 1; w <= log; w++) {
			u32 at = start, total = 0;
			for (u32 q = 0; q < 4; q++) {
				const bool mine = cnt[q] && log + 1 - len[q] == w;
				const u64 m = wv_ballot(mine);
				if (mine) {
					const u32 st = at + wv_mbcnt(m) * (1u << (w - 1));
					L.hcode[64 * q + (u32)lane] = (u16)((st >> (w - 1)) | len[q] << 11);
				}
				at += (u32)wv_popc(m) << (w - 1);
				total += (u32)wv_popc(m);
			}
			start += total << (w - 1);
		}
		for (u32 q = 0; q < 4; q++) {
			L.hlen[64 * q + (u32)lane] = (u8)len[q];
			if (!cnt[q])
				L.hcode[64 * q + (u32)lane] = 0;
		}
	}
	wv_sync();
	/* worth it at all?  (estimate from the histogram) */
	u32 bits = 0;
	for (u32 q = 0; q < 4; q++)
		bits += cnt[q] * len[q];
	for (int d = 32; d; d >>= 1)
		bits += wv_shfl(bits, lane ^ d);
	if (1 + (maxsym + 1) / 2 + (bits + 7) / 8 + 64 >= regen)
		return hf;
	/* tree description: maxsym explicit weights (the last symbol's is implied), as 4-bit values
	 * while they reach no further than symbol 128, FSE-coded beyond that */
	u32 tlen;
	if (maxsym <= 128) {
		tlen = 1 + (maxsym + 1) / 2;
		if (lane == 0)
			L.treebuf[0] = (u8)(127 + maxsym);
		for (u32 i = (u32)lane; i < (maxsym + 1) / 2; i += 64) {
			const u32 s0 = 2 * i, s1 = 2 * i + 1;
			const u32 w0 = L.hlen[s0] ? log + 1 - L.hlen[s0] : 0;
			const u32 w1 = (s1 < maxsym && L.hlen[s1]) ? log + 1 - L.hlen[s1] : 0;
			L.treebuf[1 + i] = (u8)(w0 << 4 | w1);
		}
	} else {
		if (lane == 0)
			L.misc[0] = ze_fse_weights(L.hlen, log, maxsym, L.treebuf, (u8 *)L.stage);
		wv_sync();
		tlen = L.misc[0];
		if (tlen == 0)
			return hf; /* the weights do not fit the FSE form either: raw literals */
	}
	wv_sync();
	hf.log = log;
	hf.maxsym = maxsym;
	hf.tree_len = tlen;
	return hf;
}

/* One literals section with the unit's code: `with_tree` = Compressed_Literals_Block (type 2, the
 * tree described by direct 4-bit weights), else Treeless_Literals_Block (type 3).  Four streams,
 * packed by prefix sum of code lengths + atomicOr into the LDS stage (the idle hash table).
 * Returns the section size written at dst (header included), 0 = does not pay (caller stores raw). */
static __device__ u32 ze_huf_encode(ZEncLds &L, u32 *stage, const ZHuf hf, const u8 *lit, u32 regen, u8 *dst,
				    bool with_tree, int lane)
{
	if (hf.log == 0 || regen < 256)
		return 0;
	const u32 tree = with_tree ? hf.tree_len : 0;
	const u32 lh = regen < 1024 ? 3u : regen < 16384 ? 4u : 5u;
	const u32 limit = regen; /* a coded section that is not smaller than the literals is dropped */
	u8 *t = dst + lh;
	for (u32 i = (u32)lane; i < tree; i += 64)
		t[i] = L.treebuf[i];
	/* four streams */
	u8 *jump = t + tree, *sp = jump + 6;
	const u32 qn = (regen + 3) / 4;
	u32 ssz[4];
	for (u32 k = 0; k < 4; k++) {
		const u32 s_lo = k * qn, s_n = k < 3 ? qn : regen - 3 * qn;
		/* symbols are laid down from the last one (lowest bits) to the first; 512 per round:
		 * lane l takes the 8 symbols ending at hi - 8 l */
		u32 bitpos = 0, base_bits = 0; /* stream bits placed so far / bit offset of stage[0] */
		u32 outb = 0;                  /* bytes of this stream already at sp */
		for (u32 i = (u32)lane; i < ZE_STAGE_WORDS; i += 64)
			stage[i] = 0;
		wv_sync();
		for (u32 hi = s_n; hi > 0;) {
			const u32 take = hi < 512 ? hi : 512;
			const u32 mine_hi = hi > 8u * (u32)lane ? hi - 8u * (u32)lane : 0; /* exclusive end */
			const u32 mine_n = mine_hi > hi - take ? (mine_hi - (hi - take) < 8 ? mine_hi - (hi - take) : 8) : 0;
			u64 lo64 = 0;
			u32 hi32 = 0, nb = 0;
			if (mine_n) {
				/* the lane's (up to) eight symbols with ONE load -- eight byte loads in a loop were eight dependent memory round
				 * trips per round; a short first piece reads on into symbols that are not its own (or the buffer's 64-byte slack) */
				const u64 w8 = ld64u(lit + s_lo + mine_hi - mine_n);
				ZMT_UNROLL
				for (u32 jj = 0; jj < 8; jj++) { /* last symbol first */
					const u32 j = 7u - jj;
					if (j < mine_n) {
						const u32 e = L.hcode[(u32)(w8 >> (8u * j)) & 255u];
						const u32 l = e >> 11, c = e & 2047;
						if (nb < 64)
							lo64 |= (u64)c << nb;
						if (nb + l > 64)
							hi32 |= nb >= 64 ? c << (nb - 64) : c >> (64 - nb);
						nb += l;
					}
				}
			}
			const u32 incl = wv_scan_incl(nb);
			const u32 off = bitpos - base_bits + incl - nb; /* bit offset inside the stage */
			if (nb) {
				const u32 wi = off >> 5, sh = off & 31;
				const u64 a = lo64 << sh;
				atomicOr(&stage[wi], (u32)a);
				atomicOr(&stage[wi + 1], (u32)(a >> 32));
				const u64 b = (sh ? lo64 >> (64 - sh) : 0) | (u64)hi32 << sh;
				if (b) {
					atomicOr(&stage[wi + 2], (u32)b);
					atomicOr(&stage[wi + 3], (u32)(b >> 32));
				}
			}
			bitpos += wv_readlane(incl, 63);
			hi -= take;
			wv_sync();
			/* flush whole words once the stage is three quarters full */
			if (bitpos - base_bits > (ZE_STAGE_WORDS * 3 / 4) * 32 || hi == 0) {
				if (hi == 0) {
					if (lane == 0)
						atomicOr(&stage[(bitpos - base_bits) >> 5], 1u << ((bitpos - base_bits) & 31));
					bitpos++;
					wv_sync();
				}
				const u32 nwords = hi == 0 ? (bitpos - base_bits + 31) >> 5 : (bitpos - base_bits) >> 5;
				const u32 nbytes = hi == 0 ? (bitpos - base_bits + 7) >> 3 : nwords * 4;
				for (u32 i = (u32)lane; i < nwords; i += 64) {
					const u32 v = stage[i];
					if (4 * i + 4 <= nbytes) {
						st32u(sp + outb + 4 * i, v);
					} else {
						for (u32 b2 = 0; 4 * i + b2 < nbytes; b2++)
							sp[outb + 4 * i + b2] = (u8)(v >> (8 * b2));
					}
				}
				const u32 carry = hi == 0 ? 0 : stage[nwords];
				wv_sync();
				for (u32 i = (u32)lane; i < ZE_STAGE_WORDS; i += 64)
					stage[i] = 0;
				wv_sync();
				if (lane == 0)
					stage[0] = carry;
				outb += nbytes;
				base_bits += nwords * 32;
				wv_sync();
			}
		}
		ssz[k] = outb;
		sp += outb;
		if ((u32)(sp - dst) + 8 >= limit)
			return 0; /* incompressible slice: the room check of the caller covers raw only */
	}
	const u32 csz = tree + 6 + ssz[0] + ssz[1] + ssz[2] + ssz[3];
	if (lane == 0) {
		jump[0] = (u8)ssz[0];
		jump[1] = (u8)(ssz[0] >> 8);
		jump[2] = (u8)ssz[1];
		jump[3] = (u8)(ssz[1] >> 8);
		jump[4] = (u8)ssz[2];
		jump[5] = (u8)(ssz[2] >> 8);
		/* Literals_Section_Header: type 2 / 3, size format by regen, both sizes */
		const u32 ty = with_tree ? 2u : 3u;
		if (lh == 3) {
			const u32 hv = ty | 1u << 2 | regen << 4 | csz << 14;
			dst[0] = (u8)hv;
			dst[1] = (u8)(hv >> 8);
			dst[2] = (u8)(hv >> 16);
		} else if (lh == 4) {
			const u32 hv = ty | 2u << 2 | regen << 4 | csz << 18;
			dst[0] = (u8)hv;
			dst[1] = (u8)(hv >> 8);
			dst[2] = (u8)(hv >> 16);
			dst[3] = (u8)(hv >> 24);
		} else {
			const u64 hv = (u64)ty | 3ull << 2 | (u64)regen << 4 | (u64)csz << 22;
			for (u32 k = 0; k < 5; k++)
				dst[k] = (u8)(hv >> (8 * k));
		}
	}
	return lh + csz;
}

/* FSE-code the sequences [lo, hi) of one sub-block into its own bitstream (one lane per sub-block,
 * up to ZE_G lanes side by side; the wave stages 8 sequences per sub-block and round into LDS) */
#define ZE_G 16u
#define ZE_BSTMP 20544u /* bytes of temporary bitstream per sub-block: 2048 sequences x 75 bits + slack */
#define ZE_LITBUF (ZE_BLOCK + 64u)
#define ZE_WSCRATCH (3u * ZE_MAXSEQ * 4u + ZE_G * ZE_BSTMP + ZE_LITBUF) /* per persistent wave */

#ifndef ZMT_EMU
#define ZET() (PROF ? (u64)clock64() : 0ull)
#else
#define ZET() 0ull
#endif
#define ZEP(i)                                                                                     \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = ZET();                                                      \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

template <bool PROF, int HB, u32 MM, int HLOG, bool REP = (ZE_REP != 0), u32 LAZYW = ZE_LAZYW>
static __device__ __forceinline__ void
zstd_enc_body(ZEncLds &L, const u8 *__restrict__ in, u64 n, u32 chunk, u32 nblk_total, u32 blk_per_rec,
	      u8 *__restrict__ slots, u64 stride, u32 *__restrict__ blk_len, u8 *__restrict__ scratch,
	      unsigned long long *prof)
{
	const int lane = wv_lane();
	u64 pc[PROF ? 8 : 1] = {0}, tq = ZET();
	const u64 t_begin = tq;
	(void)t_begin;
	u8 *const wscr = scratch + (u64)blockIdx.x * ZE_WSCRATCH;
	u32 *const sq_ll = (u32 *)wscr;
	u32 *const sq_ml = sq_ll + ZE_MAXSEQ, *const sq_of = sq_ml + ZE_MAXSEQ;
	u8 *const bstmp = wscr + 3u * ZE_MAXSEQ * 4u;
	u8 *const litbuf = bstmp + ZE_G * ZE_BSTMP;

	/* tables that do not depend on the data, once per wave: code tables (lane-parallel counts),
	 * predefined FSE compression tables (lanes 0..2 build LL / ML / OF) */
	if (lane < 36)
		L.llx[lane] = ZE_LL_BASE[lane] | (u32)ZE_LL_BITS[lane] << 24;
	if (lane < 53)
		L.mlx[lane] = ZE_ML_BASE[lane] | (u32)ZE_ML_BITS[lane] << 24;
	wv_sync();
	{
		u32 c = 0;
		for (u32 k = 1; k < 36; k++)
			c += (L.llx[k] & 0xFFFFFFu) <= (u32)lane;
		L.llcode[lane] = (u8)c; /* literal lengths 0..63 */
		for (u32 v = (u32)lane; v < 128; v += 64) {
			c = 0;
			for (u32 k = 1; k < 53; k++)
				c += (L.mlx[k] & 0xFFFFFFu) <= v + 3;
			L.mlcode[v] = (u8)c; /* match lengths 3..130 */
		}
		u8 *scr = (u8 *)L.table + 64 * lane; /* the hash table is not live yet */
		if (lane == 0)
			fse_ctable(L.st_ll, L.tt_ll, ZE_LL_DEF, 36, ZE_LL_LOG, scr);
		else if (lane == 1)
			fse_ctable(L.st_ml, L.tt_ml, ZE_ML_DEF, 53, ZE_ML_LOG, scr);
		else if (lane == 2)
			fse_ctable(L.st_of, L.tt_of, ZE_OF_DEF, 29, ZE_OF_LOG, scr);
	}
	wv_sync();
	bool tabs_pre = true; /* st_* / tt_* hold the predefined distributions */

	for (u32 g = blockIdx.x; g < nblk_total; g += gridDim.x) {
		const u32 rec = g / blk_per_rec, bi = g % blk_per_rec;
		const u64 cstart = (u64)rec * chunk;
		const u32 clen = (u32)(n - cstart < chunk ? n - cstart : chunk);
		const u32 bstart = bi * ZE_BLOCK;
		if (bstart >= clen) {
			if (lane == 0)
				blk_len[g] = 0; /* no such block (short last chunk, or an empty input) */
			continue;
		}
		const u32 bsize = clen - bstart < ZE_BLOCK ? clen - bstart : ZE_BLOCK;
		const u32 last = bstart + bsize == clen;
		const u8 *src = in + cstart + bstart;
		u8 *out = slots + (u64)rec * stride + ZE_HDR + (u64)bi * ZE_BSTRIDE;

		ZEP(5);
		/* ------------------------------------------------ match finding + greedy parse
		 * 64 positions per step.  Software pipeline: the 8 input bytes of step t+3 and the
		 * candidate bytes of steps t+1 and t+2 are in flight while step t is parsed, so neither
		 * global round trip sits on the critical path.
		 * Table entries are positions mod 64 Ki; a candidate is rebuilt as the newest position
		 * below p with those low bits and then verified, so stale or aliased entries only cost a
		 * missed match. */
		/* 2^HLOG entries: for the larger tiers the table runs on into ZEncLdsExt::more, right behind the struct -- so
		 * the pointer is derived from the byte address of the enclosing object, never from the 4 Ki-entry member array
		 * (indexing that array past its end would be undefined) */
		u16 *const tab = (u16 *)((u8 *)&L + offsetof(ZEncLds, table));
		for (u32 i = (u32)lane; i < (1u << HLOG); i += 64)
			tab[i] = 0;
		wv_sync();
		u32 ns = 0, anchor = 0, cursor = 0;
		u32 r_ll = 0, r_ml = 0, r_of = 0;
		/* rp1: the offset of the newest sequence (0 = none yet in this unit); replive: steps the look-out for repeat matches
		 * stays open; rcarry: the offset of the last sequence that left the registers (ZE_FLUSH_SEQS).  All three are reset
		 * here, i.e. per 128 KiB block this wave encodes: the decoder's history does run on through the
		 * blocks of a unit, but Offset_Value 1 is only written when the predecessor IN THE SAME BLOCK has the same offset, so
		 * a reset costs the repeat code of a block's first sequence and nothing else */
		u32 rp1 = 0, replive = 0, rcarry = 0;
/* 64 sequences leave the registers.  Offset_Value (RFC 8878 3.1.1.3.2.1.1) is settled here, for all of them at once: only
 * the FIRST entry of the decoder's offset history is ever used -- value 1 behind literals -- so the history matters in its
 * first entry alone, and that is the offset of the sequence before, whatever value coded it (a new offset goes in front;
 * value 1 leaves the front where it is): a sequence with literals whose offset equals its predecessor's gets value 1,
 * every other one offset + 3.  A unit starts with an unknown history (rcarry = 0 equals no offset): the wave does not know
 * what the units in front of it leave behind, the previous unit may even be one Raw block.  (Values 2 and 3 -- the entries
 * behind the first -- were tried with the bookkeeping in the serial loop: 6 scalar instructions per sequence on a pipe that
 * is the kernel's bottleneck cost more time than they save bytes.)  Three or more repeats among the 64 keep the look-out for
 * repeat matches open */
#define ZE_FLUSH_SEQS(AT, LIVE)                                                                    \
	do {                                                                                       \
		u32 ov_ = r_of + 3u;                                                               \
		if (REP) {                                                                         \
			const u32 prev_ = wv_shr1(r_of, rcarry);                                   \
			const bool same_ = (LIVE) && r_of == prev_;                                \
			ov_ = (same_ && r_ll != 0u) ? 1u : ov_;                                    \
			if (wv_popc(wv_ballot(same_)) >= 3)                                        \
				replive = ZE_REPLIVE;                                              \
		}                                                                                  \
		if (LIVE) {                                                                        \
			sq_ll[(AT) + (u32)lane] = r_ll;                                            \
			sq_ml[(AT) + (u32)lane] = r_ml;                                            \
			sq_of[(AT) + (u32)lane] = ov_;                                             \
		}                                                                                  \
	} while (0)
		const u32 steps = bsize >= MM ? (bsize - MM) / 64 + 1 : 0;
/* All loads of the pipeline are unconditional (addresses clamped, results of invalid lanes ignored):
 * a load under an exec mask needs its destination initialised first, and that write would have to
 * wait for every load still in flight. */
#define ZE_LOADV(t, V)                                                                             \
	do {                                                                                       \
		const u32 p_ = (t) * 64u + (u32)lane;                                              \
		(V) = ld64u(src + (p_ < bsize ? p_ : bsize - 1));                                  \
	} while (0)
#define ZE_LOOKUP(t, V, Cc, M)                                                                     \
	do {                                                                                       \
		const u32 p_ = (t) * 64u + (u32)lane;                                              \
		const bool ok_ = (t) < steps && p_ + MM <= bsize;                                  \
		const u32 h_ = ze_hash<HB, HLOG>(V);                                               \
		const u32 e_ = ok_ ? tab[h_] : 0;                                                  \
		wv_sync();                                                                         \
		if (ok_)                                                                           \
			tab[h_] = (u16)p_;                                                         \
		wv_sync();                                                                         \
		/* equal hashes inside one step: the highest position must stay, whatever order the   \
		 * LDS served the conflicting lanes in (a step never straddles a 64 Ki boundary) */    \
		while (wv_any(wv_opaque(ok_ ? (u32)tab[h_] : 0xFFFFu) < (u32)(u16)p_)) { /* (one compare: wave.h) */ \
			if (ok_ && tab[h_] < (u16)p_)                                              \
				tab[h_] = (u16)p_;                                                 \
			wv_sync();                                                                 \
		}                                                                                  \
		u32 c_ = (p_ & ~0xFFFFu) | e_;                                                     \
		if (c_ >= p_)                                                                      \
			c_ -= 65536u;                                                              \
		/* (the four positions at the very start of a block are no candidates: the window below  \
		 * starts 4 bytes in front of one) */                                                  \
		(Cc) = (ok_ && c_ < p_ && c_ >= 4u) ? c_ : 0xFFFFFFFFu;                            \
		/* 24 bytes around the candidate -- 4 in front of it, 20 from it on -- and the same of    \
		 * the input: most matches are measured right here, without the wave-wide extension     \
		 * below, and a match grows backwards into the literals in front of it (its first bytes \
		 * often hashed to an entry that was overwritten).  One window, so that looking back    \
		 * costs no gather of its own [MI355X, 8 GiB: 92.2 ms without looking back, 108.1 ms    \
		 * with a separate 4-byte gather, 91.0 ms this way] */                                  \
		{                                                                                  \
			const bool v_ = (Cc) != 0xFFFFFFFFu;                                       \
			const u8 *cp_ = src + (v_ ? (Cc) - 4u : 0u);                               \
			const u8 *ip_ = src + ((REP ? ok_ : v_) ? p_ : 0u);                        \
			(M).v = (V);                                                               \
			(M).a = ld64u(cp_);                                                        \
			(M).b = ld64u(cp_ + 8);                                                    \
			(M).c = ld64u(cp_ + 16);                                                   \
			(M).d = ld64u(ip_ + 8);                                                    \
			(M).e = ld32u(ip_ + 16);                                                   \
			/* the input's own 4 bytes in front: the low half of the word 4 lanes down (a      \
			 * step's first four positions go without: 94.3 -> 91.0 ms against a load) */     \
			const u32 n_ = wv_shfl((u32)(V), lane - 4);                                \
			(M).pb = lane >= 4 ? n_ : ~(u32)(M).a;                                     \
		}                                                                                  \
	} while (0)
		struct Cmp {
			u64 v, a, b, c, d; /* input bytes 0..7, candidate window (bytes -4..19), input bytes 8..15 */
			u32 e, pb;         /* input bytes 16..19, input bytes -4..-1 (lanes 4..63) */
		};
		/* three register sets rotate by unrolling (copying a set would wait for its loads) */
		Cmp M[3]; /* compare data of steps t, t+1, t+2 at index step % 3 */
		u32 Cn[3];
		u64 V[3]; /* hashed input word of the step that is looked up next, same indexing */
		ZE_LOADV(0u, V[0]);
		ZE_LOADV(1u, V[1]);
		ZE_LOADV(2u, V[2]);
		ZE_LOOKUP(0u, V[0], Cn[0], M[0]);
		ZE_LOOKUP(1u, V[1], Cn[1], M[1]);
		/* repeat-offset compare data: the 16 bytes `RR` in front of every position of the NEXT step, asked for a step
		 * ahead with the newest offset known then (a load whose address waits for this step's parse would put a memory
		 * round trip on every step's chain: 91 -> 129 ms per 8 GiB [MI355X]).  Where repeat offsets pay -- records,
		 * tables, markup -- the offset stays the same from match to match and a step's delay loses nothing; where it
		 * changes with every match (the bench text) there is nothing to find either way */
		u64 RA[3] = {0, 0, 0}, RB[3] = {0, 0, 0};
		u32 RR[3] = {0, 0, 0};
		for (u32 t0 = 0; t0 < steps; t0 += 3) {
		ZMT_UNROLL
		for (int k = 0; k < 3; k++) {
			const u32 t = t0 + (u32)k;
			if (t >= steps)
				break;
			ZE_LOADV(t + 3, V[k]);
			ZE_LOOKUP(t + 2, V[(k + 2) % 3], Cn[(k + 2) % 3], M[(k + 2) % 3]);
			if (REP) {
				/* (the two loads are a third more work for the address unit than a step has without them: 94 -> 108 ms
				 * per 8 GiB when every step asks.  They are asked for while the data shows that offsets repeat -- three
				 * or more sequences at their predecessor's offset among the 64 that last left the registers keep the
				 * look-out open for ZE_REPLIVE steps (ZE_FLUSH_SEQS) -- which on the bench text is almost never and on
				 * records or markup nearly always) */
				RR[(k + 1) % 3] = 0;
				if (replive) {
					const u32 pn = (t + 1) * 64u + (u32)lane, rn = rp1;
					const u8 *rq = src + ((rn && pn >= rn && pn + MM <= bsize) ? pn - rn : 0u);
					RR[(k + 1) % 3] = rn;
					RA[(k + 1) % 3] = ld64u(rq);
					RB[(k + 1) % 3] = ld64u(rq + 8);
					replive--;
				}
			}
			ZEP(6);
			const Cmp &m0 = M[k];
			const u32 c0 = Cn[k];
			const u64 v0 = m0.v;
			const u32 p0 = t * 64u, p = p0 + (u32)lane;
			if (p0 + 64 > cursor) { /* else the whole step lies inside the previous match */
				const u32 repR = RR[k]; /* the offset this step's repeat-offset compare data was asked for with; 0 = none */
				const u64 x0 = v0 ^ (m0.a >> 32 | m0.b << 32), x1 = m0.d ^ (m0.b >> 32 | m0.c << 32);
				const u32 x2 = m0.e ^ (u32)(m0.c >> 32);
#ifdef ZE_BRANCHY
				u32 m = x0   ? (u32)__builtin_ctzll(x0) >> 3
					: x1 ? 8u + ((u32)__builtin_ctzll(x1) >> 3)
					: x2 ? 16u + ((u32)__builtin_ctz(x2) >> 3)
					     : ZE_FWD;
#else
				/* the first differing bit of the 8 + 8 + 4 bytes as a chain of minima (the three-way choice compiles to nested
				 * exec-mask regions: scalar instructions on every step of a kernel that lives on its scalar pipe; lz4_enc5.hip) */
				u32 m;
				{
					const u32 z0 = x0 ? (u32)__builtin_ctzll(x0) : 255u; /* (255: "none here", above every sum below) */
					const u32 z1 = x1 ? (u32)__builtin_ctzll(x1) : 255u;
					const u32 z2 = x2 ? (u32)__builtin_ctz(x2) : 32u;
					const u32 z12 = 64u + (z1 < 64u + z2 ? z1 : 64u + z2);
					m = (z0 < z12 ? z0 : z12) >> 3; /* 8 + 8 + 4 equal bytes: 160 >> 3 = ZE_FWD */
				}
#endif
				const bool cand = c0 != 0xFFFFFFFFu && p >= cursor;
				if (cand && m > bsize - p)
					m = bsize - p;
				const u32 xb = m0.pb ^ (u32)m0.a;
				const u32 back = !cand ? 0u : xb ? (u32)__builtin_clz(xb) >> 3 : 4u; /* equal bytes right in front */
				/* what this position offers, settled per lane so that the serial loop below reads it with three
				 * v_readlane and nothing else: the repeat offset costs a few bits where a new one costs its logarithm, so
				 * it wins unless the table's candidate is clearly longer.  mlx = length | "measured to the end of what was
				 * loaded: the wave extends it" << 31 */
				const bool hc = cand && m >= MM;
				bool ur = false;
				u32 mlx = m | (m == ZE_FWD ? 0x80000000u : 0u), cj_v = c0, bk_v = back;
				/* the same bytes at the newest offset (compare data of a step ago, see RA / RB); a wave-uniform branch:
				 * a step without a look-out pays nothing for it.  (An offset that is no longer the newest would cost as
				 * much as a new one: not looked at) */
				if (REP && repR && repR == rp1) {
					const bool rv = p >= repR && p + MM <= bsize && p >= cursor;
					const u64 y0 = v0 ^ RA[k], y1 = m0.d ^ RB[k];
					u32 mr = y0 ? (u32)__builtin_ctzll(y0) >> 3 : y1 ? 8u + ((u32)__builtin_ctzll(y1) >> 3) : 16u;
					const bool full = mr == 16u;
					mr = mr < bsize - p ? mr : bsize - p; /* (a match ends with its block) */
					ur = rv && mr >= MM && (!hc || mr + ZE_REPBONUS >= m);
					mlx = ur ? (mr | (full ? 0x80000000u : 0u)) : mlx;
					cj_v = ur ? p - repR : cj_v;
					bk_v = ur ? 0u : bk_v;
				}
#ifdef ZE_BRANCHY
				u64 mask = wv_ballot(hc || ur);
#else
				/* (ballots of one compare each, combined as scalar masks: a ballot of `a && b` compiles to mask algebra plus a
				 * select and a second compare that re-materialise the mask) */
				u64 mask = wv_ballot(m >= MM) & wv_ballot(c0 != 0xFFFFFFFFu) & wv_ballot(p >= cursor);
				if (REP)
					mask |= wv_ballot(ur);
#endif
				/* look-ahead (every lane's length is measured anyway), for all positions of the step at once: a match
				 * that starts d <= LAZYW bytes further on wins when it is longer by more than the d literals it adds
				 * -- bytes it reaches backwards over this position's side count for it.  Such a position is skipped;
				 * the loop comes to that match (or to a better one in front of it) by itself.  Lane j reads lanes
				 * j + 1 .. j + LAZYW through wave_shl:1 steps of one packed word (length | backward bytes << 8) */
				if (LAZYW) {
					u32 w = hc ? (m | back << 8) : 0u;
#ifdef ZE_BRANCHY
					bool lz = false;
					ZMT_UNROLL
					for (u32 d = 1; d <= LAZYW; d++) {
						w = wv_shl1(w, 0u);
						const u32 bq = (w >> 8) < d ? (w >> 8) : d;
						lz = lz || (w != 0u && (w & 255u) + bq >= m + d + 1u);
					}
					mask &= ~wv_ballot(lz && m < ZE_FWD && !ur);
#else
					/* the best offer of the LAZYW positions behind as one signed maximum of (length + reach - d) -- a position
					 * without a match offers -d -- and ONE compare against what this position needs to be beaten */
					int best = -1;
					ZMT_UNROLL
					for (u32 d = 1; d <= LAZYW; d++) {
						w = wv_shl1(w, 0u);
						const u32 bq = (w >> 8) < d ? (w >> 8) : d;
						const int offer = (int)((w & 255u) + bq) - (int)d;
						best = offer > best ? offer : best;
					}
					const u32 need = (m < ZE_FWD && !ur) ? m + 1u : 0x7FFFFFFFu;
					mask &= ~wv_ballot(best >= (int)wv_opaque(need));
#endif
				}
				ZEP(7);
				/* (Choosing the step's matches in vector code -- the chain of "first candidate at or behind this match's
				 * end" per lane, one v_readlane per chosen match, literal lengths from a prefix maximum of the chosen ends,
				 * three ds_bpermute into the collecting registers -- was built in round 5, byte-identical on 48 inputs, and
				 * measured: 95.64 against 95.82 ms per 8 GiB.  The loop is not what a step waits for: its ~40 scalar
				 * instructions per match run beside the other 15 waves' work, while the vector version puts a scan, three
				 * bpermutes and the hops on the wave's own dependent chain.  Removed again;
				 * profiles/r05_sweeps/zstd_enc_steps.txt) */
				while (mask) {
					const int j = wv_ffs(mask) - 1;
					mask &= mask - 1;
					const u32 pj = p0 + (u32)j;
					if (__builtin_expect(pj < cursor, 0))
						continue;
					const u32 mlw = wv_readlane(mlx, j);
					u32 ml = mlw & 0x7FFFFFFFu;
					const u32 cj = wv_readlane(cj_v, j);
					const u32 mcap = ml;
					if (__builtin_expect((mlw >> 31) != 0, 0)) {
						const u64 tx_ = ZET();
						/* extend: 64 lanes x 8 bytes per step */
						for (u32 base = mcap;; base += 512) {
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
						if (PROF) {
							const u64 t_ = ZET();
							pc[PROF ? 5 : 0] += t_ - tx_;
							tq += t_ - tx_; /* not charged to the parse phase */
						}
					}
					/* sequences collect in registers (lane = index mod 64) and leave 64 at a time */
					{
						u32 bk = wv_readlane(bk_v, j); /* as far back as the literals since the last match reach */
						bk = bk < pj - anchor ? bk : pj - anchor;
						const u32 llj = pj - bk - anchor, ofj = pj - cj;
						rp1 = ofj; /* the newest offset: what the next repeat-offset compare data are asked for with */
						const bool me = (u32)lane == (ns & 63);
						r_ll = me ? llj : r_ll;
						r_ml = me ? ml + bk : r_ml;
						r_of = me ? ofj : r_of;
					}
					ns++;
					if ((ns & 63) == 0) {
						ZE_FLUSH_SEQS(ns - 64, true);
						rcarry = wv_readlane(r_of, 63);
					}
					anchor = cursor = pj + ml;
					/* drop every candidate the match covers in one go */
					mask = cursor - p0 >= 64 ? 0 : mask & ~((1ull << (cursor - p0)) - 1);
				}
			}
			ZEP(0);
		}
		}
#undef ZE_LOADV
#undef ZE_LOOKUP
		ZE_FLUSH_SEQS(ns & ~63u, (u32)lane < (ns & 63)); /* the sequences still in registers */
#undef ZE_FLUSH_SEQS
		wave_mem_fence();
#ifdef ZMT_EMU
		if (getenv("ZMT_EMU_DEBUG") && lane == 0) {
			u32 hsum = 0;
			for (u32 i = 0; i < ns; i++)
				hsum = hsum * 31 + sq_ll[i] * 7 + sq_ml[i] * 3 + sq_of[i];
			fprintf(stderr, "zstd_enc: block %u bsize %u ns %u seqhash %08x\n", g, bsize, ns, hsum);
			if (getenv("ZMT_EMU_DEBUG")[0] == '2') {
				u32 pos = 0;
				for (u32 i = 0; i < ns && i < 40000; i++) {
					fprintf(stderr, "S %u %u %u %u %u\n", g, pos + sq_ll[i], sq_ll[i], sq_ml[i], sq_of[i]);
					pos += sq_ll[i] + sq_ml[i];
				}
			}
		}
#endif

		ZEP(0);
		/* ------------------------------------------------ sequence tables of the unit
		 * (see ze_normalize): histogram of the three codes by all lanes, then lanes 0 / 1 / 2
		 * fit, build and describe the LL / OF / ML table.  The hash table is idle from here on;
		 * the bit-packing stage of the literals is not in use yet and lends its first KiB. */
		bool fitted = false;
		u32 desc_len = 0;
		{
			u32 *shist = L.bitstage;                    /* LL at 0, ML at 36, OF at 89 */
			short *snorm = (short *)(L.bitstage + 128); /* same layout */
			u8 *sscr = (u8 *)(L.bitstage + 192);        /* 64 bytes per table: spread scratch, then its description */
			const int t_off = lane == 0 ? 0 : lane == 1 ? 89 : 36;
			const int t_alpha = lane == 0 ? 36 : lane == 1 ? 29 : 53, t_log = lane == 0 ? ZE_LL_LOG : lane == 1 ? ZE_OF_LOG : ZE_ML_LOG;
			u16 *t_st = lane == 0 ? L.st_ll : lane == 1 ? L.st_of : L.st_ml;
			u32(*t_tt)[2] = lane == 0 ? L.tt_ll : lane == 1 ? L.tt_of : L.tt_ml;
			if (ns >= ZE_ADAPT_MIN) {
				for (u32 i = (u32)lane; i < 128; i += 64)
					shist[i] = 0;
				wv_sync();
				for (u32 i = (u32)lane; i < ns; i += 64) {
					const u32 ll = sq_ll[i], mlb = sq_ml[i] - 3, ofv = sq_of[i];
					atomicAdd(&shist[ll < 64 ? L.llcode[ll] : (u32)hb32(ll) + 19], 1u);
					atomicAdd(&shist[36 + (mlb < 128 ? L.mlcode[mlb] : (u32)hb32(mlb) + 36)], 1u);
					atomicAdd(&shist[89 + (u32)hb32(ofv)], 1u);
				}
				wv_sync();
				u32 present = 2;
				if (lane < 3)
					present = ze_normalize(shist + t_off, t_alpha, ns, t_log, snorm + t_off);
				/* a table with a single symbol would be RLE mode: such a unit keeps the predefined ones */
				fitted = !wv_any(present < 2);
			}
			if (fitted) {
				u32 dl = 0;
				if (lane < 3) {
					int nsym = t_alpha;
					while (!snorm[t_off + nsym - 1])
						nsym--;
					fse_ctable(t_st, t_tt, snorm + t_off, nsym, t_log, sscr + 64 * lane);
					for (int s2 = nsym; s2 < t_alpha; s2++) { /* symbols behind the last present one */
						t_tt[s2][0] = ((u32)(t_log + 1) << 16) - (1u << t_log);
						t_tt[s2][1] = 0;
					}
					dl = ze_write_ncount(sscr + 64 * lane, snorm + t_off, nsym, t_log);
				}
				wv_sync();
				const u32 d_ll = wv_readlane(dl, 0), d_of = wv_readlane(dl, 1), d_ml = wv_readlane(dl, 2);
				desc_len = d_ll + d_of + d_ml;
				/* LL, OF, ML descriptions end where run 0's bitstream starts */
				for (u32 i = (u32)lane; i < desc_len; i += 64)
					bstmp[ZE_DESC - desc_len + i] = i < d_ll          ? sscr[i]
									: i < d_ll + d_of ? sscr[64 + i - d_ll]
											  : sscr[128 + i - d_ll - d_of];
				tabs_pre = false;
			} else if (!tabs_pre) {
				if (lane == 0)
					fse_ctable(L.st_ll, L.tt_ll, ZE_LL_DEF, 36, ZE_LL_LOG, sscr);
				else if (lane == 1)
					fse_ctable(L.st_of, L.tt_of, ZE_OF_DEF, 29, ZE_OF_LOG, sscr + 64);
				else if (lane == 2)
					fse_ctable(L.st_ml, L.tt_ml, ZE_ML_DEF, 53, ZE_ML_LOG, sscr + 128);
				tabs_pre = true;
			}
			wv_sync();
		}
		/* ------------------------------------------------ sequences -> bitstreams
		 * The sequence list is cut into G runs of equal count; each run becomes its own zstd
		 * block, so G lanes can FSE-code side by side (an FSE stream is a serial chain; a block
		 * boundary costs 8 bytes).  Lane s codes run s into bstmp + s * ZE_BSTMP + ZE_DESC. */
		u32 total = 0; /* bytes written at out; 0 = store the block raw */
		if (ns) {
			const u32 G = ns < 512 ? 1u : (ns / 512 < ZE_G ? ns / 512 : ZE_G);
			if ((u32)lane < ZE_G) {
				L.sb_lo[lane] = (u32)(((u64)lane * ns) / G);
				L.sb_hi[lane] = (u32)lane < G ? (u32)(((u64)(lane + 1) * ns) / G) : L.sb_lo[lane];
			}
			wv_sync();
			BitW w;
			w.p = bstmp + (u32)lane * ZE_BSTMP + ZE_DESC;
			w.limit = w.p + ZE_BSTMP - ZE_DESC;
			w.acc = 0;
			w.nb = 0;
			w.ovf = false;
			u32 s_ll = 0, s_ml = 0, s_of = 0;
			bool first = true;
			const bool coder = (u32)lane < G;
			for (;;) {
				/* stage: up to 8 sequences per run, taken from the end of each run backwards */
				u32 any = 0;
				for (u32 e = (u32)lane; e < ZE_G * 8; e += 64) {
					const u32 sb = e >> 3, k = e & 7;
					const u32 lo = L.sb_lo[sb], hi = L.sb_hi[sb];
					if (sb < G && k < hi - lo) {
						const u32 i = hi - 1 - k;
						L.stage[sb][k][0] = sq_ll[i];
						L.stage[sb][k][1] = sq_ml[i];
						L.stage[sb][k][2] = sq_of[i];
						any = 1;
					}
				}
				if (!wv_any(any != 0))
					break;
				wv_sync();
				if (coder) {
					const u32 lo = L.sb_lo[lane], hi = L.sb_hi[lane];
					const u32 cnt = hi - lo < 8 ? hi - lo : 8;
					for (u32 k = 0; k < cnt; k++) {
						const u32 ll = L.stage[lane][k][0], mlb = L.stage[lane][k][1] - 3;
						const u32 ofv = L.stage[lane][k][2]; /* Offset_Value */
						const u32 lc = ll < 64 ? L.llcode[ll] : (u32)hb32(ll) + 19;
						const u32 mc = mlb < 128 ? L.mlcode[mlb] : (u32)hb32(mlb) + 36;
						const u32 oc = (u32)hb32(ofv);
						u32 sbv = 0, sbn = 0; /* state bits of this sequence: value, count (none in front of a run's first) */
						if (first) {
							/* FSE_initCState2 x3: ML, OF, LL */
							u32 nbo = (L.tt_ml[mc][0] + (1u << 15)) >> 16;
							s_ml = L.st_ml[(((nbo << 16) - L.tt_ml[mc][0]) >> nbo) + L.tt_ml[mc][1]];
							nbo = (L.tt_of[oc][0] + (1u << 15)) >> 16;
							s_of = L.st_of[(((nbo << 16) - L.tt_of[oc][0]) >> nbo) + L.tt_of[oc][1]];
							nbo = (L.tt_ll[lc][0] + (1u << 15)) >> 16;
							s_ll = L.st_ll[(((nbo << 16) - L.tt_ll[lc][0]) >> nbo) + L.tt_ll[lc][1]];
							first = false;
						} else {
							/* the three state transitions are independent: their bits -- OF, ML, LL state bits, in
							 * the stream's order -- go out as one field (at most ZE_OF_LOG + ZE_ML_LOG + ZE_LL_LOG = 17 bits) */
							const u32 n_of = (s_of + L.tt_of[oc][0]) >> 16, n_ml = (s_ml + L.tt_ml[mc][0]) >> 16;
							const u32 n_ll = (s_ll + L.tt_ll[lc][0]) >> 16;
							sbv = (s_of & ((1u << n_of) - 1u)) | (s_ml & ((1u << n_ml) - 1u)) << n_of |
							      (s_ll & ((1u << n_ll) - 1u)) << (n_of + n_ml);
							sbn = n_of + n_ml + n_ll;
							s_of = L.st_of[(s_of >> n_of) + L.tt_of[oc][1]];
							s_ml = L.st_ml[(s_ml >> n_ml) + L.tt_ml[mc][1]];
							s_ll = L.st_ll[(s_ll >> n_ll) + L.tt_ll[lc][1]];
						}
						{
							/* ... followed by the extra bits of LL and ML in the same field when that stays below 32 bits
							 * (nearly always: runs of 64 Ki have 16 extra bits each), then the offset's.  Two steps of
							 * the bit writer per sequence instead of six */
							const u32 nl = L.llx[lc] >> 24, nm = L.mlx[mc] >> 24;
							const u32 lv = ll - (L.llx[lc] & 0xFFFFFFu), mv = mlb + 3 - (L.mlx[mc] & 0xFFFFFFu);
							if (sbn + nl + nm <= 31u) {
								bw_add(w, sbv | lv << sbn | mv << (sbn + nl), sbn + nl + nm);
							} else {
								bw_add(w, sbv, sbn);
								bw_add(w, lv, nl);
								bw_add(w, mv, nm);
							}
							bw_add(w, ofv - (1u << oc), oc);
						}
					}
					L.sb_hi[lane] = hi - cnt;
				}
				wv_sync();
			}
			if (coder) {
				bw_add(w, s_ml, ZE_ML_LOG);
				bw_add(w, s_of, ZE_OF_LOG);
				bw_add(w, s_ll, ZE_LL_LOG);
				bw_add(w, 1, 1); /* end mark */
				const u32 tail = (w.nb + 7) >> 3;
				if (w.p + tail > w.limit) {
					w.ovf = true;
				} else {
					for (u32 t = 0; t < tail; t++)
						w.p[t] = (u8)(w.acc >> (8 * t));
				}
				w.p += tail;
				L.sb_bits[lane] = w.ovf ? 0xFFFFFFFFu : (u32)(w.p - (bstmp + (u32)lane * ZE_BSTMP + ZE_DESC));
			}
			wave_mem_fence();
			wv_sync();

			ZEP(1);
			/* -------------------------------------------- literals of the whole unit -> litbuf
			 * (lane per sequence, long runs by the whole wave), then one Huffman code for all of
			 * them: the first block that uses it carries the tree, the others are treeless */
			u32 lit_total = 0;
			{
				u32 ip = 0, lpos = 0;
				/* 128 sequences per pass, two per lane (the loads of both before the stores of either); the lengths of the next
				 * pass are asked for before this pass's copies */
				u32 n_ll0 = (u32)lane < ns ? sq_ll[lane] : 0, n_ml0 = (u32)lane < ns ? sq_ml[lane] : 0;
				u32 n_ll1 = 64u + (u32)lane < ns ? sq_ll[64u + (u32)lane] : 0, n_ml1 = 64u + (u32)lane < ns ? sq_ml[64u + (u32)lane] : 0;
				for (u32 b = 0; b < ns; b += 128) {
					const u32 ll0 = n_ll0, ml0 = n_ml0, ll1 = n_ll1, ml1 = n_ml1;
					{
						const u32 i2 = b + 128u + (u32)lane, i3 = i2 + 64u;
						n_ll0 = i2 < ns ? sq_ll[i2] : 0;
						n_ml0 = i2 < ns ? sq_ml[i2] : 0;
						n_ll1 = i3 < ns ? sq_ll[i3] : 0;
						n_ml1 = i3 < ns ? sq_ml[i3] : 0;
					}
					const u32 incl0 = wv_scan_incl(ll0 + ml0), lincl0 = wv_scan_incl(ll0);
					const u32 ip1 = ip + wv_readlane(incl0, 63), lpos1 = lpos + wv_readlane(lincl0, 63);
					const u32 incl1 = wv_scan_incl(ll1 + ml1), lincl1 = wv_scan_incl(ll1);
					const u32 s0 = ip + incl0 - ll0 - ml0, d0 = lpos + lincl0 - ll0;
					const u32 s1 = ip1 + incl1 - ll1 - ml1, d1 = lpos1 + lincl1 - ll1;
					const u32 c0 = ll0 <= ZE_CAP ? ll0 : 0u, c1 = ll1 <= ZE_CAP ? ll1 : 0u;
					ZeLits q0, q1;
					ze_lits_load(q0, src + s0, c0); /* (every lane: the routines vote) */
					ze_lits_load(q1, src + s1, c1);
					ze_lits_store(q0, litbuf + d0, c0);
					ze_lits_store(q1, litbuf + d1, c1);
					u64 lm = wv_ballot(ll0 > ZE_CAP);
					while (lm) {
						const int j = wv_ffs(lm) - 1;
						lm &= lm - 1;
						wave_copy(litbuf + wv_readlane(d0, j), src + wv_readlane(s0, j), wv_readlane(ll0, j), lane);
					}
					lm = wv_ballot(ll1 > ZE_CAP);
					while (lm) {
						const int j = wv_ffs(lm) - 1;
						lm &= lm - 1;
						wave_copy(litbuf + wv_readlane(d1, j), src + wv_readlane(s1, j), wv_readlane(ll1, j), lane);
					}
					ip = ip1 + wv_readlane(incl1, 63);
					lpos = lpos1 + wv_readlane(lincl1, 63);
				}
				wave_copy(litbuf + lpos, src + ip, bsize - ip, lane); /* after the last match */
				lit_total = lpos + (bsize - ip);
			}
			wave_mem_fence();
			ZEP(2);
			const ZHuf hf = ze_huf_build(L, litbuf, lit_total, lane);
			bool tree_sent = false;
			ZEP(3);
			/* -------------------------------------------- assemble the G blocks at out */
			u32 ipos = 0, lbase = 0, at = 0; /* input / literal bytes consumed, bytes written */
			bool fits = true;
			for (u32 sb = 0; sb < G && fits; sb++) {
				const u32 lo = (u32)(((u64)sb * ns) / G), hi = (u32)(((u64)(sb + 1) * ns) / G);
				const u32 nsq = hi - lo, bits = L.sb_bits[sb];
				const u32 dsc = fitted && sb == 0 ? desc_len : 0; /* the unit's table descriptions */
				/* sizes of this run: literal bytes, input bytes */
				u32 lsum = 0, isum = 0;
				for (u32 b = lo; b < hi; b += 64) {
					const u32 i = b + (u32)lane;
					const u32 ll = i < hi ? sq_ll[i] : 0, ml = i < hi ? sq_ml[i] : 0;
					lsum += wv_readlane(wv_scan_incl(ll), 63);
					isum += wv_readlane(wv_scan_incl(ll + ml), 63);
				}
				const bool fin = sb + 1 == G;
				const u32 trail = fin ? bsize - (ipos + isum) : 0;
				const u32 regen = lsum + trail;
				const u32 lh = regen < 32 ? 1u : regen < 4096 ? 2u : 3u;
				const u32 sh = nsq < 128 ? 1u : nsq < 0x7F00 ? 2u : 3u;
				/* room check with the literals raw (coding them only shrinks the block) */
				if (bits == 0xFFFFFFFFu || at + 3 + lh + regen + sh + 1 + dsc + bits + 8 > bsize) {
					fits = false;
					break;
				}
				u8 *o = out + at;
				const u8 *lit = litbuf + lbase;
				/* literals section: Huffman-coded when that pays, else raw */
				u32 lsec = ze_huf_encode(L, L.bitstage, hf, lit, regen, o + 3, !tree_sent, lane);
				if (lsec) {
					tree_sent = true;
				} else {
					if (lane == 0) {
						const u32 hv = lh == 1 ? regen << 3 : regen << 4 | (lh == 2 ? 1u : 3u) << 2;
						for (u32 k = 0; k < lh; k++)
							o[3 + k] = (u8)(hv >> (8 * k));
					}
					wave_copy(o + 3 + lh, lit, regen, lane);
					lsec = lh + regen;
				}
				ZEP(3);
				const u32 csize = lsec + sh + 1 + dsc + bits;
				if (lane == 0) {
					const u32 bh = (fin ? last : 0u) | 2u << 1 | csize << 3;
					o[0] = (u8)bh;
					o[1] = (u8)(bh >> 8);
					o[2] = (u8)(bh >> 16);
				}
				u8 *sp = o + 3 + lsec;
				if (lane == 0) {
					if (sh == 1) {
						sp[0] = (u8)nsq;
					} else if (sh == 2) {
						sp[0] = (u8)((nsq >> 8) + 128);
						sp[1] = (u8)nsq;
					} else {
						sp[0] = 255;
						sp[1] = (u8)(nsq - 0x7F00);
						sp[2] = (u8)((nsq - 0x7F00) >> 8);
					}
					/* Symbol_Compression_Modes: three predefined tables, or the unit's own: Compressed
					 * in its first block, Repeat in the others */
					sp[sh] = !fitted ? 0 : sb == 0 ? 0xA8 : 0xFC;
				}
				wave_copy(sp + sh + 1, bstmp + sb * ZE_BSTMP + ZE_DESC - dsc, dsc + bits, lane);
				at += 3 + csize;
				ipos += isum;
				lbase += regen;
				ZEP(4);
			}
			if (fits)
				total = at;
		}
		if (total) {
			if (lane == 0)
				blk_len[g] = total;
		} else {
			wv_sync(); /* every lane's stores of the attempt lie behind it before the same bytes are rewritten */
			if (lane == 0) {
				const u32 bh = last | 0u << 1 | bsize << 3;
				out[0] = (u8)bh;
				out[1] = (u8)(bh >> 8);
				out[2] = (u8)(bh >> 16);
				blk_len[g] = 3 + bsize;
			}
			wave_copy(out + 3, src, bsize, lane);
		}
		wave_mem_fence();
		ZEP(4);
	}
#ifndef ZMT_EMU
	if (PROF && prof && lane == 0) {
		for (int i = 0; i < (PROF ? 8 : 1); i++)
			atomicAdd(prof + i, (unsigned long long)pc[i]);
		atomicAdd(prof + 8, (unsigned long long)(ZET() - t_begin));
		atomicAdd(prof + 9, 1ull);
	}
#endif
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
zmt_zstd_enc_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nblk_total, u32 blk_per_rec,
		    u8 *__restrict__ slots, u64 stride, u32 *__restrict__ blk_len, u8 *__restrict__ scratch)
{
	__shared__ __attribute__((aligned(16))) ZEncLds L;
	zstd_enc_body<false, ZE_HBYTES, ZE_MINMATCH, ZE_HLOG>(L, in, n, chunk, nblk_total, blk_per_rec, slots, stride, blk_len,
							       scratch, nullptr);
}
/* tiers 2 and 3 (levels 3..9 and 10..22): 6 bytes hashed, minimum match 6, 8 Ki / 16 Ki table entries */
extern "C" __global__ void __launch_bounds__(64)
zmt_zstd_enc_t2_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nblk_total, u32 blk_per_rec,
		       u8 *__restrict__ slots, u64 stride, u32 *__restrict__ blk_len, u8 *__restrict__ scratch)
{
	__shared__ __attribute__((aligned(16))) ZEncLdsExt<13> S;
	zstd_enc_body<false, 6, 6u, 13>(S.L, in, n, chunk, nblk_total, blk_per_rec, slots, stride, blk_len, scratch, nullptr);
}
extern "C" __global__ void __launch_bounds__(64)
zmt_zstd_enc_t3_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nblk_total, u32 blk_per_rec,
		       u8 *__restrict__ slots, u64 stride, u32 *__restrict__ blk_len, u8 *__restrict__ scratch)
{
	__shared__ __attribute__((aligned(16))) ZEncLdsExt<14> S;
	zstd_enc_body<false, 6, 6u, 14>(S.L, in, n, chunk, nblk_total, blk_per_rec, slots, stride, blk_len, scratch, nullptr);
}

#ifndef ZMT_EMU
/* same kernel with per-phase cycle counters (developer tool) */
extern "C" __global__ void __launch_bounds__(64)
zmt_zstd_enc_kernel_prof(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nblk_total, u32 blk_per_rec,
			 u8 *__restrict__ slots, u64 stride, u32 *__restrict__ blk_len, u8 *__restrict__ scratch,
			 unsigned long long *prof)
{
	__shared__ __attribute__((aligned(16))) ZEncLds L;
	zstd_enc_body<true, ZE_HBYTES, ZE_MINMATCH, ZE_HLOG>(L, in, n, chunk, nblk_total, blk_per_rec, slots, stride, blk_len,
							      scratch, prof);
}
#endif

/* Record slot -> finished record: skippable header (lib/zstd-mt_compress.c:296-302), frame header
 * (magic, single-segment descriptor, content size), the chunk's blocks moved together. */
extern "C" __global__ void __launch_bounds__(256)
zmt_zstd_assemble_kernel(u64 n, u32 chunk, u32 nrec, u32 blk_per_rec, u8 *__restrict__ slots, u64 stride,
			 const u32 *__restrict__ blk_len, u32 *__restrict__ rec_len)
{
	const u32 rec = blockIdx.x, t = threadIdx.x;
	if (rec >= nrec)
		return;
	const u64 cstart = (u64)rec * chunk;
	const u32 clen = (u32)(n - cstart < chunk ? n - cstart : chunk);
	u8 *slot = slots + (u64)rec * stride;
	const u32 fcs_len = clen < 256 ? 1u : clen < 65536 + 256 ? 2u : 4u;
	/* A single-segment header makes Window_Size = content size; libzstd's streaming decoder (what the
	 * reference's pt_decompress uses, lib/zstd-mt_decompress.c:464) refuses windows above 2^27.  No
	 * match of this encoder leaves its 128 KiB block, so chunks beyond 128 MiB get a Window_Descriptor
	 * of 128 KiB and a 4-byte content size instead. */
	const bool with_wd = clen > (128u << 20);
	const u32 fh = 4 + 1 + fcs_len + (with_wd ? 1u : 0u);
	u32 at = 12 + fh;
	const u32 nb = clen ? (clen + ZE_BLOCK - 1) / ZE_BLOCK : 0;
	for (u32 b = 0; b < nb; b++) {
		const u32 len = blk_len[(u64)rec * blk_per_rec + b];
		const u8 *s = slot + ZE_HDR + (u64)b * ZE_BSTRIDE;
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
		u8 *f = slot + 12;
		if (nb == 0) {
			/* empty chunk: one empty raw block marked last (the 9-byte frame libzstd writes too) */
			slot[at] = 1;
			slot[at + 1] = 0;
			slot[at + 2] = 0;
			at += 3;
		}
		const u32 csz = at - 12;
		slot[0] = 0x50; slot[1] = 0x2A; slot[2] = 0x4D; slot[3] = 0x18;
		slot[4] = 4; slot[5] = 0; slot[6] = 0; slot[7] = 0;
		slot[8] = (u8)csz; slot[9] = (u8)(csz >> 8); slot[10] = (u8)(csz >> 16); slot[11] = (u8)(csz >> 24);
		f[0] = 0x28; f[1] = 0xB5; f[2] = 0x2F; f[3] = 0xFD;
		if (fcs_len == 1) {
			f[4] = 0x20; /* single segment, 1-byte content size */
			f[5] = (u8)clen;
		} else if (fcs_len == 2) {
			f[4] = 0x20 | 1u << 6;
			f[5] = (u8)(clen - 256);
			f[6] = (u8)((clen - 256) >> 8);
		} else if (!with_wd) {
			f[4] = 0x20 | 2u << 6;
			f[5] = (u8)clen; f[6] = (u8)(clen >> 8); f[7] = (u8)(clen >> 16); f[8] = (u8)(clen >> 24);
		} else {
			f[4] = 2u << 6; /* 4-byte content size, Window_Descriptor follows */
			f[5] = 0x38;    /* exponent 7, mantissa 0: 2^17 */
			f[6] = (u8)clen; f[7] = (u8)(clen >> 8); f[8] = (u8)(clen >> 16); f[9] = (u8)(clen >> 24);
		}
		rec_len[rec] = at;
	}
}
