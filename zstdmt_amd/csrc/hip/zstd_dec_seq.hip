/*
 * zstd_dec_seq.hip -- sequence pre-pass of the zstd frame decoder: the FSE sequence bitstreams of a frame's blocks,
 * decoded side by side.
 *
 * Why: in a frame written by the reference (ZSTD_compress per chunk, /root/reference/lib/zstd-mt_compress.c:285) every
 * block carries its own fitted FSE tables, and the walk of a block's sequence bitstream is one serial chain (state ->
 * cell -> bits -> state).  zstd_dec.hip walks it on one lane triple per wave: two thirds of a frame's decode time, and
 * the CU issues the walk's instructions for one sequence at a time.  The bitstreams of different blocks are
 * independent, though -- what a block's sequences need from its predecessors (repeat offsets, the output they copy
 * from) only matters when they are EXECUTED.  So this kernel, launched before the frame decoder, takes up to ZS_NB
 * blocks of a frame at a time, builds their tables into per-block LDS slots (2.5 KiB each with 16-bit cells; with the
 * rest 26 KiB per wave, which is why it is a kernel of its own: 6 waves per CU here, 16 in the decoder) and walks the
 * ZS_NB bitstreams in lockstep, four lanes per block (LL / OF / ML state + one idle): one pass of the loop decodes
 * one sequence of every block.  [MI355X] 8 GiB of reference-written level-1 frames: 21.3 ms for the 601 M sequences,
 * then 30.0 ms in the frame decoder (125.9 ms without this kernel).  Sequences go to a
 * per-record region of `seqbuf` as ll | ml << 18 | offset value << 36 (the format of the decoder's own unit path);
 * a header at the head of the region says for every block of the frame where its sequences start.  The decoder then
 * skips table building and the walk for those blocks and only executes.
 *
 * Verdicts stay the decoder's: anything odd (a table that does not parse, a bitstream that ends early or late, an
 * offset code beyond the packing, a region that is full) leaves the frame, or the rest of it, to the serial path of
 * zstd_dec.hip, which then judges the block exactly as before.  A block is marked only when its tables, its
 * bitstream and its end-of-stream condition were clean, i.e. when the serial walk would have produced the same
 * sequences.  The region of a record is carved out of a buffer as large as the batch's output: [out_off, out_off +
 * out_len) of `seqbuf`, so a frame has room for one sequence per 8 bytes of content (the reference's level-1 streams
 * of text hold one per 14); denser blocks are simply not marked.
 *
 * RFC 8878 3.1.1.3.2 (sequences section), 4.1 (FSE); reference call replaced: ZSTD_decompressStream,
 * /root/reference/lib/zstd-mt_decompress.c:464.
 */
#include "zstd_dec_common.h"
#include "zstd_dec_seq.h"

#define ZS_PB 2u /* blocks whose tables are read and built at a time */
#define ZS_WSTRIDE 288u
/* a block's bitstream window in its stage row, and the bits that may lie behind the cursor before it is refilled (a refill is a memory
 * round trip in front of the steps, with six waves per CU to hide it: at 160 bytes / 544 bits -- rounds 4-5 -- there were 2.4 x as
 * many; the row is 288 bytes either way, it holds the 256-byte table descriptions first) */
#ifndef ZS_WIN
#define ZS_WIN 256u
#define ZS_REFILL 1320u
#endif
static_assert(ZS_WIN + 20u <= ZS_WSTRIDE + 0u && ZS_WIN % 32u == 0, "the window and its 16-byte reads stay inside the stage row");
struct ZSeqLds {
	u16 ll[ZS_NB][512], of[ZS_NB][256], ml[ZS_NB][512]; /* FSE cells (16 bits, see zs_build16), slot s of every kind */
	u16 pre_ll[64], pre_of[32], pre_ml[64];               /* the predefined tables, built once */
	u8 below[16];
	u8 stage[ZS_NB][ZS_WSTRIDE]; /* table descriptions of the group's blocks (256 bytes); afterwards their 256-byte bitstream
	                              * windows.  The stride puts block g's window 8 banks behind block g - 1's: the eight blocks
	                              * read five dwords each at nearly the same offset of their windows (their streams advance at
	                              * the same pace), and a stride of 256 bytes is all 64 banks */
	u8 above[32];
	short norm[ZS_PB][3][64]; /* tables are read and built ZS_PB blocks at a time: LDS for six waves per CU */
	u16 next[ZS_PB][3][64];
	u32 llx[36], mlx[53];
	u32 valx[4][64]; /* value base of code `sym` for the lane kinds LL / OF / ML / idle: one branch-free read per step */
	u32 blk[ZS_NB + 1][8]; /* per block of the group, see ZB_* (one more row: the group may end at its last block) */
	u32 spec[ZS_NB][4];    /* what the block says about its table of kind t: ZS_PRE / ZS_RLE | symbol / ZS_REP / nsym | log << 8 */
	u32 tp[ZS_NB][4];      /* the table of kind t block g decodes with: cell offset from L.ll[0][0] | log << 24; ~0 = none */
};
enum { ZB_DESC = 0, ZB_END, ZB_NSEQ, ZB_MODES, ZB_BI, ZB_HDR, ZB_ERR };

/* 16 LDS bytes at any address as four dwords, from five ALIGNED dword reads + funnel shifts (a misaligned LDS read costs
 * the LDS pipe a cycle per active lane) */
static __device__ __forceinline__ void zs_ld128(const u8 *p, u32 &d0, u32 &d1, u32 &d2, u32 &d3)
{
#ifdef ZMT_EMU
	d0 = ld32u(p);
	d1 = ld32u(p + 4);
	d2 = ld32u(p + 8);
	d3 = ld32u(p + 12);
#else
	const u32 a = (u32)(size_t)(const __attribute__((address_space(3))) u8 *)p;
	const __attribute__((address_space(3))) u32 *d = (const __attribute__((address_space(3))) u32 *)(size_t)(a & ~3u);
	const u32 e0 = d[0], e1 = d[1], e2 = d[2], e3 = d[3], e4 = d[4];
	d0 = wv_alignbyte(e1, e0, a);
	d1 = wv_alignbyte(e2, e1, a);
	d2 = wv_alignbyte(e3, e2, a);
	d3 = wv_alignbyte(e4, e3, a);
#endif
}
/* `width` (<= 31) bits that start `topoff` bits below the top of the 128-bit window d3:d2:d1:d0: the two dwords the
 * field lies in are picked with selects, then one funnel shift (no 64-bit shifts) */
static __device__ __forceinline__ u32 zs_bits(u32 d0, u32 d1, u32 d2, u32 d3, u32 topoff, u32 width)
{
	const u32 sh = 128u - topoff - width; /* bit offset of the field from the window's low end */
	const bool up = (sh & 64u) != 0;
	const u32 a0 = up ? d2 : d0, a1 = up ? d3 : d1, a2 = up ? 0u : d2;
	const bool odd = (sh & 32u) != 0;
	const u32 lo = odd ? a1 : a0, hi = odd ? a2 : a1;
	const u32 v = (u32)((((u64)hi << 32) | lo) >> (sh & 31u));
	return v & ((1u << width) - 1u);
}

/* Decoding cells of 16 bits: sym (6) | x << 6, x = the cell's "next state number" of RFC 8878 4.1.1 (next[sym]++ while the
 * table is numbered).  x lies in [2^(log - nb), 2^(log - nb + 1)), so the number of state bits is nb = log - floor(log2 x)
 * and the next state is ((x << nb) + bits) & (size - 1): both fall out of x with a count-leading-zeros and a shift, and
 * the extra bits of the code are a function of sym (zs_extra_bits).  At 2 bytes a cell the three tables of a block take
 * 2.5 KiB (5 KiB with zstd_dec.hip's 32-bit cells), which is what lets five of these waves share a CU.
 * Spread + numbering as fse_build (serial, one lane per table); returns 0 or -1. */
static __device__ int zs_build16(u16 *cell, const short *norm, int nsym, int log, u16 *next)
{
	const u32 size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
	u32 high = size - 1, pos = 0;
	for (int s = 0; s < nsym; s++) {
		if (norm[s] == -1) {
			cell[high--] = (u16)s;
			next[s] = 1;
		} else {
			next[s] = (u16)norm[s];
		}
	}
	for (int s = 0; s < nsym; s++) {
		const int c = norm[s];
		for (int i = 0; i < c; i++) {
			cell[pos] = (u16)s;
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
		cell[u] = (u16)(s | x << 6);
	}
	return 0;
}
/* Number_of_Bits of code `sym` for the lane's kind, without a table (the extra bits are on the state's chain; a lookup
 * would be a second LDS round trip per sequence): 0 below `lo`, three bits of `lut` per code for the ten codes from `lo`
 * on, sym - hoff (- 1 for the one code `fix`) above them.  LL: lo 16, hoff 19; ML: lo 32, hoff 36, fix 42 (5 bits, not 6);
 * OF: lo = hoff = 0 and no table codes (the code is the count). */
#define ZS_LUT_LL 0x346D2249u /* 1 1 1 1 2 2 3 3 4 6 (codes 16..25), three bits each, low code first */
#define ZS_LUT_ML 0x246D2249u /* 1 1 1 1 2 2 3 3 4 4 (codes 32..41) */
static __device__ __forceinline__ u32 zs_extra_bits(u32 sym, u32 lo, u32 hi, u32 hoff, u32 fix, u32 lut)
{
	const u32 mid = (lut >> ((3u * (sym - lo)) & 31u)) & 7u;
	const u32 up = sym - hoff - (sym == fix ? 1u : 0u);
	const u32 low = sym >= lo ? mid : 0u;
	return sym >= hi ? up : low;
}

#define ZS_PRE 0x40000000u
#define ZS_RLE 0x80000000u
#define ZS_REP 0xFFFFFFFFu

extern "C" __global__ void __launch_bounds__(64)
zmt_zstd_seq_kernel(const u8 *__restrict__ stream, u64 stream_bytes, const u64 *__restrict__ rec_off,
		    const u32 *__restrict__ rec_len, u32 nrec, const u64 *__restrict__ out_off,
		    const u32 *__restrict__ out_len, const u32 *__restrict__ status, u8 *__restrict__ seqbuf, u64 seqbuf_bytes)
{
	__shared__ __attribute__((aligned(16))) ZSeqLds L;
	const int lane = wv_lane();
	const u32 rec = blockIdx.x;
	if (rec >= nrec)
		return;
	const u32 cap = wv_readfirst(out_len[rec]);
	if (!zs_eligible(seqbuf_bytes, out_off[rec], cap))
		return;
	u64 *const reg = zs_region(seqbuf, out_off[rec]);
	u32 *const hdr = (u32 *)reg;
	const u32 nhdr = zs_nhdr(cap);
	const u32 seqcap = zs_seqcap(out_off[rec], cap);
	u64 *const seqs = reg + nhdr / 2;
	for (u32 i = (u32)lane; i < nhdr; i += 64)
		hdr[i] = 0;
	if (wv_readfirst(status[rec]) != ST_OK)
		return;
	if (lane < 36)
		L.llx[lane] = Z_LL_BASE[lane] | (u32)Z_LL_BITS[lane] << 24;
	if (lane < 53)
		L.mlx[lane] = Z_ML_BASE[lane] | (u32)Z_ML_BITS[lane] << 24;
	const u64 roff = rec_off[rec];
	const u32 rlen = wv_readfirst(rec_len[rec]);
	const u8 *r = stream + roff;
	const u8 *mem_lo = stream, *mem_hi = stream + stream_bytes + 256;

	/* ---- record + frame header: what zstd_dec.hip accepts; anything else is its business ---- */
	if (rlen < 12 + 6 || uld32(r) != ZMT_SKIP_MAGIC || uld32(r + 4) != 4 || uld32(r + 8) != rlen - 12)
		return;
	const u8 *f = r + 12;
	const u32 flen = rlen - 12;
	if (uld32(f) != ZMT_ZSTD_MAGIC)
		return;
	const u32 fhd = uld8(f + 4);
	const u32 fcs = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
	u32 hp = 5;
	u64 window = 0;
	{
		const u32 did_len = did == 3 ? 4 : did, fcs_len = fcs == 0 ? single : (1u << fcs);
		if ((fhd & 8) || did || flen < 5 + (1 - single) + did_len + fcs_len)
			return;
		if (!single) {
			const u32 wd = uld8(f + hp++);
			const u64 base = 1ull << (10 + (wd >> 3));
			window = base + (base >> 3) * (wd & 7);
		} else {
			window = cap; /* (the decoder checks content == cap before anything else) */
		}
		hp += fcs_len;
	}
	const u32 block_max = window < Z_BLOCK_MAX ? (u32)window : Z_BLOCK_MAX;

	/* ---- the predefined tables, once (lanes 0..2) ---- */
	if (lane < 3) {
		const short *def = lane == 0 ? Z_LL_DEF : lane == 1 ? Z_OF_DEF : Z_ML_DEF;
		const int n = lane == 0 ? 36 : lane == 1 ? 29 : 53;
		for (int i = 0; i < n; i++)
			L.norm[0][lane][i] = def[i];
		zs_build16(lane == 0 ? L.pre_ll : lane == 1 ? L.pre_of : L.pre_ml, L.norm[0][lane], n, lane == 1 ? 5 : 6, L.next[0][lane]);
	}
	wv_sync();
	L.valx[0][lane] = lane < 36 ? L.llx[lane] & 0xFFFFFFu : 0u;
	L.valx[1][lane] = lane < 32 ? 1u << lane : 0u;
	L.valx[2][lane] = lane < 53 ? L.mlx[lane] & 0xFFFFFFu : 0u;
	L.valx[3][lane] = 0;
	wv_sync();

	const u32 g = (u32)lane >> 2, sl = (u32)lane & 3;
	u32 ip = hp, bi = 0, used = 0; /* frame position, block index, sequences stored so far */
	bool endf = false, first_group = true, drop_all = false;
	/* lanes 0..2: the table of kind `lane` that a repeat mode refers to (as tp[] words), and the slot it lives in */
	u32 carried = 0, cslot = 0;
	bool have_carried = false;

	while (!endf) {
		/* ---- collect up to ZS_NB compressed blocks that have sequences (wave-uniform walk of the headers) ---- */
		u32 n = 0;
		while (!endf) {
			if (flen - ip < 3 || bi >= nhdr) {
				/* (a frame with more blocks than header words: what follows would be the decoder's, and what its first
				 * block needs from the tables before it is not known here) */
				if (bi >= nhdr)
					drop_all = true;
				endf = true;
				break;
			}
			const u64 h8 = (u64)uld32(f + ip) | (u64)uld32(f + ip + 4) << 32;
			const u32 bh = (u32)h8 & 0xFFFFFFu;
			const u32 last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
			const u32 q = ip + 3;
			if (btype == 3 || bsize > block_max || (btype != 1 && flen - q < bsize) || (btype == 1 && flen - q < 1)) {
				endf = true;
				break;
			}
			if (btype == 2) {
				if (bsize < 2) {
					endf = true;
					break;
				}
				const u64 v = h8 >> 24; /* the five bytes behind the block header */
				const u32 ltype = (u32)v & 3, sf = (u32)(v >> 2) & 3;
				u32 hl, csz, regen;
				if (ltype < 2) {
					if (sf == 0 || sf == 2) {
						regen = (u32)(v & 255) >> 3;
						hl = 1;
					} else if (sf == 1) {
						regen = (u32)(v & 0xFFFF) >> 4;
						hl = 2;
					} else {
						regen = (u32)(v & 0xFFFFFF) >> 4;
						hl = 3;
					}
					csz = ltype == 0 ? regen : 1;
				} else if (sf < 2) {
					regen = (u32)(v >> 4) & 1023;
					csz = (u32)(v >> 14) & 1023;
					hl = 3;
				} else if (sf == 2) {
					regen = (u32)(v >> 4) & 16383;
					csz = (u32)(v >> 18) & 16383;
					hl = 4;
				} else {
					regen = (u32)(v >> 4) & 262143;
					csz = (u32)(v >> 22) & 262143;
					hl = 5;
				}
				const u32 sq0 = hl + csz;
				if (regen > block_max || sq0 >= bsize) {
					endf = true;
					break;
				}
				const u32 s4 = uld32(f + q + sq0); /* (the buffer carries 256 readable bytes behind the stream) */
				u32 nseq = s4 & 255, p = 1;
				if (nseq == 255) {
					nseq = ((s4 >> 8) & 0xFFFF) + 0x7F00;
					p = 3;
				} else if (nseq >= 128) {
					nseq = ((nseq - 128) << 8) + ((s4 >> 8) & 255);
					p = 2;
				}
				if (nseq) {
					if (sq0 + p >= bsize) {
						endf = true;
						break;
					}
					const u32 modes = (s4 >> (8 * p)) & 255;
					if (modes & 3) {
						endf = true;
						break;
					}
					if (n == ZS_NB)
						break; /* the next group starts here */
					if (lane == 0) {
						L.blk[n][ZB_DESC] = q + sq0 + p + 1;
						L.blk[n][ZB_END] = q + bsize;
						L.blk[n][ZB_NSEQ] = nseq;
						L.blk[n][ZB_MODES] = modes;
						L.blk[n][ZB_BI] = bi;
					}
					n++;
				}
			}
			bi++;
			ip = q + (btype == 1 ? 1 : bsize);
			if (last)
				endf = true;
		}
		wv_sync();
		if (first_group) {
			first_group = false;
			/* frames with a single block of sequences have nothing to decode side by side, and the units of the device
			 * encoder (predefined tables everywhere, or described once and repeated) are the decoder's unit path */
			if (n < 2)
				return;
			const u32 m0 = L.blk[0][ZB_MODES], m1 = L.blk[1][ZB_MODES];
			if (m1 == 0xFC || (m0 == 0 && m1 == 0))
				return;
		}
		if (n == 0 || drop_all)
			break;

		/* ---- table descriptions into LDS: 16 lanes x 16 bytes per block, four blocks a pass ---- */
		for (u32 pass = 0; pass * 4 < n; pass++) {
			const u32 b = pass * 4 + ((u32)lane >> 4), o = 16u * ((u32)lane & 15);
			if (b < n) {
				const u8 *p = f + L.blk[b][ZB_DESC] + o;
				u64 a = 0, c = 0;
				if (p + 16 <= mem_hi) {
					a = ld64u(p);
					c = ld64u(p + 8);
				} else {
					for (int k = 0; k < 8; k++) {
						if (p + k < mem_hi)
							a |= (u64)p[k] << (8 * k);
						if (p + 8 + k < mem_hi)
							c |= (u64)p[8 + k] << (8 * k);
					}
				}
				*(u64 *)(&L.stage[b][o]) = a;
				*(u64 *)(&L.stage[b][o + 8]) = c;
			}
		}
		(void)mem_lo;
		wv_sync();
		/* ---- tables, ZS_PB blocks at a time: one lane per block reads its three descriptions, then lane (block, kind)
		 * builds the table the block describes into slot (cslot of the kind + j) mod ZS_NB, j = the block's place in the
		 * group: block 0 gets the slot of the table a repeat mode would refer to (it either repeats it or replaces it),
		 * so nothing that is still needed is overwritten ---- */
		const u32 cs_kind = wv_readlane(cslot, (int)(sl < 3 ? sl : 0u)); /* (lanes 0..2 hold the kinds' carried slots) */
		for (u32 j0 = 0; j0 < n; j0 += ZS_PB) {
			const u32 j = j0 + g;
			if (sl == 0 && g < ZS_PB && j < n) {
				const u8 *d = L.stage[j];
				const u32 room = L.blk[j][ZB_END] - L.blk[j][ZB_DESC];
				const u32 avail = room < 256 ? room : 256;
				const u32 modes = L.blk[j][ZB_MODES];
				u32 p = 0, err = 0;
				for (int t = 0; t < 3 && !err; t++) {
					const int mode = (int)(modes >> (6 - 2 * t)) & 3;
					const int max_sym = t == 0 ? 36 : t == 1 ? 32 : 53, max_log = t == 1 ? 8 : 9;
					u32 spec = ZS_REP;
					if (mode == 0) {
						spec = ZS_PRE;
					} else if (mode == 1) {
						if (p >= avail || d[p] >= max_sym)
							err = 1;
						else
							spec = ZS_RLE | d[p++];
					} else if (mode == 2) {
						int nsym = 0, lg = 0;
						const int u = p < avail ? fse_read_ncount(d + p, avail - p, L.norm[g][t], max_sym, max_log, &nsym, &lg) : -1;
						if (u < 0) {
							err = 1;
						} else {
							p += (u32)u;
							spec = (u32)nsym | (u32)lg << 8;
						}
					}
					L.spec[j][t] = err ? ZS_REP : spec;
				}
				L.blk[j][ZB_HDR] = p;
				L.blk[j][ZB_ERR] = err;
			}
			wv_sync();
			if (sl < 3 && g < ZS_PB && j < n && !L.blk[j][ZB_ERR]) {
				const u32 spec = L.spec[j][sl];
				if (spec != ZS_REP && spec != ZS_PRE) {
					const u32 slot = (cs_kind + j) % ZS_NB;
					u16 *cells = sl == 0 ? L.ll[slot] : sl == 1 ? L.of[slot] : L.ml[slot];
					if (spec & ZS_RLE)
						cells[0] = (u16)((spec & 255) | 1u << 6); /* RLE: one cell, log 0, x = 1: no state bits */
					else if (zs_build16(cells, L.norm[g][sl], (int)(spec & 255), (int)((spec >> 8) & 255), L.next[g][sl]) != 0)
						L.blk[j][ZB_ERR] = 2; /* (the three lanes of a block may all write this: the same value) */
				}
			}
			wv_sync();
		}
		/* ---- which table every block decodes with: lane t < 3 walks the blocks of the group for kind t ---- */
		u32 nok = n; /* blocks of the group whose tables are sound */
		if (lane < 3) {
			u32 cur = carried, cs = cslot;
			bool have = have_carried;
			for (u32 j = 0; j < n; j++) {
				const u32 spec = L.spec[j][lane];
				u32 w = 0xFFFFFFFFu;
				if (L.blk[j][ZB_ERR] || (spec == ZS_REP && !have)) {
					nok = nok < j ? nok : j;
				} else if (spec == ZS_REP) {
					w = cur;
				} else if (spec == ZS_PRE) {
					w = (u32)((lane == 0 ? L.pre_ll : lane == 1 ? L.pre_of : L.pre_ml) - &L.ll[0][0]) | (u32)(lane == 1 ? 5 : 6) << 24;
				} else {
					const u32 slot = (cslot + j) % ZS_NB;
					const u16 *cells = lane == 0 ? L.ll[slot] : lane == 1 ? L.of[slot] : L.ml[slot];
					w = (u32)(cells - &L.ll[0][0]) | ((spec & ZS_RLE) ? 0u : (spec >> 8) & 255) << 24;
					cs = slot;
				}
				if (w != 0xFFFFFFFFu) {
					cur = w;
					have = true;
				}
				L.tp[j][lane] = w;
			}
			carried = cur;
			cslot = cs;
			have_carried = have;
		}
		nok = wv_readlane(nok, 0) < wv_readlane(nok, 1) ? wv_readlane(nok, 0) : wv_readlane(nok, 1);
		nok = nok < wv_readlane(nok, 2) ? nok : wv_readlane(nok, 2);
		wv_sync();
		if (nok < n) {
			/* an unsound table: the decoder will reject the frame at that block; nothing from here on is marked, and
			 * what was marked stays valid (the serial path judges the block with its own tables... which it does not
			 * have for repeat modes behind marked blocks: drop everything) */
			drop_all = true;
			break;
		}

		/* ---- the bitstreams: group g = block g, lanes LL / OF / ML / idle ---- */
		bool act = g < n, gbad = false;
		const u32 w_my = act && sl < 3 ? L.tp[g][sl] : 0;
		const u16 *mytab = &L.ll[0][0] + (w_my & 0xFFFFFFu);
		const u32 x_lo = sl == 0 ? 16u : sl == 2 ? 32u : 0u, x_hi = sl == 0 ? 26u : sl == 2 ? 42u : sl == 1 ? 0u : 64u;
		const u32 x_off = sl == 0 ? 19u : sl == 2 ? 36u : 0u, x_fix = sl == 2 ? 42u : 255u;
		const u32 x_lut = sl == 0 ? ZS_LUT_LL : ZS_LUT_ML;
		const u32 mylog = sl < 3 ? w_my >> 24 : 0u;
		const u32 ll_log = wv_quad(mylog, 0), of_log = wv_quad(mylog, 1), ml_log = wv_quad(mylog, 2);
		const u32 tmask = (1u << mylog) - 1;
		u32 g_off = 0, g_len = 0, g_n = 0;
		if (act) {
			g_off = L.blk[g][ZB_DESC] + L.blk[g][ZB_HDR];
			g_n = L.blk[g][ZB_NSEQ];
			if (g_off >= L.blk[g][ZB_END])
				gbad = true;
			else
				g_len = L.blk[g][ZB_END] - g_off;
		}
		int bp = 0;
		if (act && !gbad) {
			const u32 lb = f[g_off + g_len - 1];
			if (lb == 0)
				gbad = true;
			else
				bp = 8 * (int)(g_len - 1) + hb32(lb);
			if (bp < (int)(ll_log + of_log + ml_log))
				gbad = true;
		}
		u32 ngrp = n;
		{
			const u64 bm = wv_ballot(act && gbad);
			if (bm)
				ngrp = (u32)(wv_ffs(bm) - 1) >> 2;
		}
		bool stopped = ngrp < n; /* a stream the decoder has to look at itself */
		act = g < ngrp;
		const u32 incl = wv_scan_incl(sl == 0 && act ? g_n : 0u);
		{
			const u64 om = wv_ballot(act && (u64)used + incl > (u64)seqcap);
			if (om) {
				ngrp = (u32)(wv_ffs(om) - 1) >> 2; /* the region is full: the blocks from here on are the decoder's */
				stopped = true;
			}
		}
		act = g < ngrp;
		const u32 my_start = used + incl - g_n;
		u64 *myseq = seqs + my_start;
		if (ngrp) {
			const u32 e_of = sl == 1 ? 0u : ~0u, e_ml = sl == 0 ? ~0u : 0u;
			const u32 s_ll = sl == 0 ? 0u : ~0u, s_ml = sl == 1 ? ~0u : 0u;
			u8 *gwin = &L.stage[0][0] + ZS_WSTRIDE * (g < ZS_NB ? g : 0u); /* ZS_WIN-byte window per block, a quarter per lane */
			const u32 *myval = L.valx[sl];
			u32 state = 0, done = 0;
			bool first = true;
			while (wv_any(act && done < g_n)) {
				const int whi = (bp + 7) >> 3, wlo = whi - (int)ZS_WIN;
				wv_sync();
				if (act && done < g_n) {
					const int rel = (int)g_off + wlo + (int)(ZS_WIN / 4u) * (int)sl;
					constexpr int NW = (int)(ZS_WIN / 32u); /* 8-byte words of a lane's quarter */
					u64 w[NW];
					ZMT_UNROLL
					for (int j = 0; j < NW; j++)
						w[j] = 0;
					if (rel >= 0) {
						ZMT_UNROLL
						for (int j = 0; j < NW; j++)
							w[j] = ld64u(f + rel + 8 * j);
					} else {
						for (int j = 0; j < NW; j++)
							for (int k2 = 0; k2 < 8; k2++)
								if (rel + 8 * j + k2 >= 0)
									w[j] |= (u64)f[rel + 8 * j + k2] << (8 * k2);
					}
					ZMT_UNROLL
					for (int j = 0; j < NW; j++)
						*(u64 *)(gwin + (ZS_WIN / 4u) * sl + 8u * (u32)j) = w[j];
				}
				wv_sync();
				const u8 *winb = gwin - wlo - 15; /* winb[b + 15] = byte b of my stream */
				if (first) {
					first = false;
					const int tb = (bp - 1) >> 3;
					u64 w1 = 0, w0 = 0;
					if (act) {
						w1 = ld64u(winb + tb);
						w0 = ld64u(winb + tb + 8);
					}
					const u32 skip = (u32)(8 * (tb + 1) - bp);
					state = xbits(w0, w1, skip + (sl == 0 ? 0u : sl == 1 ? ll_log : ll_log + of_log), mylog);
					bp -= (int)(ll_log + of_log + ml_log);
				}
				for (;;) {
					/* eight sequences of every block; lane sl keeps 2 sl, 2 sl + 1.  While every block that is still being
					 * walked has more than eight sequences left, the steps carry no per-sequence tests (FAST): an idle
					 * lane's state and bit position may then run on, nothing reads them again */
					u64 r0 = 0, r1 = 0;
#define ZS_STEPS8(FAST)                                                                                            \
	ZMT_UNROLL                                                                                                 \
	for (int i = 0; i < 8; i++) {                                                                              \
		const bool on = (FAST) ? act : (act && done + (u32)i < g_n);                                       \
		int tb = (bp - 1) >> 3;                                                                            \
		tb = tb < wlo + 15 ? wlo + 15 : tb; /* a stream gone bad stays inside its window */                \
		u32 d0, d1, d2, d3;                                                                                \
		zs_ld128(winb + tb, d0, d1, d2, d3);                                                               \
		const u32 cell = mytab[state & tmask];                                                             \
		const u32 sym = cell & 63u, x = cell >> 6;                                                         \
		const u32 vbase = myval[sym]; /* (asked for as soon as the cell is there: off the state's chain) */ \
		const u32 skip = (u32)(8 * (tb + 1) - bp) & 127u;                                                  \
		const u32 nbx = (mylog + (u32)__builtin_clz(x | 1u) - 31u) & 15u; /* log - floor(log2 x) */       \
		const u32 nb = (FAST) ? nbx : (done + (u32)i + 1 == g_n ? 0u : nbx);                               \
		const u32 ab = zs_extra_bits(sym, x_lo, x_hi, x_off, x_fix, x_lut);                                \
		const u32 pk = ab | nb << 8;                                                                       \
		const u32 p_ll = wv_quad(pk, 0), p_of = wv_quad(pk, 1), p_ml = wv_quad(pk, 2);                     \
		const u32 a_ll = p_ll & 255, n_ll = p_ll >> 8, a_of = p_of & 255, n_of = p_of >> 8;                \
		const u32 a_ml = p_ml & 255, n_ml = p_ml >> 8;                                                     \
		const u32 base3 = skip + a_of + a_ml + a_ll;                                                       \
		const u32 eo = skip + (a_of & e_of) + (a_ml & e_ml);                                               \
		const u32 so = base3 + (n_ll & s_ll) + (n_ml & s_ml);                                              \
		const u32 extra = zs_bits(d0, d1, d2, d3, eo, ab);                                                 \
		const u32 sbits = zs_bits(d0, d1, d2, d3, so, nb);                                                 \
		const u32 val = vbase + extra;                                                                     \
		if (on && sl == 1 && sym > 27)                                                                     \
			gbad = true; /* does not fit the packing: left to the decoder */                           \
		const u32 v_ll = wv_quad(val, 0), v_of = wv_quad(val, 1), v_ml = wv_quad(val, 2);                  \
		const u64 rcd = (u64)v_ll | (u64)v_ml << 18 | (u64)v_of << 36;                                     \
		if ((u32)(i >> 1) == sl) {                                                                         \
			if (i & 1)                                                                                 \
				r1 = rcd;                                                                          \
			else                                                                                       \
				r0 = rcd;                                                                          \
		}                                                                                                  \
		if (FAST) {                                                                                        \
			state = (x << nbx) + sbits; /* (masked where it is used) */                                \
			bp -= (int)(base3 - skip + n_ll + n_ml + n_of);                                            \
		} else {                                                                                           \
			state = on ? (x << nbx) + sbits : state;                                                   \
			bp -= on ? (int)(base3 - skip + n_ll + n_ml + n_of) : 0;                                   \
		}                                                                                                  \
	}
					if (!wv_any(act && done + 8u >= g_n)) {
						ZS_STEPS8(true)
					} else {
						ZS_STEPS8(false)
					}
					if (act && done + 2 * sl < g_n)
						st64g((u8 *)(myseq + done + 2 * sl), r0);
					if (act && done + 2 * sl + 1 < g_n)
						st64g((u8 *)(myseq + done + 2 * sl + 1), r1);
					if (act && done < g_n) {
						done = g_n - done < 8 ? g_n : done + 8;
						if (bp < 0 || (done == g_n && bp != 0))
							gbad = true;
					}
					/* (the four lanes of a block stop together) */
					gbad = (wv_quad((u32)gbad, 0) | wv_quad((u32)gbad, 1) | wv_quad((u32)gbad, 2)) != 0;
					if (gbad)
						act = false;
					const bool more = act && done < g_n;
					/* a sequence consumes at most 27 + 16 + 16 offset / match / literal extra bits + 9 + 9 + 8 state
					 * bits = 85; with ZS_REFILL bits already behind the cursor, the 8th sequence's 16-byte read -- issued
					 * with ZS_REFILL + 7 x 85 bits consumed -- still lies inside the window */
					static_assert(ZS_REFILL + 7 * 85 + 8 * 16 <= 8 * ZS_WIN, "the 16-byte read of the 8th sequence leaves the window");
					if (!wv_any(more) || wv_any(more && 8 * whi - bp > (int)ZS_REFILL))
						break;
				}
			}
			{
				const u64 bm = wv_ballot(gbad && g < ngrp);
				if (bm) {
					const u32 gb = (u32)(wv_ffs(bm) - 1) >> 2;
					ngrp = gb < ngrp ? gb : ngrp;
					stopped = true;
				}
			}
		}
		/* ---- mark the blocks that came out clean ---- */
		if (sl == 0 && g < ngrp)
			hdr[L.blk[g][ZB_BI]] = my_start + 1;
		{
			const u32 tot = wv_readlane(incl, (int)(4 * (ngrp ? ngrp - 1 : 0)));
			used += ngrp ? tot : 0;
		}
		wv_sync();
		if (stopped) {
			/* the decoder continues by itself from block `ngrp` of this group.  Its tables are then the ones it built
			 * itself -- none for the blocks marked here -- so a repeat mode in that block (or a later one, before all
			 * three kinds were described again) would find nothing: drop the frame's marks unless that block describes
			 * all three tables itself */
			const u32 m = L.blk[ngrp][ZB_MODES];
			const bool self = ((m >> 6) & 3) != 3 && ((m >> 4) & 3) != 3 && ((m >> 2) & 3) != 3;
			if (!self)
				drop_all = true;
			break;
		}
	}
	/* the decoder may also have to go on by itself where this kernel ended for a reason of the stream's (an odd header:
	 * endf without `last`): that block is rejected by the decoder whatever its tables are -- nothing to drop */
#ifdef ZMT_EMU
	if (getenv("ZMT_EMU_DEBUG") && lane == 0)
		fprintf(stderr, "zstd_seq: record %u: %u sequences of %u blocks decoded ahead%s\n", rec, used, bi, drop_all ? " (dropped)" : "");
#endif
	if (drop_all) {
		wave_mem_fence();
		for (u32 i = (u32)lane; i < nhdr; i += 64)
			hdr[i] = 0;
	}
}
