/*
 * zstd_dec.hip -- zstd frame decoder for gfx950, one wave per record.
 *
 * Replaces ZSTD_decompressStream as called per record by the reference
 * (/root/reference/lib/zstd-mt_decompress.c:464, one frame per 12-byte skippable record, :300-369).
 * Format: RFC 8878 -- raw / RLE / compressed blocks, Huffman literals (direct or FSE-compressed
 * weights, 1 or 4 streams, treeless), FSE sequence tables (predefined / RLE / compressed /
 * repeat), repeat offsets.  No dictionaries.
 *
 * Work split inside the wave (the format is a chain of serial bitstreams, so lanes are used where
 * the format offers independent streams and for all byte moving):
 *   - every header / table description is staged into LDS with one coalesced load and parsed there;
 *   - Huffman literals: the 4 streams decode on 4 lanes from 256-byte LDS windows that the wave
 *     refills cooperatively every 128 symbols; decoded literals go to a per-record scratch in HBM;
 *   - FSE tables: the three tables of a block are built on three lanes at once;
 *   - sequences: lane 0 walks the FSE bitstream from a 1 KiB LDS window, 64 sequences at a time;
 *     the wave then executes the 64 sequences in parallel: prefix sums give output and literal
 *     positions, every lane copies its own literals and match (8-byte unaligned global accesses),
 *     matches that read this batch's own output resolve in watermark rounds, long copies are
 *     done by the whole wave;
 *   - units: a block with a Huffman tree (or raw literals) followed by blocks with treeless (raw)
 *     literals, all with the predefined sequence tables or all with the tables the first one
 *     describes (repeat mode) -- what the device encoder writes per 128 KiB -- is
 *     decoded side by side: up to 64 Huffman streams on 64 lanes, then up to 16 sequence
 *     bitstreams on 16 x (LL, OF, ML) lanes into the record's scratch; the blocks then only
 *     execute.  Anything else ends the unit and takes the per-block path above.
 */
#include "zstd_dec_common.h"
#include "zstd_dec_seq.h"

/* LDS of one wave.  Table capacities are template parameters: the format allows Huffman codes of up
 * to 11 bits and FSE tables of 2^9 / 2^8 / 2^9 cells (13 KiB per wave, 12 waves per CU), but frames
 * whose blocks stay within 10 bits and the predefined 2^6 / 2^5 / 2^6 tables -- everything the
 * device encoder writes -- fit in 6 KiB, i.e. the 16 waves per CU the register budget allows.  The
 * small variant runs first and hands records that exceed it to the general one. */
template <int HCAP, int LLCAP, int OFCAP, int MLCAP> struct ZLdsT {
	u16 huf[1 << HCAP];  /* sym | nbits << 8 */
	u32 ll[1 << LLCAP], of[1 << OFCAP], ml[1 << MLCAP]; /* FSE cells, see ZC_* */
	u8 below[16];   /* stage[-16..0): 8-byte reads may start below the window */
	u8 stage[Z_STAGE + 16];
	u64 sq[3][64]; /* one batch of sequences, per LL / OF / ML: extra bits | code << 32 */
	u8 w[256];
	short norm[3][64];
	u16 next[3][64];
	u32 llx[36], mlx[53]; /* value base | extra bits << 24 of every LL / ML code */
	u32 misc[16];
	static constexpr int hcap = HCAP, llcap = LLCAP, ofcap = OFCAP, mlcap = MLCAP;
};
typedef ZLdsT<11, 9, 8, 9> ZLds;      /* everything RFC 8878 allows */
typedef ZLdsT<11, 6, 5, 6> ZLdsSmall; /* predefined-size sequence tables (frames whose sequences the pre-pass decoded need none) */
static_assert(offsetof(ZLdsSmall, sq) - offsetof(ZLdsSmall, below) == 1056 &&
		      offsetof(ZLds, sq) - offsetof(ZLds, below) == 1056,
	      "below | stage | sq are one region (per-lane stream windows)");

enum { ZM_ERR = 0, ZM_A, ZM_B, ZM_C, ZM_D, ZM_E, ZM_F, ZM_G, ZM_H };


/* ------------------------------------------------------------------ the kernel */
#ifdef ZMT_EMU
static inline u32 zbad_(int line)
{
	if (getenv("ZMT_EMU_DEBUG") && wv_lane() == 0)
		fprintf(stderr, "zstd_dec: bad block flagged at line %d\n", line);
	return ST_BAD_BLOCK;
}
#define ZBAD() zbad_(__LINE__)
#else
#define ZBAD() ST_BAD_BLOCK
#endif
#ifndef ZMT_EMU
#define ZT() (PROF ? (u64)clock64() : 0ull)
#else
#define ZT() 0ull
#endif
#define ZP(i)                                                                                      \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = ZT();                                                       \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

template <bool PROF, typename LDS>
static __device__ __forceinline__ void
zstd_dec_body(LDS &L, u32 want_status, const u8 *__restrict__ stream, u64 stream_bytes, const u64 *__restrict__ rec_off,
	      const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base, const u64 *__restrict__ out_off,
	      u32 *__restrict__ out_len, u8 *__restrict__ litbuf, u32 *__restrict__ status,
	      u32 *__restrict__ chk_expect, u32 *__restrict__ chk_valid, unsigned long long *prof, u8 *seqbuf, u64 seqcap)
{
	const int lane = wv_lane();
	u64 pc[PROF ? 8 : 1] = {0}, tq = ZT();
	const u64 t_begin = tq;
	(void)t_begin;
	if (lane < 36)
		L.llx[lane] = Z_LL_BASE[lane] | (u32)Z_LL_BITS[lane] << 24;
	if (lane < 53)
		L.mlx[lane] = Z_ML_BASE[lane] | (u32)Z_ML_BITS[lane] << 24;
	const u32 rec = blockIdx.x;
	if (rec >= nrec)
		return;
	if (wv_readfirst(status[rec]) != want_status)
		return; /* rejected by the probe kernel, or not this variant's record */
	if (lane == 0)
		chk_valid[rec] = 0;
	const u64 roff = rec_off[rec];
	const u32 rlen = wv_readfirst(rec_len[rec]);
	const u8 *r = stream + roff;
	u8 *out = out_base + out_off[rec];
	const u32 cap = wv_readfirst(out_len[rec]);
	u8 *lit_scratch = litbuf + (u64)rec * Z_LITSLOT;
	const u8 *mem_lo = stream, *mem_hi = stream + stream_bytes + 256; /* readable range */
	u32 stc = ST_OK;

	/* ---- record + frame header (RFC 8878 3.1.1) ---- */
	if (rlen < 12 + 6 || uld32(r) != ZMT_SKIP_MAGIC || uld32(r + 4) != 4 || uld32(r + 8) != rlen - 12) {
		if (lane == 0)
			status[rec] = ST_BAD_RECORD;
		return;
	}
	const u8 *f = r + 12;
	const u32 flen = rlen - 12;
	if (uld32(f) != ZMT_ZSTD_MAGIC) {
		if (lane == 0)
			status[rec] = ST_BAD_FRAME;
		return;
	}
	const u32 fhd = uld8(f + 4);
	const u32 fcs = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3, has_chk = (fhd >> 2) & 1;
	u32 hp = 5;
	u64 window = 0, content = ~0ull;
	{
		const u32 did_len = did == 3 ? 4 : did, fcs_len = fcs == 0 ? single : (1u << fcs);
		if ((fhd & 8) || flen < 5 + (1 - single) + did_len + fcs_len) {
			if (lane == 0)
				status[rec] = ST_BAD_FRAME;
			return;
		}
		if (did) {
			u32 id = 0;
			for (u32 i = 0; i < did_len; i++)
				id |= uld8(f + hp + (1 - single) + i);
			if (id) {
				if (lane == 0)
					status[rec] = ST_UNSUPPORTED;
				return;
			}
		}
		if (!single) {
			const u32 wd = uld8(f + hp++);
			const u64 base = 1ull << (10 + (wd >> 3));
			window = base + (base >> 3) * (wd & 7);
		}
		hp += did_len;
		if (fcs == 0 && single)
			content = uld8(f + hp);
		else if (fcs == 1)
			content = uld16(f + hp) + 256;
		else if (fcs == 2)
			content = uld32(f + hp);
		else if (fcs == 3)
			content = (u64)uld32(f + hp) | (u64)uld32(f + hp + 4) << 32;
		hp += fcs_len;
		if (single)
			window = content;
	}
	if (content != ~0ull && content != cap) {
		if (lane == 0)
			status[rec] = ST_SIZE_MISMATCH;
		return;
	}
	const u32 block_max = window < Z_BLOCK_MAX ? (u32)window : Z_BLOCK_MAX;

	u32 rep0 = 1, rep1 = 4, rep2 = 8;
	bool huf_ok = false;
	int huf_log = 0;
	u32 pre_left = 0, pre_lit = 0; /* blocks ahead whose literals are already decoded / where they start */
	u32 sq_left = 0, sq_pos = 0; /* blocks (this one first) whose sequences are decoded / where they start */
	bool unit_head = false;
	u32 unit_n = 0; /* blocks behind a unit head whose sequence sections are listed in L.w */
	u64 *seq_scratch = (u64 *)(lit_scratch + Z_BLOCK_MAX + 256u);
	bool my_tab_ok = false; /* lanes 0..2: state of the LL / OF / ML table this lane builds */
	bool my_tab_pre = false; /* ... and whether it currently holds the predefined distribution */
	int my_tab_log = 0;
	u32 opos = 0, ip = hp;
	/* sequences decoded ahead by zmt_zstd_seq_kernel (zstd_dec_seq.hip): hdr[i] != 0 says block i's are at seq[hdr[i] - 1] */
	const bool pre_on = seqbuf != nullptr && zs_eligible(seqcap, out_off[rec], cap);
	const u32 *pre_hdr = pre_on ? (const u32 *)zs_region(seqbuf, out_off[rec]) : nullptr;
	const u64 *pre_seq = pre_on ? zs_region(seqbuf, out_off[rec]) + zs_nhdr(cap) / 2 : nullptr;
	const u32 pre_nhdr = pre_on ? zs_nhdr(cap) : 0;
	u32 bi = 0; /* index of the block, counting every block of the frame */

	for (;;) {
		if (flen - ip < 3) {
			stc = ZBAD();
			break;
		}
		const u32 bh = uld8(f + ip) | uld8(f + ip + 1) << 8 | uld8(f + ip + 2) << 16;
		const u32 last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
		ip += 3;
		const u8 *src = f + ip;
		if (btype == 3 || bsize > block_max || (btype != 1 && flen - ip < bsize) ||
		    (btype == 1 && flen - ip < 1)) {
			stc = ZBAD();
			break;
		}
		if (btype == 0) {
			if (cap - opos < bsize) {
				stc = ST_SIZE_MISMATCH;
				break;
			}
			wave_copy(out + opos, src, bsize, lane);
			opos += bsize;
			ip += bsize;
		} else if (btype == 1) {
			if (cap - opos < bsize) {
				stc = ST_SIZE_MISMATCH;
				break;
			}
			const u8 v = (u8)uld8(src);
			for (u32 i = (u32)lane; i < bsize; i += 64)
				out[opos + i] = v;
			opos += bsize;
			ip += 1;
		} else {
			/* ================= compressed block ================= */
			ZP(6);
			const u32 bstart = opos;
			if (bsize < 2) {
				stc = ZBAD();
				break;
			}
			/* ---- literals section header + tree description, from LDS ---- */
			wv_sync();
			stage_load(L.stage, src, 0, 256, 0, mem_lo, mem_hi, lane);
			wv_sync();
			if (lane == 0) {
				const u8 *d = L.stage;
				const u32 ltype = d[0] & 3, sf = (d[0] >> 2) & 3;
				u32 regen = 0, csz = 0, hl = 0, streams = 1, err = 0, tree = 0;
				if (ltype < 2) {
					if (sf == 0 || sf == 2) {
						regen = d[0] >> 3;
						hl = 1;
					} else if (sf == 1) {
						regen = (d[0] | d[1] << 8) >> 4;
						hl = 2;
					} else {
						regen = (d[0] | d[1] << 8 | d[2] << 16) >> 4;
						hl = 3;
					}
					csz = ltype == 0 ? regen : 1;
				} else {
					const u64 v = (u64)ld32u(d) | (u64)d[4] << 32;
					if (sf < 2) {
						regen = (u32)(v >> 4) & 1023;
						csz = (u32)(v >> 14) & 1023;
						streams = sf == 0 ? 1 : 4;
						hl = 3;
					} else if (sf == 2) {
						regen = (u32)(v >> 4) & 16383;
						csz = (u32)(v >> 18) & 16383;
						streams = 4;
						hl = 4;
					} else {
						regen = (u32)(v >> 4) & 262143;
						csz = (u32)(v >> 22) & 262143;
						streams = 4;
						hl = 5;
					}
					if (regen == 0)
						err = 1;
				}
				if (regen > block_max || hl + csz > bsize)
					err = 1;
				if (!err && ltype == 2) {
					/* tree description: at most 1 + 128 bytes, inside the staged 256 */
					u32 avail = csz < 256 - hl ? csz : 256 - hl;
					int nw = 0, lg = 0;
					const int used = huf_read_weights(d + hl, avail, L.w, &nw, &lg, (u32 *)L.sq[0],
									  L.norm[0], L.next[0]);
					if (used < 0) {
						err = 1;
					} else {
						tree = (u32)used;
						L.misc[ZM_G] = (u32)nw;
						L.misc[ZM_H] = (u32)lg;
					}
				}
				L.misc[ZM_ERR] = err;
				L.misc[ZM_A] = ltype;
				L.misc[ZM_B] = regen;
				L.misc[ZM_C] = csz;
				L.misc[ZM_D] = hl;
				L.misc[ZM_E] = streams;
				L.misc[ZM_F] = tree;
			}
			wv_sync();
			if (L.misc[ZM_ERR]) {
				stc = ZBAD();
				break;
			}
			const u32 ltype = L.misc[ZM_A], regen = L.misc[ZM_B], lcsz = L.misc[ZM_C], lhl = L.misc[ZM_D];
			const u32 nstreams = L.misc[ZM_E], tree = L.misc[ZM_F];
			const u8 *lit = lit_scratch;
			if (ltype == 0) {
				lit = src + lhl;
				/* Raw literals (a unit whose literals did not compress): the blocks that follow the
				 * same way still share the predefined sequence tables; list them for the unit-wide
				 * sequence decode below (same records as the Huffman look-ahead writes) */
				if (sq_left == 0 && !last) {
					u32 *pre = (u32 *)L.w;
					wv_sync();
					if (lane == 0) {
						u32 n = 0, q = ip + bsize;
						while (n < 15) {
							if (flen - q < 8)
								break;
							const u8 *h = f + q;
							const u32 bh2 = (u32)h[0] | (u32)h[1] << 8 | (u32)h[2] << 16;
							const u32 bs2 = bh2 >> 3;
							if (((bh2 >> 1) & 3) != 2 || bs2 > block_max || bs2 < 5 || flen - (q + 3) < bs2)
								break;
							const u32 b0 = h[3], sf2 = (b0 >> 2) & 3;
							if ((b0 & 3) != 0)
								break;
							u32 rg, hl2;
							if (sf2 == 1) {
								rg = (b0 | (u32)h[4] << 8) >> 4;
								hl2 = 2;
							} else if (sf2 == 3) {
								rg = (b0 | (u32)h[4] << 8 | (u32)h[5] << 16) >> 4;
								hl2 = 3;
							} else {
								rg = b0 >> 3;
								hl2 = 1;
							}
							if (rg > block_max || hl2 + rg > bs2)
								break;
							pre[4 * n] = q + 3 + hl2;
							pre[4 * n + 1] = rg;
							pre[4 * n + 3] = q + 3 + bs2;
							n++;
							q += 3 + bs2;
							if (bh2 & 1)
								break;
						}
						L.misc[ZM_A] = n;
					}
					wv_sync();
					unit_n = L.misc[ZM_A];
					unit_head = unit_n != 0;
				}
			} else if (ltype == 1) {
				const u8 v = (u8)uld8(src + lhl);
				for (u32 i = (u32)lane * 8; i < regen + 8; i += 512)
					st64g(lit_scratch + i, 0x0101010101010101ull * v);
			} else {
				if (ltype == 2) {
					const int nw = (int)L.misc[ZM_G];
					huf_log = (int)L.misc[ZM_H];
					if (huf_log > LDS::hcap) {
						stc = ST_NEEDS_GENERAL;
						break;
					}
					for (u32 i = (u32)lane; i < (1u << LDS::hcap); i += 64)
						L.huf[i] = 0;
					wv_sync();
					huf_fill(L.huf, L.w, nw, huf_log, lane);
					huf_ok = true;
					wv_sync();
				} else if (!huf_ok) {
					stc = ZBAD();
					break;
				}
				bool lit_done = false;
				if (ltype == 3 && pre_left) {
					/* decoded together with the block that carried the tree (below) */
					lit = lit_scratch + pre_lit;
					pre_lit += regen;
					pre_left--;
					lit_done = true;
				} else {
					pre_left = 0;
				}
				/* ---- blocks that follow with `treeless` literals: all their streams at once ----
				 * The device encoder writes one tree per 128 KiB unit and up to 15 more blocks that
				 * reuse it; four lanes per block decode up to 64 streams side by side instead of 4.
				 * Anything odd in a block ahead just ends the look-ahead there: the block is then
				 * decoded (and judged) when its turn comes. */
				if (ltype == 2 && nstreams == 4 && !last && lcsz >= tree + 10u) {
					u32 *pre = (u32 *)L.w; /* the weights are in the table now */
					wv_sync();
					if (lane == 0) {
						u32 n = 0, cum = regen, q = ip + bsize;
						while (n < 15) {
							if (flen - q < 8)
								break;
							const u8 *h = f + q;
							const u32 bh2 = (u32)h[0] | (u32)h[1] << 8 | (u32)h[2] << 16;
							const u32 bs2 = bh2 >> 3;
							if (((bh2 >> 1) & 3) != 2 || bs2 > block_max || bs2 < 5 || flen - (q + 3) < bs2)
								break;
							const u64 v = (u64)h[3] | (u64)h[4] << 8 | (u64)h[5] << 16 | (u64)h[6] << 24 | (u64)h[7] << 32;
							const u32 sf2 = (u32)(v >> 2) & 3;
							if ((v & 3) != 3 || sf2 == 0)
								break;
							u32 rg, cs, hl2;
							if (sf2 == 1) {
								rg = (u32)(v >> 4) & 1023;
								cs = (u32)(v >> 14) & 1023;
								hl2 = 3;
							} else if (sf2 == 2) {
								rg = (u32)(v >> 4) & 16383;
								cs = (u32)(v >> 18) & 16383;
								hl2 = 4;
							} else {
								rg = (u32)(v >> 4) & 262143;
								cs = (u32)(v >> 22) & 262143;
								hl2 = 5;
							}
							if (rg == 0 || rg > block_max || hl2 + cs > bs2 || cs < 10 || cum + rg > Z_BLOCK_MAX)
								break;
							pre[4 * n] = q + 3 + hl2; /* jump table, as an offset into the frame */
							pre[4 * n + 1] = cs;
							pre[4 * n + 2] = rg;
							pre[4 * n + 3] = q + 3 + bs2; /* end of the block */
							cum += rg;
							n++;
							q += 3 + bs2;
							if (bh2 & 1)
								break;
						}
						L.misc[ZM_A] = n;
					}
					wv_sync();
					u32 npre = L.misc[ZM_A];
					if (npre) {
						const u32 g = (u32)lane >> 2, sl = (u32)lane & 3;
						bool dec = g <= npre, bad = false;
						u32 jt = (u32)(src - f) + lhl + tree, cs = lcsz - tree, rg = regen, base = 0;
						if (g && dec) {
							jt = pre[4 * (g - 1)];
							cs = pre[4 * (g - 1) + 1];
							rg = pre[4 * (g - 1) + 2];
						}
						base = wv_scan_incl(sl == 0 && dec ? rg : 0u) - rg; /* literals of the blocks before mine */
						u32 s_off = 0, s_len = 0, s_n = 0, s_dst = 0;
						int ipos = 0;
						if (dec) {
							const u8 *jp = f + jt;
							const u32 j1 = (u32)jp[0] | (u32)jp[1] << 8, j2 = (u32)jp[2] | (u32)jp[3] << 8,
								  j3 = (u32)jp[4] | (u32)jp[5] << 8;
							const u32 tot = cs - 6, q4 = (rg + 3) / 4;
							if (j1 + j2 + j3 >= tot || 3 * q4 > rg) {
								bad = true;
							} else {
								s_off = jt + 6 + (sl > 0 ? j1 : 0) + (sl > 1 ? j2 : 0) + (sl > 2 ? j3 : 0);
								s_len = sl == 0 ? j1 : sl == 1 ? j2 : sl == 2 ? j3 : tot - j1 - j2 - j3;
								s_n = sl < 3 ? q4 : rg - 3 * q4;
								s_dst = base + sl * q4;
								const u32 lastb = s_len ? f[s_off + s_len - 1] : 0;
								if (s_len == 0 || lastb == 0)
									bad = true;
								else
									ipos = 8 * (int)(s_len - 1) + hb32(lastb);
							}
						}
						/* every lane keeps a 40-byte window of its own stream in LDS (over the
						 * header stage and the sequence batch, both idle here): 24 symbols a round
						 * (the sixth group of four starts at most 5 x 44 bits = 28 bytes down and
						 * reads 8 bytes from there) */
						u8 *win = L.below + 40u * (u32)lane;
						const int lg = huf_log;
						const u32 hm = (1u << lg) - 1;
						u32 done = 0;
						while (wv_any(dec && !bad && done < s_n)) {
							if (dec && !bad && done < s_n) {
								const int whi = (ipos + 7) >> 3, iwlo = whi - 40;
								const u8 *p = f + s_off + iwlo;
								u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0;
								if (iwlo >= 0) {
									w0 = ld64u(p);
									w1 = ld64u(p + 8);
									w2 = ld64u(p + 16);
									w3 = ld64u(p + 24);
									w4 = ld64u(p + 32);
								} else {
									for (int k = 0; k < 8; k++) {
										if (iwlo + k >= 0)
											w0 |= (u64)p[k] << (8 * k);
										if (iwlo + 8 + k >= 0)
											w1 |= (u64)p[8 + k] << (8 * k);
										if (iwlo + 16 + k >= 0)
											w2 |= (u64)p[16 + k] << (8 * k);
										if (iwlo + 24 + k >= 0)
											w3 |= (u64)p[24 + k] << (8 * k);
										if (iwlo + 32 + k >= 0)
											w4 |= (u64)p[32 + k] << (8 * k);
									}
								}
								*(u64 *)win = w0;
								*(u64 *)(win + 8) = w1;
								*(u64 *)(win + 16) = w2;
								*(u64 *)(win + 24) = w3;
								*(u64 *)(win + 32) = w4;
								const u32 todo = s_n - done < 24 ? s_n - done : 24;
								u8 *dst = lit_scratch + s_dst + done;
								u32 i = 0;
								for (; i + 8 <= todo && ipos >= 0; i += 8) {
									u64 acc = 0;
									ZMT_UNROLL
									for (int hlf = 0; hlf < 2; hlf++) {
										const int tb = (ipos - 1) >> 3;
										const u64 word = ld64u(win + (tb - 7 - iwlo));
										int cb = ipos - 8 * (tb - 7);
										ZMT_UNROLL
										for (int k = 0; k < 4; k++) {
											const u32 e = L.huf[(u32)(word >> (cb - lg)) & hm];
											cb -= (int)(e >> 8);
											acc |= (u64)(e & 255) << (8 * (4 * hlf + k));
										}
										ipos = cb + 8 * (tb - 7);
									}
									st64g(dst + i, acc);
								}
								{
									u64 c = 0;
									int cb = 0;
									for (; i < todo && ipos >= 0; i++) {
										if (cb < lg) {
											const int tb = (ipos - 1) >> 3;
											c = ld64u(win + (tb - 7 - iwlo));
											cb = ipos - 8 * (tb - 7);
										}
										const u32 e = L.huf[(u32)(c >> (cb - lg)) & hm];
										const int nb = (int)(e >> 8);
										cb -= nb;
										ipos -= nb;
										dst[i] = (u8)e;
									}
								}
								if (ipos < 0)
									bad = true;
								done += todo;
								if (!bad && done == s_n && ipos != 0)
									bad = true;
							}
						}
						const u64 bm = wv_ballot(dec && bad);
						if (bm) {
							const u32 gb = (u32)(wv_ffs(bm) - 1) >> 2;
							if (gb == 0) {
								stc = ZBAD();
								break;
							}
							npre = gb - 1; /* the blocks from there on are decoded in their turn */
						}
						pre_left = npre;
						pre_lit = regen;
						lit_done = true;
						unit_head = npre != 0;
						unit_n = npre;
						wv_sync();
						wave_mem_fence();
					}
				}
				if (!lit_done) {
				ZP(0);
				/* ---- Huffman streams: lane s < nstreams decodes stream s ---- */
				const u32 body = lhl + tree; /* offset of jump table / single stream in src */
				if (lcsz < tree + (nstreams == 4 ? 10u : 1u)) {
					stc = ZBAD();
					break;
				}
				u32 s_off = body, s_len = lcsz - tree, s_n = regen, s_dst = 0;
				bool bad = false;
				if (nstreams == 4) {
					const u32 j1 = uld16(src + body), j2 = uld16(src + body + 2), j3 = uld16(src + body + 4);
					const u32 tot = lcsz - tree - 6, q = (regen + 3) / 4;
					if (j1 + j2 + j3 >= tot || 3 * q > regen) {
						stc = ZBAD();
						break;
					}
					const u32 sl = lane & 3;
					s_off = body + 6 + (sl > 0 ? j1 : 0) + (sl > 1 ? j2 : 0) + (sl > 2 ? j3 : 0);
					s_len = sl == 0 ? j1 : sl == 1 ? j2 : sl == 2 ? j3 : tot - j1 - j2 - j3;
					s_n = sl < 3 ? q : regen - 3 * q;
					s_dst = sl * q;
				}
				const bool dec = (u32)lane < nstreams;
				long pos = 0; /* unread bits of my stream */
				if (dec) {
					const u32 lastb = s_len ? src[s_off + s_len - 1] : 0;
					if (s_len == 0 || lastb == 0)
						bad = true;
					else
						pos = 8 * (long)(s_len - 1) + hb32(lastb);
				}
				if (wv_any(dec && bad)) {
					stc = ZBAD();
					break;
				}
				u32 done = 0;
				while (wv_any(dec && done < s_n && !bad)) {
					/* cooperative refill: 16 lanes per stream, 256-byte window ending at the
					 * byte that holds the next unread bit; bytes below the stream read 0 */
					wv_sync();
					{
						const int sg = lane >> 4;
						const u32 g_off = wv_shfl(s_off, sg);
						const u32 g_poslo = wv_shfl((u32)pos, sg);
						const u32 g_poshi = wv_shfl((u32)((u64)pos >> 32), sg);
						const long g_pos = (long)((u64)g_poshi << 32 | g_poslo);
						const long whi = (g_pos + 7) >> 3;
						if ((u32)sg < nstreams) {
							const u32 o = 16u * ((u32)lane & 15);
							const long rel = whi - 256 + (long)o;
							const u8 *p = src + g_off + rel;
							u64 a = 0, b = 0;
							if (rel >= 0 && p + 16 <= mem_hi) {
								a = ld64u(p);
								b = ld64u(p + 8);
							} else {
								for (int k = 0; k < 8; k++) {
									if (rel + k >= 0 && p + k < mem_hi)
										a |= (u64)p[k] << (8 * k);
									if (rel + 8 + k >= 0 && p + 8 + k < mem_hi)
										b |= (u64)p[8 + k] << (8 * k);
								}
							}
							u8 *w = L.stage + 256u * (u32)sg + o;
							*(u64 *)w = a;
							*(u64 *)(w + 8) = b;
						}
					}
					wv_sync();
					if (dec && !bad && done < s_n) {
						const u8 *win = L.stage + 256u * (u32)lane; /* byte (whi-256+i) of my stream */
						const long wlo = ((pos + 7) >> 3) - 256;
						const u32 todo = s_n - done < 128 ? s_n - done : 128;
						u8 *dst = lit_scratch + s_dst + done;
						const int lg = huf_log;
						const u32 hm = (1u << lg) - 1;
						int ipos = (int)pos; /* a stream holds < 2^21 bits */
						const int iwlo = (int)wlo;
						u32 i = 0;
						/* eight symbols per store; the container is refilled before every four
						 * (>= 57 fresh bits cover 4 x 11), so the symbol steps carry no branch */
						for (; i + 8 <= todo && ipos >= 0; i += 8) {
							u64 acc = 0;
							ZMT_UNROLL
							for (int hlf = 0; hlf < 2; hlf++) {
								const int tb = (ipos - 1) >> 3;
								const u64 word = ld64u(win + (tb - 7 - iwlo));
								int cb = ipos - 8 * (tb - 7);
								ZMT_UNROLL
								for (int k = 0; k < 4; k++) {
									const u32 e = L.huf[(u32)(word >> (cb - lg)) & hm];
									cb -= (int)(e >> 8);
									acc |= (u64)(e & 255) << (8 * (4 * hlf + k));
								}
								ipos = cb + 8 * (tb - 7);
							}
							st64g(dst + i, acc);
						}
						/* tail: fewer than eight symbols left in this round */
						{
							u64 c = 0;
							int cb = 0;
							for (; i < todo && ipos >= 0; i++) {
								if (cb < lg) {
									const int tb = (ipos - 1) >> 3;
									c = ld64u(win + (tb - 7 - iwlo));
									cb = ipos - 8 * (tb - 7);
								}
								const u32 e = L.huf[(u32)(c >> (cb - lg)) & hm];
								const int nb = (int)(e >> 8);
								cb -= nb;
								ipos -= nb;
								dst[i] = (u8)e;
							}
						}
						if (ipos < 0)
							bad = true; /* ran past the first bit of the stream */
						pos = ipos;
						done += todo;
						if (!bad && done == s_n && pos != 0)
							bad = true;
					}
				}
				if (wv_any(dec && bad)) {
					stc = ZBAD();
					break;
				}
				wave_mem_fence(); /* literals are read back by other lanes below */
				} /* !lit_done */
			}

			ZP(1);
			/* ---- sequences section header + table descriptions, from LDS ---- */
			const u32 sq0 = lhl + lcsz; /* offset of the sequences section in src */
			if (sq0 >= bsize) {
				stc = ZBAD();
				break;
			}
			/* decoded ahead by the pre-pass: only Number_of_Sequences is read here; the tables of this block are then
			 * not built (the pre-pass marks a prefix of the frame's blocks and leaves the decoder a block that describes
			 * all three tables itself, zstd_dec_seq.hip) */
			const u32 pre_at = (bi < pre_nhdr) ? wv_readfirst(pre_hdr[bi]) : 0u;
			const u64 *pseq = nullptr;
			u32 nseq, sq_hdr;
			bool all_pre = false, all_fit = false;
			if (pre_at) {
				const u32 s4 = uld32(src + sq0);
				nseq = s4 & 255;
				if (nseq == 255)
					nseq = ((s4 >> 8) & 0xFFFF) + 0x7F00;
				else if (nseq >= 128)
					nseq = ((nseq - 128) << 8) + ((s4 >> 8) & 255);
				sq_hdr = 0;
				pseq = pre_seq + (pre_at - 1);
				unit_head = false;
				sq_left = 0;
				if (lane < 3)
					my_tab_ok = false; /* what a repeat mode would refer to is not in this wave's LDS */
			} else {
			wv_sync();
			stage_load(L.stage, src, (long)sq0, 512, 0, mem_lo, mem_hi, lane);
			wv_sync();
			if (lane == 0) {
				const u8 *d = L.stage;
				const u32 avail = bsize - sq0 < 512 ? bsize - sq0 : 512;
				u32 p = 1, err = 0, nseq = d[0], modes = 0;
				if (nseq >= 128) {
					if (nseq == 255) {
						nseq = (d[1] | d[2] << 8) + 0x7F00;
						p = 3;
					} else {
						nseq = ((nseq - 128) << 8) + d[1];
						p = 2;
					}
				}
				if (p > avail)
					err = 1;
				if (!err && nseq) {
					if (p >= avail)
						err = 1;
					modes = d[p++];
					if (modes & 3)
						err = 1;
					for (int t = 0; t < 3 && !err; t++) {
						const int mode = (modes >> (6 - 2 * t)) & 3;
						const int max_sym = t == 0 ? 36 : t == 1 ? 32 : 53, max_log = t == 1 ? 8 : 9;
						L.misc[ZM_A + t] = 0xFFFFFFFFu; /* table stays as it is */
						if (mode == 0) {
							const short *def = t == 0 ? Z_LL_DEF : t == 1 ? Z_OF_DEF : Z_ML_DEF;
							const int n = t == 0 ? 36 : t == 1 ? 29 : 53;
							for (int i = 0; i < n; i++)
								L.norm[t][i] = def[i];
							L.misc[ZM_A + t] = (u32)n | (u32)(t == 1 ? 5 : 6) << 8 | 0x40000000u; /* predefined */
						} else if (mode == 1) {
							if (p >= avail || d[p] >= max_sym)
								err = 1;
							else
								L.misc[ZM_A + t] = 0x80000000u | d[p++]; /* RLE */
						} else if (mode == 2) {
							int nsym = 0, lg = 0;
							const int used = fse_read_ncount(d + p, avail - p, L.norm[t], max_sym, max_log, &nsym, &lg);
							if (used < 0) {
								err = 1;
							} else {
								p += (u32)used;
								L.misc[ZM_A + t] = (u32)nsym | (u32)lg << 8;
							}
						}
					}
				}
				L.misc[ZM_ERR] = err;
				L.misc[ZM_D] = nseq;
				L.misc[ZM_E] = p;
			}
			wv_sync();
			if (L.misc[ZM_ERR]) {
				stc = ZBAD();
				break;
			}
			nseq = L.misc[ZM_D];
			sq_hdr = L.misc[ZM_E];
			/* what the device encoder writes at the head of a unit: three predefined tables (the
			 * blocks behind it say "predefined" again), or three described ones (the blocks behind
			 * it say "repeat") */
			all_pre = nseq && (L.misc[ZM_A] & 0xC0000000u) == 0x40000000u &&
				  (L.misc[ZM_A + 1] & 0xC0000000u) == 0x40000000u &&
				  (L.misc[ZM_A + 2] & 0xC0000000u) == 0x40000000u;
			all_fit = nseq && L.misc[ZM_A] < 0x40000000u && L.misc[ZM_A + 1] < 0x40000000u &&
				  L.misc[ZM_A + 2] < 0x40000000u;
			}
			const u32 unit_modes = all_fit ? 0xFCu : 0u; /* Symbol_Compression_Modes of the unit's other blocks */
			u32 lpos = 0; /* literals consumed */
			if (nseq) {
				int ll_log = 0, of_log = 0, ml_log = 0;
				u32 bs_off = 0;
				long pos = 0;
				if (!pseq) {
				/* ---- build the three tables on lanes 0..2 ---- */
				bool terr = false, too_big = false;
				if (lane < 3) {
					const u32 spec = L.misc[ZM_A + lane];
					u32 *cells = lane == 0 ? L.ll : lane == 1 ? L.of : L.ml;
					if (spec == 0xFFFFFFFFu) {
						terr = !my_tab_ok; /* repeat mode needs a previous table */
					} else if (spec & 0x80000000u) {
						const u32 sy = spec & 255; /* RLE: one cell, no state bits */
						cells[0] = sy | (lane == 1 ? sy : (lane == 0 ? L.llx[sy] : L.mlx[sy]) >> 24) << 10;
						my_tab_log = 0;
						my_tab_ok = true;
						my_tab_pre = false;
					} else if ((spec & 0x40000000u) && my_tab_pre && my_tab_ok) {
						/* predefined table still in place from an earlier block: nothing to build */
					} else {
						const int lg = (int)((spec >> 8) & 255);
						if (lg > (lane == 0 ? LDS::llcap : lane == 1 ? LDS::ofcap : LDS::mlcap)) {
							too_big = true;
						} else {
						terr = fse_build(cells, L.norm[lane], (int)(spec & 255), lg, L.next[lane],
								 lane == 0 ? L.llx : lane == 2 ? L.mlx : (const u32 *)nullptr, lane) != 0;
						my_tab_log = lg;
						my_tab_ok = !terr;
						my_tab_pre = (spec & 0x40000000u) != 0;
						}
					}
				}
				wv_sync();
				if (wv_any(too_big)) {
					stc = ST_NEEDS_GENERAL;
					break;
				}
				if (wv_any(terr)) {
					stc = ZBAD();
					break;
				}
				ll_log = (int)wv_readlane((u32)my_tab_log, 0);
				of_log = (int)wv_readlane((u32)my_tab_log, 1);
				ml_log = (int)wv_readlane((u32)my_tab_log, 2);
				ZP(2);
				/* ---- FSE bitstream: src[sq0 + sq_hdr, bsize) read backwards ---- */
				bs_off = sq0 + sq_hdr;
				if (bs_off >= bsize) {
					stc = ZBAD();
					break;
				}
				const u32 bs_len = bsize - bs_off;
				const u32 lastb = uld8(src + bs_off + bs_len - 1);
				if (lastb == 0) {
					stc = ZBAD();
					break;
				}
				pos = 8 * (long)(bs_len - 1) + hb32(lastb); /* unread bits, wave-uniform */
				/* ---- unit: the sequences of this block and of the blocks looked at above ----
				 * Their bitstreams are independent and, with one set of tables for the unit, read
				 * through the same three tables: four lanes per block (LL / OF / ML state + one
				 * idle) decode up to 16 blocks side by side into the record's scratch, 8 bytes a
				 * sequence (ll | ml << 18 | offset value << 36); the blocks then only execute.
				 * As with the literals, anything odd ends the unit before that block. */
				if (unit_head) {
					unit_head = false;
					sq_left = 0;
					if ((all_pre || all_fit) && unit_n) {
						const u32 *pre = (const u32 *)L.w;
						const u32 g = (u32)lane >> 2, sl = (u32)lane & 3;
						bool act = g <= unit_n, gbad = false;
						u32 g_off = (u32)(src - f) + bs_off, g_len = bs_len, g_n = nseq;
						if (g && act) {
							const u32 s0 = pre[4 * (g - 1)] + pre[4 * (g - 1) + 1], bend = pre[4 * (g - 1) + 3];
							g_n = 0;
							if (bend - s0 >= 4) {
								const u32 b0 = f[s0];
								u32 p = 1;
								g_n = b0;
								if (b0 == 255) {
									g_n = ((u32)f[s0 + 1] | (u32)f[s0 + 2] << 8) + 0x7F00u;
									p = 3;
								} else if (b0 >= 128) {
									g_n = ((b0 - 128) << 8) + f[s0 + 1];
									p = 2;
								}
								if (f[s0 + p] != unit_modes)
									g_n = 0; /* a table of its own */
								g_off = s0 + p + 1;
								if (g_off >= bend)
									g_n = 0;
								else
									g_len = bend - g_off;
							}
							if (g_n == 0)
								gbad = true;
						}
						int bp = 0; /* unread bits of my block's stream */
						if (act && !gbad) {
							const u32 lb = f[g_off + g_len - 1];
							if (lb == 0)
								gbad = true;
							else
								bp = 8 * (int)(g_len - 1) + hb32(lb);
							if (bp < ll_log + of_log + ml_log)
								gbad = true;
						}
						u32 ngrp = unit_n + 1;
						{
							const u64 bm = wv_ballot(act && gbad);
							if (bm)
								ngrp = (u32)(wv_ffs(bm) - 1) >> 2;
						}
						act = g < ngrp;
						const u32 incl = wv_scan_incl(sl == 0 && act ? g_n : 0u);
						{
							const u64 om = wv_ballot(act && incl > Z_SEQCAP);
							if (om)
								ngrp = (u32)(wv_ffs(om) - 1) >> 2;
						}
						act = g < ngrp;
						u64 *myseq = seq_scratch + (incl - g_n);
						if (ngrp >= 2) {
							const u32 *mytab = sl == 1 ? L.of : sl == 2 ? L.ml : L.ll;
							const u32 mylog = sl == 0 ? (u32)ll_log : sl == 1 ? (u32)of_log : sl == 2 ? (u32)ml_log : 0u;
							const u32 tmask = (1u << mylog) - 1;
							const u32 e_of = sl == 1 ? 0u : ~0u, e_ml = sl == 0 ? ~0u : 0u;
							const u32 s_ll = sl == 0 ? 0u : ~0u, s_ml = sl == 1 ? ~0u : 0u;
							u8 *gwin = L.below + 160u * g; /* 160-byte window per block, 40 per lane */
							u32 state = 0, done = 0;
							bool first = true;
							while (wv_any(act && done < g_n)) {
								const int whi = (bp + 7) >> 3, wlo = whi - 160;
								wv_sync();
								if (act && done < g_n) {
									const int rel = (int)g_off + wlo + 40 * (int)sl;
									u64 w[5] = {0, 0, 0, 0, 0};
									if (rel >= 0) {
										ZMT_UNROLL
										for (int j = 0; j < 5; j++)
											w[j] = ld64u(f + rel + 8 * j);
									} else {
										for (int j = 0; j < 5; j++)
											for (int k2 = 0; k2 < 8; k2++)
												if (rel + 8 * j + k2 >= 0)
													w[j] |= (u64)f[rel + 8 * j + k2] << (8 * k2);
									}
									ZMT_UNROLL
									for (int j = 0; j < 5; j++)
										*(u64 *)(gwin + 40u * sl + 8u * (u32)j) = w[j];
								}
								wv_sync();
								const u8 *winb = gwin - wlo - 15; /* winb[b + 15] = byte b of my stream */
								if (first) {
									first = false;
									const int tb = (bp - 1) >> 3;
									const u64 w1 = ld64u(winb + tb), w0 = ld64u(winb + tb + 8);
									const u32 skip = (u32)(8 * (tb + 1) - bp);
									state = xbits(w0, w1, skip + (sl == 0 ? 0u : sl == 1 ? (u32)ll_log : (u32)(ll_log + of_log)),
										      mylog);
									bp -= ll_log + of_log + ml_log;
								}
								for (;;) {
									/* eight sequences of every block; lane sl keeps 2 sl, 2 sl + 1 */
									u64 r0 = 0, r1 = 0;
									ZMT_UNROLL
									for (int i = 0; i < 8; i++) {
										const bool on = act && done + (u32)i < g_n;
										int tb = (bp - 1) >> 3;
										tb = tb < wlo + 15 ? wlo + 15 : tb; /* a stream gone bad stays inside its window */
										u64 w1, w0;
										lds_ld128(winb + tb, w1, w0);
										const u32 cell = mytab[state & tmask];
										const u32 skip = (u32)(8 * (tb + 1) - bp) & 127u;
										const u32 nb = done + (u32)i + 1 == g_n ? 0u : ZC_NB(cell), ab = ZC_AB(cell);
										const u32 pk = ab | nb << 8;
										const u32 p_ll = wv_quad(pk, 0), p_of = wv_quad(pk, 1), p_ml = wv_quad(pk, 2);
										const u32 a_ll = p_ll & 255, n_ll = p_ll >> 8, a_of = p_of & 255, n_of = p_of >> 8;
										const u32 a_ml = p_ml & 255, n_ml = p_ml >> 8;
										const u32 base3 = skip + a_of + a_ml + a_ll;
										const u32 eo = skip + (a_of & e_of) + (a_ml & e_ml);
										const u32 so = base3 + (n_ll & s_ll) + (n_ml & s_ml);
										const u32 extra = xbits(w0, w1, eo, ab);
										const u32 sbits = xbits(w0, w1, so, nb);
										const u32 sym = ZC_SYM(cell);
										const u32 val = sl == 0   ? (L.llx[sym < 36 ? sym : 0] & 0xFFFFFFu) + extra
												: sl == 2 ? (L.mlx[sym < 53 ? sym : 0] & 0xFFFFFFu) + extra
													  : (1u << (sym & 31)) + extra;
										if (on && sl == 1 && sym > 27)
											gbad = true; /* does not fit the packing: left to the block itself */
										const u32 v_ll = wv_quad(val, 0), v_of = wv_quad(val, 1), v_ml = wv_quad(val, 2);
										const u64 rcd = (u64)v_ll | (u64)v_ml << 18 | (u64)v_of << 36;
										if ((u32)(i >> 1) == sl) {
											if (i & 1)
												r1 = rcd;
											else
												r0 = rcd;
										}
										state = on ? ZC_BASE(cell) + sbits : state;
										bp -= on ? (int)(base3 - skip + n_ll + n_ml + n_of) : 0;
									}
									if (act && done + 2 * sl < g_n)
										st64g((u8 *)(myseq + done + 2 * sl), r0);
									if (act && done + 2 * sl + 1 < g_n)
										st64g((u8 *)(myseq + done + 2 * sl + 1), r1);
									if (act && done < g_n) {
										done = g_n - done < 8 ? g_n : done + 8;
										if (bp < 0 || (done == g_n && bp != 0))
											gbad = true;
										if (gbad)
											act = false;
									}
									const bool more = act && done < g_n;
									/* 8 x 76 more bits and the 16-byte read must stay inside the 160 bytes */
									if (!wv_any(more) || wv_any(more && 8 * whi - bp > 544))
										break;
								}
							}
							wave_mem_fence();
							{
								const u64 bm = wv_ballot(gbad && g < ngrp);
								if (bm) {
									const u32 gb = (u32)(wv_ffs(bm) - 1) >> 2;
									ngrp = gb < ngrp ? gb : ngrp;
								}
							}
							sq_left = ngrp;
							sq_pos = 0;
#ifdef ZMT_EMU
							if (getenv("ZMT_EMU_DEBUG") && lane == 0)
								fprintf(stderr, "zstd_dec: unit of %u blocks decoded side by side (%s tables)\n", ngrp,
									all_fit ? "described" : "predefined");
#endif
							wv_sync();
						}
					}
				}
				} /* !pseq */
				const bool from_scr = sq_left != 0 || pseq != nullptr;
				/* lanes 0 / 1 / 2 carry the LL / OF / ML state and decode their own code; the
				 * loop below is wave-uniform (one LDS round trip per sequence): every lane reads
				 * its state's cell and the same 128-bit window of the bitstream, the six field
				 * widths are exchanged through SGPRs, each lane cuts its two fields out */
				const u32 *mytab = lane == 1 ? L.of : lane == 2 ? L.ml : L.ll;
				u64 *myval = L.sq[lane < 3 ? lane : 0];
				/* where a lane's two fields start, as masks over the widths of the others: the
				 * stream order is OF extra, ML extra, LL extra, LL state, ML state, OF state */
				const u32 e_of = lane == 1 ? 0u : ~0u, e_ml = lane == 0 ? ~0u : 0u;
				const u32 s_ll = lane == 0 ? 0u : ~0u, s_ml = lane == 1 ? ~0u : 0u;
				u32 state = 0;
				for (u32 sbase = 0; sbase < nseq && stc == ST_OK; sbase += 64) {
					const u32 k = nseq - sbase < 64 ? nseq - sbase : 64;
					const bool act0 = (u32)lane < k;
					u32 ll = 0, ml = 0, ofv = 4;
					if (!from_scr) {
					/* window of the bitstream: 1 KiB ending at the byte of the next unread bit */
					const long whi = (pos + 7) >> 3, wlo = whi - (long)Z_STAGE;
					wv_sync();
					stage_load(L.stage, src + bs_off, wlo, Z_STAGE, -(long)(bs_off + 16), mem_lo, mem_hi, lane);
					wv_sync();
					const u8 *win = L.stage - wlo; /* win[b] = byte b of the bitstream */
					bool err = false;
					if (sbase == 0) {
						/* initial states: LL, OF, ML (RFC 8878 3.1.1.3.2.1.1) */
						if (pos < (long)(ll_log + of_log + ml_log)) {
							stc = ZBAD();
							break;
						}
						const long tb = (pos - 1) >> 3;
						const u64 w0 = ld64u(win + tb - 7), w1 = ld64u(win + tb - 15);
						const u32 skip = (u32)(8 * (tb + 1) - pos);
						state = xbits(w0, w1, skip + (lane == 0 ? 0u : lane == 1 ? (u32)ll_log : (u32)(ll_log + of_log)),
							      (u32)(lane < 3 ? my_tab_log : 0));
						pos -= ll_log + of_log + ml_log;
					}
					int bp = (int)pos; /* a sequences bitstream holds < 2^21 bits */
					const u8 *winb = win - 15;
					for (u32 i = 0; i < k; i++) {
						/* bp == 0 is legal here: the last sequence may need no bits at all; a
						 * negative bp is caught after the loop (the reads then come from the
						 * slack below the window).  Both LDS reads of the step are issued
						 * together: one round trip per sequence. */
						const int tb = (bp - 1) >> 3;
						/* only the three state lanes read the window: a misaligned 8-byte LDS read costs the CU's LDS
						 * pipe one cycle per ACTIVE lane (tools/ubench/lds_cost.hip) -- with all 64 lanes reading the
						 * same two addresses this loop was bound by that pipe, 128 cycles per sequence and wave */
						u64 w1 = 0, w0 = 0;
						if (lane < 3) {
							w1 = ld64u(winb + tb);
							w0 = ld64u(winb + tb + 8);
						}
						const u32 cell = mytab[state];
						const u32 skip = (u32)(8 * (tb + 1) - bp);
						const u32 nb = sbase + i + 1 == nseq ? 0u : ZC_NB(cell), ab = ZC_AB(cell);
						const u32 pk = ab | nb << 8;
						const u32 p_ll = wv_readlane(pk, 0), p_of = wv_readlane(pk, 1), p_ml = wv_readlane(pk, 2);
						const u32 a_ll = p_ll & 255, n_ll = p_ll >> 8, a_of = p_of & 255, n_of = p_of >> 8;
						const u32 a_ml = p_ml & 255, n_ml = p_ml >> 8;
						const u32 base3 = skip + a_of + a_ml + a_ll;
						const u32 eo = skip + (a_of & e_of) + (a_ml & e_ml);
						const u32 so = base3 + (n_ll & s_ll) + (n_ml & s_ml);
						const u32 extra = xbits(w0, w1, eo, ab);
						const u32 sbits = xbits(w0, w1, so, nb);
						if (lane < 3)
							myval[i] = (u64)extra | (u64)ZC_SYM(cell) << 32;
						state = lane < 3 ? ZC_BASE(cell) + sbits : 0;
						bp -= (int)(base3 - skip + n_ll + n_ml + n_of);
					}
					pos = bp;
					if (err || pos < 0 || (sbase + k == nseq && pos != 0)) {
						stc = ZBAD();
						break;
					}
					wv_sync();
					ZP(3);
					/* ---- execute the k sequences ---- */
					if (act0) {
						const u64 q_ll = L.sq[0][lane], q_of = L.sq[1][lane], q_ml = L.sq[2][lane];
						const u32 c_of = (u32)(q_of >> 32);
						ll = (L.llx[(u32)(q_ll >> 32)] & 0xFFFFFFu) + (u32)q_ll;
						ml = (L.mlx[(u32)(q_ml >> 32)] & 0xFFFFFFu) + (u32)q_ml;
						ofv = c_of > 31 ? 0u : (1u << c_of) + (u32)q_of;
					}
					} else if (act0) {
						const u64 v = (pseq ? pseq : seq_scratch + sq_pos)[sbase + (u32)lane];
						ll = (u32)v & 0x3FFFFu;
						ml = (u32)(v >> 18) & 0x3FFFFu;
						ofv = (u32)(v >> 36);
					}
					if (wv_any(act0 && ofv == 0)) {
						stc = ZBAD(); /* offset code > 31 */
						break;
					}
					/* repeat offsets (RFC 8878 3.1.1.5): runs of new offsets fold into the history in
					 * one step, only the sequences that use a repeat code are walked one by one */
					u32 off = ofv - 3;
					{
						u64 repm = wv_ballot(act0 && ofv <= 3);
						u32 prev = 0;
						bool rerr = false;
						for (;;) {
							const u32 j = repm ? (u32)wv_ffs(repm) - 1 : k;
							const u32 m = j - prev;
							if (m >= 1) {
								const u32 o1 = wv_readlane(off, (int)(j - 1));
								u32 n1, n2;
								if (m >= 3) {
									n1 = wv_readlane(off, (int)(j - 2));
									n2 = wv_readlane(off, (int)(j - 3));
								} else if (m == 2) {
									n1 = wv_readlane(off, (int)(j - 2));
									n2 = rep0;
								} else {
									n1 = rep0;
									n2 = rep1;
								}
								rep2 = n2;
								rep1 = n1;
								rep0 = o1;
							}
							if (j >= k)
								break;
							const u32 idx = wv_readlane(ofv, (int)j) - 1 + (wv_readlane(ll, (int)j) == 0);
							u32 o = rep0;
							if (idx) {
								o = idx == 1 ? rep1 : idx == 2 ? rep2 : rep0 - 1;
								if (o == 0)
									rerr = true;
								if (idx > 1)
									rep2 = rep1;
								rep1 = rep0;
								rep0 = o;
							}
							if ((u32)lane == j)
								off = o;
							repm &= repm - 1;
							prev = j + 1;
						}
						if (rerr) {
							stc = ZBAD();
							break;
						}
					}
					const u32 len = ll + ml;
					const u32 incl = wv_scan_incl(len), lincl = wv_scan_incl(ll);
					const u32 tot = wv_readlane(incl, 63), ltot = wv_readlane(lincl, 63);
					/* 64 x (2 x 131074) cannot wrap 32 bits */
					if (ltot > regen - lpos || tot > cap - opos || opos + tot - bstart > block_max) {
						stc = ZBAD();
						break;
					}
					const u32 op = opos + incl - len, mpos = op + ll, lsrc = lpos + lincl - ll;
					if (wv_any(act0 && off > mpos)) {
						stc = ZBAD(); /* reaches before the start of the frame */
						break;
					}
					const u32 src_pos = mpos - off, eff = ml < off ? ml : off;
					const u64 longm = wv_ballot(act0 && (ll > Z_CAP || ml > Z_CAP));
					/* literals: independent of everything in this batch */
					if (act0 && ll <= Z_CAP)
						g_copy(out + op, lit + lsrc, ll);
					{
						u64 m = wv_ballot(act0 && ll > Z_CAP);
						while (m) {
							const int j = wv_ffs(m) - 1;
							m &= m - 1;
							wave_copy(out + wv_readlane(op, j), lit + wv_readlane(lsrc, j), wv_readlane(ll, j), lane);
						}
					}
					wave_mem_fence();
					ZP(4);
					/* matches: watermark rounds.  W = everything below is written and visible */
					{
						bool fin = !act0;
						for (;;) {
							const u64 unf = wv_ballot(!fin);
							if (!unf)
								break;
							const int fst = wv_ffs(unf) - 1;
							const u32 W = wv_readlane(mpos, fst);
							if ((longm >> fst) & 1 && wv_readlane(ml, fst) > Z_CAP) {
								/* long match at the head of the queue: whole wave */
								wave_match(out + W, wv_readlane(off, fst), wv_readlane(ml, fst), lane);
								if (lane == fst)
									fin = true;
							} else {
								const bool ready = !fin && ml <= Z_CAP && src_pos + eff <= W;
								if (ready) {
									g_match(out + mpos, off, ml);
									fin = true;
								}
							}
							wave_mem_fence();
						}
					}
					ZP(5);
					opos += tot;
					lpos += ltot;
				}
				if (stc != ST_OK)
					break;
				if (sq_left) {
					sq_pos += nseq;
					sq_left--;
				}
			} else if (sq_hdr != bsize - sq0) {
				stc = ZBAD();
				break;
			}
			/* literals after the last sequence */
			{
				const u32 restl = regen - lpos;
				if (restl > cap - opos || opos + restl - bstart > block_max) {
					stc = ZBAD();
					break;
				}
				wave_copy(out + opos, lit + lpos, restl, lane);
				opos += restl;
			}
			wave_mem_fence();
			unit_head = false;
			ip += bsize;
		}
		bi++;
		if (last)
			break;
	}
	ZP(6);
#ifndef ZMT_EMU
	if (PROF && prof && lane == 0) {
		for (int i = 0; i < (PROF ? 7 : 1); i++)
			atomicAdd(prof + i, (unsigned long long)pc[i]);
		atomicAdd(prof + 8, (unsigned long long)(ZT() - t_begin));
		atomicAdd(prof + 9, 1ull);
	}
#endif
	/* a frame that states its content size must produce exactly that (= the caller's out_len); one
	 * that does not (streaming writers; never zstd-mt) was given a capacity and reports its size */
	if (stc == ST_OK && content != ~0ull && opos != cap)
		stc = ST_SIZE_MISMATCH;
	if (stc == ST_OK && content == ~0ull && lane == 0)
		out_len[rec] = opos;
	if (stc == ST_OK && has_chk) {
		/* Content_Checksum: low 32 bits of XXH64 of the content (RFC 8878 3.1.1); compared by
		 * zmt_xxh64_verify_kernel once the content is complete.  zstd-mt never writes one
		 * (one-shot ZSTD_compress, lib/zstd-mt_compress.c:285); the zstd CLI does. */
		if (flen - ip < 4) {
			stc = ST_BAD_FRAME;
		} else {
			if (lane == 0) {
				chk_expect[rec] = ld32u(f + ip);
				chk_valid[rec] = 1;
			}
			ip += 4;
		}
	}
	if (stc == ST_OK && ip != flen)
		stc = ST_TRAILING;
	if (lane == 0 && (stc != ST_OK || want_status != ST_OK))
		status[rec] = stc;
}

#ifdef ZMT_EMU
#define ZMT_WAVES4
#else
#ifndef ZD_WAVES
#define ZD_WAVES 4 /* 128 VGPRs: 16 waves per CU */
#endif
#define ZMT_WAVES4 __attribute__((amdgpu_waves_per_eu(ZD_WAVES, ZD_WAVES)))
#endif
/* records whose status is `want` (GPUMT_ST_OK after the probe, or ST_NEEDS_GENERAL after the small
 * variant) are decoded; their status becomes OK / an error / ST_NEEDS_GENERAL (small variant only) */
extern "C" __global__ void __launch_bounds__(64) ZMT_WAVES4
zmt_zstd_dec_small_kernel(const u8 *__restrict__ stream, u64 stream_bytes, const u64 *__restrict__ rec_off,
			  const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
			  const u64 *__restrict__ out_off, u32 *__restrict__ out_len,
			  u8 *__restrict__ litbuf, u32 *__restrict__ status, u32 *__restrict__ chk_expect,
			  u32 *__restrict__ chk_valid, u8 *seqbuf, u64 seqcap)
{
	__shared__ __attribute__((aligned(16))) ZLdsSmall L;
	zstd_dec_body<false>(L, (u32)ST_OK, stream, stream_bytes, rec_off, rec_len, nrec, out_base, out_off, out_len,
			     litbuf, status, chk_expect, chk_valid, nullptr, seqbuf, seqcap);
}

extern "C" __global__ void __launch_bounds__(64)
zmt_zstd_dec_kernel(const u8 *__restrict__ stream, u64 stream_bytes, const u64 *__restrict__ rec_off,
		    const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
		    const u64 *__restrict__ out_off, u32 *__restrict__ out_len,
		    u8 *__restrict__ litbuf, u32 *__restrict__ status, u32 *__restrict__ chk_expect,
		    u32 *__restrict__ chk_valid, u32 want, u8 *seqbuf, u64 seqcap)
{
	__shared__ __attribute__((aligned(16))) ZLds L;
	zstd_dec_body<false>(L, want, stream, stream_bytes, rec_off, rec_len, nrec, out_base, out_off, out_len, litbuf,
			     status, chk_expect, chk_valid, nullptr, seqbuf, seqcap);
}

#ifndef ZMT_EMU
/* same kernel with per-phase cycle counters (developer tool) */
extern "C" __global__ void __launch_bounds__(64)
zmt_zstd_dec_kernel_prof(const u8 *__restrict__ stream, u64 stream_bytes, const u64 *__restrict__ rec_off,
			 const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
			 const u64 *__restrict__ out_off, u32 *__restrict__ out_len,
			 u8 *__restrict__ litbuf, u32 *__restrict__ status, u32 *__restrict__ chk_expect,
			 u32 *__restrict__ chk_valid, u32 want, unsigned long long *prof, u8 *seqbuf, u64 seqcap)
{
	__shared__ __attribute__((aligned(16))) ZLds L;
	zstd_dec_body<true>(L, want, stream, stream_bytes, rec_off, rec_len, nrec, out_base, out_off, out_len, litbuf,
			    status, chk_expect, chk_valid, prof, seqbuf, seqcap);
}
#endif

/* ------------------------------------------------------------------ XXH64 content checksum
 * Four lanes per record = the four accumulators of XXH64 (one 32-byte stripe per step); only records
 * that carry a checksum do any work. */
#define X64_1 0x9E3779B185EBCA87ull
#define X64_2 0xC2B2AE3D27D4EB4Full
#define X64_3 0x165667B19E3779F9ull
#define X64_4 0x85EBCA77C2B2AE63ull
#define X64_5 0x27D4EB2F165667C5ull
static __device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
static __device__ __forceinline__ u64 x64_round(u64 acc, u64 in) { return rotl64(acc + in * X64_2, 31) * X64_1; }
static __device__ __forceinline__ u64 x64_merge(u64 h, u64 v) { return (h ^ x64_round(0, v)) * X64_1 + X64_4; }
static __device__ __forceinline__ u64 shfl64(u64 v, int src)
{
	return (u64)wv_shfl((u32)(v >> 32), src) << 32 | wv_shfl((u32)v, src);
}

extern "C" __global__ void __launch_bounds__(256)
zmt_xxh64_verify_kernel(const u8 *__restrict__ base, const u64 *__restrict__ off,
			const u32 *__restrict__ len, u32 n, const u32 *__restrict__ expect,
			const u32 *__restrict__ valid, u32 *__restrict__ status)
{
	const u32 gtid = blockIdx.x * 256 + threadIdx.x;
	const u32 item = gtid >> 2, a = gtid & 3;
	const int lane = wv_lane();
	const bool live = item < n && valid[item] && status[item] == ST_OK;
	const u8 *p = live ? base + off[item] : base;
	const u32 N = live ? len[item] : 0;
	u64 v = a == 0 ? X64_1 + X64_2 : a == 1 ? X64_2 : a == 2 ? 0ull : 0ull - X64_1;
	if (N >= 32) {
		const u8 *q = p + 8 * a;
		for (u32 s = 0; s < (N >> 5); s++) {
			v = x64_round(v, ld64u(q));
			q += 32;
		}
	}
	/* all four accumulators to lane 0 of the quad (wave-uniform shuffles) */
	const int q0 = lane & ~3;
	const u64 v1 = shfl64(v, q0), v2 = shfl64(v, q0 + 1), v3 = shfl64(v, q0 + 2), v4 = shfl64(v, q0 + 3);
	if (live && a == 0) {
		u64 h;
		if (N >= 32) {
			h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
			h = x64_merge(h, v1);
			h = x64_merge(h, v2);
			h = x64_merge(h, v3);
			h = x64_merge(h, v4);
		} else {
			h = X64_5;
		}
		h += N;
		const u8 *t = p + (N & ~31u), *e = p + N;
		for (; t + 8 <= e; t += 8)
			h = rotl64(h ^ x64_round(0, ld64u(t)), 27) * X64_1 + X64_4;
		if (t + 4 <= e) {
			h = rotl64(h ^ (u64)ld32u(t) * X64_1, 23) * X64_2 + X64_3;
			t += 4;
		}
		for (; t < e; t++)
			h = rotl64(h ^ (u64)*t * X64_5, 11) * X64_1;
		h ^= h >> 33;
		h *= X64_2;
		h ^= h >> 29;
		h *= X64_3;
		h ^= h >> 32;
		if ((u32)h != expect[item])
			status[item] = ST_BAD_CHECKSUM;
	}
}

/* out_len[i] = Frame_Content_Size of record i (what the host needs before it can size d_out);
 * status[i] = ST_OK, or why the record cannot be decoded here. */
extern "C" __global__ void __launch_bounds__(256)
zmt_zstd_probe_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
		      const u32 *__restrict__ rec_len, u32 nrec, u32 *__restrict__ out_len,
		      u32 *__restrict__ status)
{
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i >= nrec)
		return;
	const u8 *r = stream + rec_off[i];
	const u32 rlen = rec_len[i];
	u32 st = ST_OK, n = 0;
	if (rlen < 18 || ld32u(r) != ZMT_SKIP_MAGIC || ld32u(r + 4) != 4 || ld32u(r + 8) != rlen - 12) {
		st = ST_BAD_RECORD;
	} else if (ld32u(r + 12) != ZMT_ZSTD_MAGIC) {
		st = ST_BAD_FRAME;
	} else {
		const u8 *f = r + 12;
		const u32 flen = rlen - 12, fhd = f[4], fcs = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
		const u32 did_len = did == 3 ? 4 : did, fcs_len = fcs == 0 ? single : (1u << fcs);
		const u32 hp = 5 + (1 - single) + did_len;
		if ((fhd & 8) || flen < hp + fcs_len) {
			st = ST_BAD_FRAME;
		} else if (fcs_len == 0) {
			st = ST_UNSUPPORTED; /* no content size: never written by zstd-mt (one-shot ZSTD_compress) */
		} else {
			u64 c = 0;
			for (u32 k = 0; k < fcs_len; k++)
				c |= (u64)f[hp + k] << (8 * k);
			if (fcs == 1)
				c += 256;
			if (c > 0x7FFFFFFFull)
				st = ST_UNSUPPORTED;
			else
				n = (u32)c;
		}
	}
	out_len[i] = n;
	status[i] = st;
}
