/*
 * lz4_dec_split.hip -- LZ4 frame decoder, first two kernels of the three-kernel pipeline.
 *
 * Same contract as the serial decoder (replaces LZ4F_decompress at
 * /root/reference/lib/lz4-mt_decompress.c:349-362 for every record of a batch).
 *
 * Why split: finding where the tokens are is a serial pointer chase per block (each token's
 * position follows from the previous token's lengths).  Inside a wave-per-record kernel that chase
 * runs on one scalar thread (~180 cycles per sequence, measured) while 64 lanes wait.  Here the
 * chase is a kernel of its own in which every *lane* walks a different block, so the SIMD is full:
 *
 *   K1 zmt_dec_frames_kernel  thread per record: record + frame header checks, block-header walk
 *                             -> block table (offset, size, stored flag), expected checksum.
 *   K2 zmt_dec_parse_kernel   lane per block: serial token walk -> u16 token positions (2 B per
 *                             sequence), block decoded sizes.
 *   K3 zmt_dec_copy2_kernel   (lz4_dec_copy2.hip) wave per record: 64 sequences per step.
 *
 * Extra HBM traffic vs a fused kernel: the token list, 2 B per sequence written by K2 and read
 * by K3 (about +20 % of the algorithmic bytes on enwik-like text).
 * Frames whose block size exceeds 64 KiB are flagged for the serial kernel (never produced by
 * lz4-mt; LZ4F allows them).
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define ST_NEEDS_SERIAL 100u /* internal: record is decoded by zmt_lz4_dec_serial afterwards */
#define BLK_STORED 0x80000000u
#define BLK_EMPTY 0xFFFFFFFFu


/* token-list base (in u16 entries) of global block gb whose compressed bytes start at stream
 * offset coff: disjoint per block because a block of c bytes holds at most c/3 + 1 sequences */
static __device__ __forceinline__ u64 tok_base(u64 coff, u32 gb)
{
	return ((coff / 3) & ~63ull) + 128ull * gb;
}

/* est[i] = number of 64 KiB blocks record i decodes to if all but its last block are full */
extern "C" __global__ void __launch_bounds__(256)
zmt_dec_nblk_kernel(const u32 *__restrict__ out_len, u32 nrec, u32 *__restrict__ est)
{
	u32 i = blockIdx.x * 256 + threadIdx.x;
	if (i < nrec)
		est[i] = (out_len[i] + ZMT_BLOCK - 1) / ZMT_BLOCK;
}

/* ------------------------------------------------------------------------------------- K1 */
extern "C" __global__ void __launch_bounds__(256)
zmt_dec_frames_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
		      const u32 *__restrict__ rec_len, u32 nrec, const u32 *__restrict__ out_len,
		      const u64 *__restrict__ blk0, /* exclusive scan of ceil(out_len/64K) */
		      u64 *__restrict__ blk_coff, u32 *__restrict__ blk_csize,
		      u32 *__restrict__ rec_nblk, u32 *__restrict__ rec_flags,
		      u32 *__restrict__ status, u32 *__restrict__ chk_expect,
		      u32 *__restrict__ chk_valid)
{
	const u32 rec = blockIdx.x * 256 + threadIdx.x;
	if (rec >= nrec)
		return;
	const u64 roff = rec_off[rec];
	const u8 *r = stream + roff;
	const u32 rlen = rec_len[rec];
	const u32 cap = out_len[rec];
	const u64 b0 = blk0[rec];
	const u32 nb_max = (u32)(blk0[rec + 1] - b0);
	u32 st = ST_OK, nb = 0, ip = 0, flen = 0, hdr, flg, bd;
	u32 has_csize = 0, has_ccheck = 0, indep = 0;

	chk_valid[rec] = 0;
	chk_expect[rec] = 0;
	rec_nblk[rec] = 0;
	rec_flags[rec] = 0;
	/* every slot of this record in the block table must be defined for the parse kernel, also
	 * when the record is rejected below */
	for (u32 i = 0; i < nb_max; i++)
		blk_csize[b0 + i] = BLK_EMPTY;
	if (rlen < 12 || ld32u(r) != ZMT_SKIP_MAGIC || ld32u(r + 4) != 4 || ld32u(r + 8) != rlen - 12) {
		status[rec] = ST_BAD_RECORD;
		return;
	}
	flen = rlen - 12;
	r += 12;
	if (flen < 7 || ld32u(r) != ZMT_LZ4F_MAGIC) {
		status[rec] = ST_BAD_FRAME;
		return;
	}
	flg = r[4];
	bd = r[5];
	if ((flg >> 6) != 1 || (flg & 0x02) || (bd & 0x8F) || (bd >> 4) < 4) {
		status[rec] = ST_BAD_FRAME;
		return;
	}
	if ((flg & 0x10) || (flg & 0x01)) {
		status[rec] = ST_NEEDS_SERIAL; /* block checksums / dictionary id: the wave-per-record decoder */
		return;
	}
	indep = (flg >> 5) & 1;
	has_csize = (flg >> 3) & 1;
	has_ccheck = (flg >> 2) & 1;
	hdr = 7 + (has_csize ? 8 : 0);
	if (flen < hdr) {
		status[rec] = ST_BAD_FRAME;
		return;
	}
	{
		u8 d[10];
		for (u32 i = 0; i < hdr - 5; i++)
			d[i] = r[4 + i];
		if (r[hdr - 1] != ((xxh32_short(d, hdr - 5) >> 8) & 0xFF)) {
			status[rec] = ST_BAD_FRAME;
			return;
		}
	}
	if ((bd >> 4) != 4) {
		status[rec] = ST_NEEDS_SERIAL; /* 256 KiB+ blocks: rare, handled by the serial kernel */
		return;
	}
	if (has_csize && (ld32u(r + 10) != 0 || ld32u(r + 6) != cap)) {
		status[rec] = ST_SIZE_MISMATCH;
		return;
	}
	ip = hdr;
	for (;;) {
		u32 bh, bsz;
		if (flen - ip < 4) {
			st = ST_BAD_BLOCK;
			break;
		}
		bh = ld32u(r + ip);
		ip += 4;
		if (bh == 0)
			break;
		bsz = bh & 0x7FFFFFFFu;
		if (bsz > ZMT_BLOCK || flen - ip < bsz || bsz == 0) {
			st = ST_BAD_BLOCK;
			break;
		}
		if (nb >= nb_max) {
			/* more blocks than full 64 KiB blocks would need: legal LZ4F, not ours */
			st = ST_NEEDS_SERIAL;
			break;
		}
		blk_coff[b0 + nb] = roff + 12 + ip;
		blk_csize[b0 + nb] = bsz | (bh & BLK_STORED);
		nb++;
		ip += bsz;
	}
	if (st == ST_OK) {
		if (has_ccheck) {
			if (flen - ip < 4) {
				st = ST_BAD_BLOCK;
			} else {
				chk_expect[rec] = ld32u(r + ip);
				chk_valid[rec] = 1;
				ip += 4;
			}
		}
		if (st == ST_OK && ip != flen)
			st = ST_TRAILING;
	}
	if (st != ST_OK)
		for (u32 i = 0; i < nb; i++)
			blk_csize[b0 + i] = BLK_EMPTY; /* rejected record: nothing to parse */
	rec_nblk[rec] = nb;
	rec_flags[rec] = indep;
	status[rec] = st;
}

/* ------------------------------------------------------------------------------------- K2 */
/*
 * One wave = 64 consecutive blocks, lane per block, blk_ntok[gb] = number of sequences (0 for
 * stored / empty / malformed), blk_olen[gb] = bytes the block decodes to (0xFFFFFFFF = malformed).
 *
 * Measured on MI355X (tools/ubench/chase.hip): a wave load whose 64 lanes hit 64 different cache
 * lines costs ~1000-2000 cycles, a scattered 2-byte store per lane several times that, an LDS
 * dependent chain ~160-250 cycles per step.  So the lanes never touch global memory themselves:
 *   - input: each lane parses out of its own 256-byte LDS ring; every 8 steps the wave tops the
 *     rings up in 128-byte pieces, eight lanes fetching one block's piece (8 lines per load
 *     instruction instead of 64);
 *   - output: token positions collect in an LDS tile [lane][P_TILE]; all lanes emit exactly one
 *     token per step, so every P_TILE steps the whole tile leaves in 16-byte pieces, P_TILE / 8 lanes
 *     per block.
 * A parse position outside the ring (after a long literal run) falls back to a direct load.
 *
 * Sizing.  A wave's time is fixed by its longest block (about 5 500 steps on the bench text, 3.9 ms
 * with one wave per SIMD), so the launch takes (rounds of resident waves) x (time of a wave): what
 * matters is that every wave of the launch is resident at once.  The first version (384-byte rings,
 * 64-token tile: 34 KiB of LDS, 4 waves per CU) ran the 2 048 waves of the 8 GiB bench in two rounds,
 * 8.0 ms; at 6 waves per CU (256-byte rings) still two rounds, each slower: 10.3 ms; 256-byte rings
 * and a 16-token tile are 19.5 KiB = 8 waves per CU, one round of 6.0 ms (a wave takes 1.5 x as long
 * with two per SIMD: 184 instructions per step, about 57 % of what a SIMD issues for two waves).
 * Token positions straight to memory (no tile, 9 waves per CU) cost 64 cache lines per store: 7.4 ms.
 */
#ifndef ZMT_EMU
#define KT() (prof ? (u64)clock64() : 0ull)
#else
#define KT() 0ull
#endif
#ifndef P_RING
#define P_RING 256u  /* two 128-byte units per lane */
#endif
#define P_UNIT 128u  /* refill granule: one aligned line of the stream */
typedef u32 v4u __attribute__((vector_size(16)));
#define P_RSTRIDE (P_RING + 16u) /* row stride of the input rings: ring + 16-byte mirror */
#ifndef P_TILE
#define P_TILE 16u   /* token positions a lane collects between two drains of the tile (16, 32 or 64) */
#endif
#define P_TSTRIDE (2u * P_TILE + 8u) /* row stride of the token tile (P_TILE x u16 + pad) */
#define P_TLPR (P_TILE / 8u)         /* lanes that move one row of the tile (16 bytes each) */
#ifndef P_CADENCE
#define P_CADENCE 16u /* scheduled top-ups every P_CADENCE steps (a power of two); lanes that run short in between ask at once (P_URGENT) */
#endif
#ifndef P_URGENT
#define P_URGENT 24u /* top the rings up at once when a lane has fewer bytes than this ahead (0 = never) */
#endif

/* ring offset of g-coordinate g for a lane whose ring lap starts at rb (0 <= g - rb < 2 * P_RING) */
static __device__ __forceinline__ u32 ring_off(u32 g, u32 rb)
{
	const u32 d = g - rb;
	return d < P_RING ? d : d - P_RING;
}

/* four bytes at ring offset o (any alignment) as two ALIGNED dword reads + a funnel shift: a
 * misaligned ds_read_b32 is replayed lane by lane on gfx950 (64 LDS cycles per wave instruction,
 * tools/ubench/lds_ops.hip), an aligned one costs 2.5.  The 16-byte mirror behind the ring keeps
 * the second dword inside the row. */
static __device__ __forceinline__ u32 ring_ld32(const u8 *ring, u32 o)
{
	const u32 *w = (const u32 *)(ring + (o & ~3u));
	return wv_alignbyte(w[1], w[0], o & 3);
}

extern "C" __global__ void __launch_bounds__(64)
zmt_dec_parse_kernel(const u8 *__restrict__ stream, u64 stream_bytes,
		     const u64 *__restrict__ blk_coff, const u32 *__restrict__ blk_csize,
		     const u64 *__restrict__ nblk_ptr, u16 *__restrict__ tok,
		     u32 *__restrict__ bidx, u32 *__restrict__ blk_ntok, u32 *__restrict__ blk_olen,
		     unsigned long long *prof, u32 xflags)
{
	__shared__ __attribute__((aligned(16))) u8 ring_lds[64 * P_RSTRIDE];
	__shared__ __attribute__((aligned(16))) u8 dump_lds[128]; /* where the pieces of rows that asked for nothing go */
#ifndef P_DIRECT_TOK
	__shared__ __attribute__((aligned(16))) u8 tile_lds[64 * P_TSTRIDE];
#endif
	const int lane = wv_lane();
	const u32 gb = blockIdx.x * 64 + (u32)lane;
	const u64 nblk = *nblk_ptr;
	if ((u64)blockIdx.x * 64 >= nblk)
		return; /* whole wave idle */
	const bool exists = (u64)gb < nblk;
	const u32 cs_raw = exists ? blk_csize[gb] : BLK_EMPTY;
	const bool parse = exists && cs_raw != BLK_EMPTY && !(cs_raw & BLK_STORED);
	const u32 cs = parse ? cs_raw : 0;
	const u64 coff = parse ? blk_coff[gb] : 0;
	const u64 tbase = tok_base(coff, gb);
	/* wave-relative 32-bit addressing: 64 consecutive blocks span < 2^32 bytes of stream */
	u64 cmin = parse ? coff : ~0ull, tmin = parse ? tbase : ~0ull;
	for (int d = 32; d; d >>= 1) {
		u32 lo = wv_shfl((u32)cmin, lane ^ d), hi = wv_shfl((u32)(cmin >> 32), lane ^ d);
		u64 o = (u64)hi << 32 | lo;
		cmin = o < cmin ? o : cmin;
		lo = wv_shfl((u32)tmin, lane ^ d);
		hi = wv_shfl((u32)(tmin >> 32), lane ^ d);
		o = (u64)hi << 32 | lo;
		tmin = o < tmin ? o : tmin;
	}
	if (!wv_any(parse)) {
		if (exists) {
			blk_ntok[gb] = 0;
			blk_olen[gb] = (cs_raw == BLK_EMPTY) ? 0 : (cs_raw & 0x7FFFFFFFu);
		}
		return;
	}
	const u32 trel = (u32)(tbase - tmin);
	const u8 *src = stream + coff;
	u8 *const myring = ring_lds + (u32)lane * P_RSTRIDE;
#ifndef P_DIRECT_TOK
	u16 *const mytile = (u16 *)(tile_lds + (u32)lane * P_TSTRIDE);
#endif

	u32 pos = 0, opos = 0, n = 0;
	/* ring bookkeeping in "g" coordinates: g = boff + block position = stream offset - abase,
	 * abase = the wave's first block start rounded down to 128, so g % 128 is the position inside
	 * a memory line.  ring[g & 255] holds byte g for g in [ghi - 256, ghi); [ghi, greq) is in flight */
	const u64 abase = cmin & ~127ull;
	const u32 boff = (u32)(coff - abase);
	u32 ghi = boff & ~(P_UNIT - 1), greq = ghi;
	u32 rb = ghi;  /* g-coordinate of ring offset 0 of the current lap */
	u32 pend_g = 0, pend_off = 0;
	u64 pendm = 0; /* lanes with a unit in flight (wave-uniform) */
	v4u pv[8];
	for (int i = 0; i < 8; i++)
		pv[i] = (v4u){0, 0, 0, 0};
	u32 bx_pending = 0;
	bool ok = true, done = !parse;
	const int dgrp = lane >> 3, dpiece = lane & 7; /* refill: 8 lanes per 128-byte line */
	const int tgrp = lane / (int)P_TLPR, tpiece = lane % (int)P_TLPR; /* tile drain: P_TLPR lanes per row */

	u64 c_refill = 0, c_token = 0, c_drain = 0, c_slow = 0, c_ext = 0, t_begin = KT();
	u64 c_t1 = 0, c_t2 = 0, c_t3 = 0, c_rounds = 0;
	for (u32 step = 0;; step++) {
		u64 tk0 = KT();
		/* ---------------- top up the rings (every P_CADENCE steps) ----------------
		 * Refill unit = one 128-byte line of the stream, aligned in *global* memory, fetched by
		 * eight lanes with one 16-byte load each: a load instruction serves 8 blocks and touches
		 * 8 lines (the texture path costs per line, not per byte).  Software-pipelined: units
		 * requested in one round land in LDS at the start of the next round. */
		/* a lane whose parse position has run past what its ring holds (a long literal run) would
		 * take the global-memory path -- a full memory round trip for the whole wave -- for every token
		 * until the next scheduled round: top up now (slow tokens 1.08 -> 0.04 per step, 6.0 -> 5.1 ms) */
		const bool urgent = P_URGENT && wv_any(!done && boff + pos + P_URGENT > ghi && ghi < boff + cs);
		if ((step & (P_CADENCE - 1)) == 0 || step < 4 || urgent) { /* start-up: back-to-back rounds fill the ring */
			c_rounds++;
			if (pendm) {
				wv_sync();
				/* the eight cross-lane reads first, then the stores: one wait instead of one per piece */
				u32 ros[8];
				ZMT_UNROLL
				for (int i = 0; i < 8; i++)
					ros[i] = wv_shfl(pend_off, 8 * i + dgrp);
				/* no branches around the stores: every exec-masked block ends in a wait of its own and
				 * eight of them in a row cost ~1 400 cycles per round; rows that asked for nothing put
				 * their (meaningless) piece into a 128-byte dump instead */
				ZMT_UNROLL
				for (int i = 0; i < 8; i++) {
					const int r = 8 * i + dgrp;
					const u32 ro = ros[i];
					const bool live = (pendm >> r) & 1;
					u8 *const dump = dump_lds + 16u * (u32)dpiece;
					u8 *const d = ring_lds + (u32)r * P_RSTRIDE + ro + 16u * (u32)dpiece;
					u8 *const d1 = live ? d : dump;
					*(v4u *)d1 = pv[i];
					/* mirror of the ring's first 16 bytes (dword reads never wrap); every other lane
					 * stores its piece a second time where it already is */
					*(v4u *)((live && ro == 0 && dpiece == 0) ? d + P_RING : d1) = pv[i];
				}
				wv_sync();
				if ((pendm >> lane) & 1)
					ghi = pend_g + P_UNIT;
			}
			{ u64 t_ = KT(); c_ext += t_ - tk0; }
			const u32 gp0 = boff + pos;
			if (!done && gp0 >= greq)
				ghi = greq = rb = gp0 & ~(P_UNIT - 1); /* long jump: restart the ring at the parse position */
			while (gp0 - rb >= P_RING)
				rb += P_RING; /* the parse position entered the next lap */
			/* the slot to be overwritten holds [greq - P_RING, greq - P_RING + 128): already parsed? */
			const bool need = !done && greq < boff + cs && greq <= gp0 + (P_RING - P_UNIT);
			pendm = wv_ballot(need);
			pend_g = greq;
			pend_off = ring_off(greq, rb);
			if (pendm) {
				u32 rgs[8];
				ZMT_UNROLL
				for (int i = 0; i < 8; i++)
					rgs[i] = wv_shfl(greq, 8 * i + dgrp);
				/* unconditional loads for the same reason: a row that asked for nothing fetches the wave's
				 * first line again (always inside the stream) and its piece goes to the dump */
				ZMT_UNROLL
				for (int i = 0; i < 8; i++) {
					const int r = 8 * i + dgrp;
					const u32 r_g = ((pendm >> r) & 1) && !(xflags & 2) ? rgs[i] : 0u;
					/* 16-byte aligned; may run up to 127 bytes past stream_bytes: the stream allocation
					 * carries that slack (include/gpumt.h) */
					pv[i] = *(const v4u *)(stream + abase + r_g + 16u * (u32)dpiece);
				}
				if (need)
					greq += P_UNIT;
			}
		}
		{ u64 t_ = KT(); c_refill += t_ - tk0; tk0 = t_; }
		/* ---------------- one token per lane ----------------
		 * Fast path, branch-free: two dependent LDS dword reads (token + first literal-length
		 * byte; offset + first match-length byte).  Anything else -- bytes not in the ring yet,
		 * a 255 continuation byte, the block's last sequence, malformed input -- sends that lane
		 * through the generic path below (rare after start-up, and then only those lanes). */
		u32 my_pos = pos, my_opos = opos;
		bool emit = false;
		bool slow = false;
		if (!done) {
			const u32 gp = boff + pos;
			const bool in1 = gp + 4 <= ghi && gp + P_RING >= ghi;
			const u32 w = ring_ld32(myring, in1 ? ring_off(gp, rb) : 0); /* may run into the mirror */
			const u32 tokb = w & 255;
			const bool lx = (tokb >> 4) == 15;
			const u32 b1 = (w >> 8) & 255;
			const u32 lit = (tokb >> 4) + (lx ? b1 : 0);
			const u32 lend = pos + 1 + (lx ? 1 : 0) + lit;
			const u32 g2 = boff + lend;
			const bool in2 = g2 + 4 <= ghi && g2 + P_RING >= ghi;
			const u32 w2 = ring_ld32(myring, in2 ? ring_off(g2, rb) : 0);
			{ u64 t_ = KT(); c_t1 += t_ - tk0; tk0 = t_; }
			const bool mx = (tokb & 15) == 15;
			const u32 b2 = (w2 >> 16) & 255;
			const u32 ml = (tokb & 15) + (mx ? b2 : 0);
			const u32 m = lend + 2 + (mx ? 1 : 0);
			/* fast path valid: both reads in the ring, no 255 continuation, a match follows
			 * and a further token follows the match (m < cs) */
			slow = !in1 || !in2 || (lx && b1 == 255) || (mx && b2 == 255) || m >= cs ||
			       opos + lit + ml + 4 > ZMT_BLOCK;
			if (!slow) {
				emit = true;
				opos += lit + ml + 4;
				pos = m;
			}
			{ u64 t_ = KT(); c_t2 += t_ - tk0; tk0 = t_; }
		}
		if (slow) {
			/* generic path (same arithmetic as the serial decoder), from global memory */
			c_slow++;
			u32 tokb = src[pos], lit = tokb >> 4, h = pos + 1;
			if (lit == 15) {
				u32 b;
				do {
					if (h >= cs) {
						ok = false;
						break;
					}
					b = src[h++];
					lit += b;
				} while (b == 255);
			}
			const u32 lend = h + lit;
			if (ok && (lend > cs || lend < h))
				ok = false;
			if (ok) {
				emit = true;
				opos += lit;
				if (lend == cs) {
					done = true; /* last sequence: literals only */
				} else if (lend + 2 > cs) {
					ok = false;
				} else {
					u32 ml = tokb & 15, m = lend + 2;
					if (ml == 15) {
						u32 b;
						do {
							if (m >= cs) {
								ok = false;
								break;
							}
							b = src[m++];
							ml += b;
						} while (b == 255);
					}
					opos += ml + 4;
					if (opos > ZMT_BLOCK || m >= cs)
						ok = false; /* a block cannot end with a match */
					pos = m;
				}
			}
			if (!ok) {
				done = true;
				emit = false;
			}
		}
		{ u64 t_ = KT(); c_t3 += t_ - tk0; tk0 = t_; }
		if (emit) {
#ifdef P_DIRECT_TOK
			/* no LDS tile: the position goes straight to the list (2-byte scattered stores: nothing
			 * waits for them, consecutive ones of a lane fall into the same L2 line) */
			tok[tmin + trel + n] = (u16)my_pos;
			if ((n & 63) == 0)
				bidx[((tmin + trel) >> 6) + (n >> 6)] = my_opos;
#else
			mytile[n & (P_TILE - 1)] = (u16)my_pos;
			if ((n & 63) == 0)
				bx_pending = my_opos;
#endif
			n++;
		}
		{ u64 t_ = KT(); c_token += t_ - tk0; tk0 = t_; }
		/* ---------------- drain the token tile (every P_TILE steps, and at the end) -------- */
		const bool all_done = !wv_any(!done);
#ifndef P_DIRECT_TOK
		if ((step & (P_TILE - 1)) == P_TILE - 1 || all_done) {
			wv_sync();
			/* lanes that emitted in this window own tile rows worth writing */
			const u32 first = step & ~(P_TILE - 1);        /* first step of the window */
			const u32 mine = n > first ? n - first : 0;    /* tokens this lane produced in it */
			const u64 havem = wv_ballot(mine > 0);
			ZMT_UNROLL
			for (int i = 0; i < (int)P_TLPR; i++) {
				const int r = (int)(64u / P_TLPR) * i + tgrp;
				const u32 r_t = wv_shfl(trel, r);
				if (((havem >> r) & 1) && !(xflags & 4)) {
					const u8 *t = tile_lds + (u32)r * P_TSTRIDE + 16u * (u32)tpiece;
					const u64 x = *(const u64 *)t, y = *(const u64 *)(t + 8);
					u64 *g = (u64 *)(tok + tmin + r_t + first + 8u * (u32)tpiece);
					g[0] = x;
					g[1] = y;
				}
			}
			if (mine > 0 && (first & 63) == 0)
				bidx[((tmin + trel) >> 6) + (first >> 6)] = bx_pending;
			wv_sync();
		}
#else
		(void)bx_pending;
		(void)tgrp;
		(void)tpiece;
#endif
		{ u64 t_ = KT(); c_drain += t_ - tk0; tk0 = t_; }
		if (all_done) {
#ifndef ZMT_EMU
			if (prof) {
				u64 slow_all = 0;
				for (int d = 0; d < 64; d++)
					slow_all += wv_readlane((u32)c_slow, d);
				if (lane == 0) {
					atomicAdd(prof + 0, (unsigned long long)(KT() - t_begin));
					atomicAdd(prof + 1, (unsigned long long)c_refill);
					atomicAdd(prof + 2, (unsigned long long)c_token);
					atomicAdd(prof + 3, (unsigned long long)c_drain);
					atomicAdd(prof + 4, (unsigned long long)(step + 1));
					atomicAdd(prof + 5, (unsigned long long)slow_all);
					atomicAdd(prof + 6, 1ull);
					atomicAdd(prof + 7, (unsigned long long)c_ext);
					atomicAdd(prof + 8, (unsigned long long)c_rounds);
					atomicAdd(prof + 10, (unsigned long long)c_t1);
					atomicAdd(prof + 11, (unsigned long long)c_t2);
					atomicAdd(prof + 12, (unsigned long long)c_t3);
				}
			}
#endif
			break;
		}
	}
	(void)c_ext;
	if (exists) {
		if (parse) {
			if (opos > ZMT_BLOCK)
				ok = false;
			blk_ntok[gb] = ok ? n : 0;
			blk_olen[gb] = ok ? opos : 0xFFFFFFFFu;
		} else {
			blk_ntok[gb] = 0;
			blk_olen[gb] = (cs_raw == BLK_EMPTY) ? 0 : (cs_raw & 0x7FFFFFFFu);
		}
	}
}
