/*
 * lz4_dec_split.hip -- LZ4 frame decoder as a three-kernel pipeline ("split" variant, default).
 *
 * Same contract as the other decoders (replaces LZ4F_decompress at
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
 *                             sequence) and the output offset of every 64th sequence.
 *   K3 zmt_dec_copy_kernel    wave per record: 64 sequences per step.  Lane k loads token k's
 *                             fields, a DPP prefix sum yields output positions, literals and
 *                             matches are copied with unaligned 8-byte LDS accesses into a sliding
 *                             8 KiB LDS output window (older sources: global memory), dependent
 *                             matches resolve in watermark rounds, the window drains to HBM as
 *                             coalesced 16-byte stores.  Lengths > 64 and stored blocks take a
 *                             wave-cooperative path.
 *
 * Extra HBM traffic vs the fused kernels: the token list, 2 B per sequence written by K2 and read
 * by K3 (about +20 % of the algorithmic bytes on enwik-like text).
 * Frames whose block size exceeds 64 KiB are flagged for the serial kernel (never produced by
 * lz4-mt; LZ4F allows them).
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#ifndef WIN
#define WIN 8192u      /* LDS output window per wave */
#endif
#ifndef WIN_KEEP
#define WIN_KEEP 4096u /* history kept when the window slides */
#endif
#define CAP_LEN 64u
#define SPAN_MAX 2048u
#define CSTAGE 1024u /* compressed bytes staged in LDS per batch */
#define CSLACK 32u
#ifndef COPY_WAVES_PER_SIMD
#define COPY_WAVES_PER_SIMD 4 /* caps VGPRs at 128: four 256-thread workgroups per CU (LDS: 4 x 36 KiB) */
#endif
#define ST_NEEDS_SERIAL 100u /* internal: record is decoded by zmt_lz4_dec_serial afterwards */
#define BLK_STORED 0x80000000u
#define BLK_EMPTY 0xFFFFFFFFu

static __device__ __forceinline__ void st64u(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }

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
		status[rec] = ST_UNSUPPORTED;
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
 *   - output: token positions collect in an LDS tile [lane][64]; all lanes emit exactly one token
 *     per step, so every 64 steps the whole tile leaves as 128-byte lines, again 8 per store.
 * A parse position outside the ring (after a long literal run) falls back to a direct load.
 */
#ifndef ZMT_EMU
#define KT() (prof ? (u64)clock64() : 0ull)
#else
#define KT() 0ull
#endif
#ifndef P_RING
#define P_RING 384u  /* three 128-byte units per lane */
#endif
#define P_UNIT 128u  /* refill granule: one aligned line of the stream */
typedef u32 v4u __attribute__((vector_size(16)));
#define P_RSTRIDE (P_RING + 16u) /* row stride of the input rings: ring + 16-byte mirror */
#define P_TSTRIDE 136u /* row stride of the token tile (64 x u16 + pad) */

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
	const int dgrp = lane >> 3, dpiece = lane & 7; /* refill and tile drain: 8 lanes per 128-byte line */

	u64 c_refill = 0, c_token = 0, c_drain = 0, c_slow = 0, c_ext = 0, t_begin = KT();
	u64 c_t1 = 0, c_t2 = 0, c_t3 = 0;
	for (u32 step = 0;; step++) {
		u64 tk0 = KT();
		/* ---------------- top up the rings (every 8 steps) ----------------
		 * Refill unit = one 128-byte line of the stream, aligned in *global* memory, fetched by
		 * eight lanes with one 16-byte load each: a load instruction serves 8 blocks and touches
		 * 8 lines (the texture path costs per line, not per byte).  Software-pipelined: units
		 * requested in one round land in LDS at the start of the next round. */
		if ((step & 7) == 0 || step < 4) { /* start-up: three back-to-back rounds fill the ring */
			if (pendm) {
				wv_sync();
				ZMT_UNROLL
				for (int i = 0; i < 8; i++) {
					const int r = 8 * i + dgrp;
					const u32 ro = wv_shfl(pend_off, r);
					if ((pendm >> r) & 1) {
						u8 *d = ring_lds + (u32)r * P_RSTRIDE + ro + 16u * (u32)dpiece;
						*(v4u *)d = pv[i];
						if (ro == 0 && dpiece == 0)
							*(v4u *)(d + P_RING) = pv[i]; /* mirror: dword reads never wrap */
					}
				}
				wv_sync();
				if ((pendm >> lane) & 1)
					ghi = pend_g + P_UNIT;
			}
			const u32 gp0 = boff + pos;
			if (!done && gp0 >= greq)
				ghi = greq = rb = gp0 & ~(P_UNIT - 1); /* long jump: restart the ring at the parse position */
			while (gp0 - rb >= P_RING)
				rb += P_RING; /* the parse position entered the next lap */
			/* the slot to be overwritten holds [greq-384, greq-256): already parsed? */
			const bool need = !done && greq < boff + cs && greq <= gp0 + (P_RING - P_UNIT);
			pendm = wv_ballot(need);
			pend_g = greq;
			pend_off = ring_off(greq, rb);
			if (pendm) {
				ZMT_UNROLL
				for (int i = 0; i < 8; i++) {
					const int r = 8 * i + dgrp;
					const u32 r_g = wv_shfl(greq, r);
					v4u v = {0, 0, 0, 0};
					if (((pendm >> r) & 1) && !(xflags & 2)) {
						/* 16-byte aligned; may run up to 127 bytes past stream_bytes: the stream
						 * allocation carries that slack (include/gpumt.h) */
						v = *(const v4u *)(stream + abase + r_g + 16u * (u32)dpiece);
					}
					pv[i] = v;
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
			mytile[n & 63] = (u16)my_pos;
			if ((n & 63) == 0)
				bx_pending = my_opos;
#endif
			n++;
		}
		{ u64 t_ = KT(); c_token += t_ - tk0; tk0 = t_; }
		/* ---------------- drain the token tile (every 64 steps, and at the end) -------- */
		const bool all_done = !wv_any(!done);
#ifndef P_DIRECT_TOK
		if ((step & 63) == 63 || all_done) {
			wv_sync();
			/* lanes that emitted in this window own tile rows worth writing */
			const u32 first = step & ~63u;                 /* first step of the window */
			const u32 mine = n > first ? n - first : 0;    /* tokens this lane produced in it */
			const u64 havem = wv_ballot(mine > 0);
			ZMT_UNROLL
			for (int i = 0; i < 8; i++) {
				const int r = 8 * i + dgrp;
				const u32 r_t = wv_shfl(trel, r);
				if (((havem >> r) & 1) && !(xflags & 4)) {
					const u8 *t = tile_lds + (u32)r * P_TSTRIDE + 16u * (u32)dpiece;
					const u64 x = *(const u64 *)t, y = *(const u64 *)(t + 8);
					u64 *g = (u64 *)(tok + tmin + r_t + first + 8u * (u32)dpiece);
					g[0] = x;
					g[1] = y;
				}
			}
			if (mine > 0)
				bidx[((tmin + trel) >> 6) + (first >> 6)] = bx_pending;
			wv_sync();
		}
#else
		(void)bx_pending;
		(void)dpiece;
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

/* ------------------------------------------------------------------------------------- K3 */
static __device__ __forceinline__ void copy_units(u8 *d, const u8 *s, u32 len, bool slop)
{
	if (len >= 8) {
		for (u32 i = 0; i + 8 < len; i += 8)
			st64u(d + i, ld64u(s + i));
		st64u(d + len - 8, ld64u(s + len - 8));
	} else if (len >= 4) {
		u32 a = ld32u(s), b = ld32u(s + len - 4);
		st32u(d, a);
		st32u(d + len - 4, b);
	} else if (len) {
		if (slop) {
			st32u(d, ld32u(s));
		} else {
			for (u32 i = 0; i < len; i++)
				d[i] = s[i];
		}
	}
}

static __device__ __forceinline__ void win_match(u8 *d, u32 off, u32 ml)
{
	const u8 *s = d - off;
	if (ml >= 4 && (off >= 8 || off >= ml)) {
		copy_units(d, s, ml, false);
	} else {
		u32 j = 0;
		for (u32 i = 0; i < ml; i++) {
			d[i] = s[j];
			if (++j == off)
				j = 0;
		}
	}
}

struct CopyState {
	u32 opos, flushed, valid_from, wbase, fenced;
};

static __device__ __forceinline__ void flush_to(CopyState &st, const u8 *win, u8 *out, u32 upto, int lane)
{
	u32 f = st.flushed;
	if (upto <= f)
		return;
	const u8 *w = win - st.wbase;
	u32 head = (16 - (f & 15)) & 15;
	if (head > upto - f)
		head = upto - f;
	if ((u32)lane < head)
		out[f + lane] = w[f + lane];
	f += head;
	u32 body_end = f + ((upto - f) & ~15u);
	for (u32 pos = f + 16 * (u32)lane; pos < body_end; pos += 1024) {
		const u8 *r = w + pos;
		u64 a = *(const u64 *)r, b = *(const u64 *)(r + 8);
		st64u(out + pos, a);
		st64u(out + pos + 8, b);
	}
	if ((u32)lane < upto - body_end)
		out[body_end + lane] = w[body_end + lane];
	st.flushed = upto;
}

static __device__ __forceinline__ void win_slide(CopyState &st, u8 *win, int lane)
{
	u32 nb = st.opos > WIN_KEEP ? (st.opos - WIN_KEEP) & ~15u : 0;
	if (nb < st.wbase)
		nb = st.wbase;
	if (st.valid_from >= st.opos || nb - st.wbase >= WIN) {
		st.wbase = nb;
		return;
	}
	u32 delta = nb - st.wbase, keep = st.opos - nb;
	wv_sync();
	for (u32 o = 0; o < keep; o += 1024) {
		u32 i = o + 16 * (u32)lane;
		u64 a = 0, b = 0;
		if (i < keep) {
			a = *(const u64 *)(win + delta + i);
			b = *(const u64 *)(win + delta + i + 8);
		}
		wv_sync();
		if (i < keep) {
			*(u64 *)(win + i) = a;
			*(u64 *)(win + i + 8) = b;
		}
	}
	wv_sync();
	st.wbase = nb;
}
static __device__ __forceinline__ void win_reserve(CopyState &st, u8 *win, u32 end, int lane)
{
	if (end - st.wbase > WIN)
		win_slide(st, win, lane);
}

/* cooperative copy of one long sequence straight to global memory (fields are wave-uniform) */
static __device__ void long_sequence(const u8 *lsrc, u32 lit, u32 off, u32 ml, u8 *out, CopyState &st,
				     const u8 *win, int lane)
{
	flush_to(st, win, out, st.opos, lane);
	u32 opos = st.opos;
	wave_copy(out + opos, lsrc, lit, lane);
	opos += lit;
	if (ml) {
		wave_mem_fence();
		const u8 *m = out + opos - off;
		u8 *d = out + opos;
		if (off >= ml) {
			u32 i = 0;
			if (ml >= 512) {
				u32 n4 = ml & ~255u;
				for (i = (u32)lane * 4; i < n4; i += 256)
					st32u(d + i, ld32u(m + i));
				i = n4;
			}
			for (i += (u32)lane; i < ml; i += 64)
				d[i] = m[i];
		} else {
			for (u32 i = (u32)lane; i < ml; i += 64)
				d[i] = m[i % off];
		}
		opos += ml;
	}
	wave_mem_fence();
	st.opos = opos;
	st.flushed = opos;
	st.fenced = opos;
	st.valid_from = opos;
}

template <bool PROF>
static __device__ __forceinline__ void
copy_body(const u8 *__restrict__ stream, u64 stream_bytes, u32 nrec, u8 *out_base,
		    const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
		    const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
		    const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
		    const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
		    const u32 *__restrict__ bidx, const u32 *__restrict__ blk_ntok,
		    const u32 *__restrict__ blk_olen, u32 *__restrict__ status,
		    unsigned long long *prof, u8 *lds)
{
	const int lane = wv_lane();
	const u32 wave = threadIdx.x >> 6;
	const u32 rec = blockIdx.x * 4 + wave;
	if (rec >= nrec)
		return;
	if (wv_readfirst(status[rec]) != ST_OK)
		return;
	u8 *win = lds + wave * (WIN + CSTAGE + CSLACK);
	u8 *cb = win + WIN; /* staged compressed bytes of the current batch */
	u8 *out = out_base + out_off[rec];
	const u32 cap = out_len[rec];
	const u64 b0 = blk0[rec];
	const u32 nb = wv_readfirst(rec_nblk[rec]);
	const bool indep = wv_readfirst(rec_flags[rec]) & 1;
	u32 stc = ST_OK;
	CopyState st;
	st.opos = st.flushed = st.valid_from = st.wbase = st.fenced = 0;
	u64 pc[PROF ? 8 : 1] = {0}, tq = PROF ? KT() : 0, t_begin3 = tq;
#define PC(i) do { if (PROF) { u64 t_ = KT(); pc[PROF ? (i) : 0] += t_ - tq; tq = t_; } } while (0)

	for (u32 bi = 0; bi < nb && stc == ST_OK; bi++) {
		const u32 gb = (u32)(b0 + bi);
		const u32 cs = wv_readfirst(blk_csize[gb]);
		const u64 coff = blk_coff[gb];
		const u8 *src = stream + coff;
		const u32 olen = wv_readfirst(blk_olen[gb]);
		const u32 bstart = st.opos;
		if (olen == 0xFFFFFFFFu || cap - bstart < olen) {
			stc = ST_BAD_BLOCK;
			break;
		}
		if (cs & BLK_STORED) {
			u32 bsz = cs & 0x7FFFFFFFu;
			flush_to(st, win, out, st.opos, lane);
			wave_copy(out + st.opos, src, bsz, lane);
			st.opos += bsz;
			st.flushed = st.opos;
			st.valid_from = st.opos;
			continue;
		}
		const u32 ntok = wv_readfirst(blk_ntok[gb]);
		const u64 tbase = tok_base(coff, gb);
		const u32 low = indep ? bstart : 0;
		/*
		 * Software pipeline over the block's batches: token positions (and the batch's output
		 * offset) are loaded two batches ahead, the batch's compressed bytes -- one coalesced
		 * 1 KiB window starting at its first token -- one batch ahead, so the global-memory
		 * latency of both is covered by the copy work of the batch before.
		 */
		const u16 *tk = tok + tbase;
		const u32 *bx = bidx + (tbase >> 6);
#define TOK_LOAD(T0) (((T0) + (u32)lane < ntok) ? (u32)tk[(T0) + lane] : 0u)
#define STAGE_LOAD(CS0, A, B)                                                                     \
	do {                                                                                       \
		/* may run up to 31 bytes past the block, hence past stream_bytes: the stream      \
		 * allocation carries 256 bytes of slack (include/gpumt.h) */                       \
		const u32 o_ = (CS0) + 16u * (u32)lane;                                            \
		(A) = 0;                                                                           \
		(B) = 0;                                                                           \
		if (o_ < cs + 8) {                                                                 \
			(A) = ld64u(src + o_);                                                     \
			(B) = ld64u(src + o_ + 8);                                                 \
		}                                                                                  \
	} while (0)
		u32 q_cur = TOK_LOAD(0), q_nxt = TOK_LOAD(64);
		u32 b_cur = ntok ? bx[0] : 0, b_nxt = ntok > 64 ? bx[1] : 0;
		u64 ca, cbv;
		STAGE_LOAD(0, ca, cbv);
		u32 cs_cur = 0; /* block position of cb[0] */
		for (u32 t0 = 0; t0 < ntok && stc == ST_OK; t0 += 64) {
			const u32 k = ntok - t0 < 64 ? ntok - t0 : 64;
			PC(7);
			/* land this batch's staged bytes, then put the next loads in flight */
			wv_sync();
			*(u64 *)(cb + 16u * (u32)lane) = ca;
			*(u64 *)(cb + 16u * (u32)lane + 8) = cbv;
			wv_sync();
			const u32 q = q_cur;
			const u32 bofs = wv_readfirst(b_cur);
			const u32 cs0 = cs_cur;
			{
				const u32 q_nn = TOK_LOAD(t0 + 128);
				const u32 b_nn = (t0 + 128 < ntok) ? bx[(t0 >> 6) + 2] : 0;
				const u32 cs_n = wv_readlane(q_nxt, 0); /* first token of the next batch */
				if (t0 + 64 < ntok)
					STAGE_LOAD(cs_n, ca, cbv);
				cs_cur = cs_n;
				q_cur = q_nxt;
				q_nxt = q_nn;
				b_cur = b_nxt;
				b_nxt = b_nn;
			}
			const bool act0 = (u32)lane < k;
			const bool is_last = act0 && t0 + (u32)lane == ntok - 1;
			/* ---- fields of sequence t0+lane: from the staged window when the whole
			 * (ordinary) sequence lies inside it, else straight from global memory ---- */
			u32 lit = 0, ml = 0, off = 1, lsrc = 0;
			const u32 qr = q - cs0;
			bool staged = qr + 80 <= CSTAGE;
			{
				/* fast path, branch-free: two LDS dword reads; a 255 continuation byte or a
				 * sequence outside the staged window takes the generic path (rare) */
				const u32 w = ring_ld32(cb, staged ? qr : 0); /* aligned pair + funnel shift (see ring_ld32) */
				const u32 tokb = w & 255;
				const bool lx = (tokb >> 4) == 15;
				const u32 b1 = (w >> 8) & 255;
				const u32 l_ = (tokb >> 4) + (lx ? b1 : 0);
				const u32 h = q + 1 + (lx ? 1 : 0);
				const u32 lend = h + l_;
				const bool st2 = staged && lend - cs0 + 4 <= CSTAGE;
				const u32 w2 = ring_ld32(cb, st2 ? lend - cs0 : 0);
				const bool mx = (tokb & 15) == 15;
				const u32 b2 = (w2 >> 16) & 255;
				const bool fast = staged && !(lx && b1 == 255) && (is_last || (st2 && !(mx && b2 == 255)));
				if (act0 && fast) {
					lit = l_;
					lsrc = h;
					if (!is_last) {
						off = w2 & 0xFFFF;
						ml = (tokb & 15) + (mx ? b2 : 0) + 4;
					}
				}
				if (act0 && !fast) {
					/* generic: straight from global memory (K2 validated the chain) */
					const u32 tk = src[q];
					u32 l2 = tk >> 4, h2 = q + 1;
					if (l2 == 15) {
						u32 b;
						do {
							b = src[h2++];
							l2 += b;
						} while (b == 255);
					}
					lit = l2;
					lsrc = h2;
					if (!is_last) {
						u32 m = h2 + l2;
						off = ld16u(src + m);
						m += 2;
						ml = tk & 15;
						if (ml == 15) {
							u32 b;
							do {
								b = src[m++];
								ml += b;
							} while (b == 255);
						}
						ml += 4;
					}
					staged = false; /* literals of this sequence come from global memory too */
				}
			}
			PC(0);
			const u32 len = lit + ml;
			const u32 incl = wv_scan_incl(len);
			const u32 bpos = bstart + bofs;
			const u32 op = bpos + incl - len;
			const u32 mpos = op + lit;
			const u32 src_pos = mpos - off;
			const u32 eff = ml < off ? ml : off;
			/* K2 summed the same lengths: positions are consistent by construction */
			if (wv_any(act0 && !is_last && (off == 0 || off > mpos - low))) {
				stc = ST_BAD_BLOCK;
				break;
			}
			/* sub-batches: runs of ordinary sequences; long ones, the block's final
			 * literal-only sequence and (rarely) sources straddling the window / global
			 * frontier are handled one at a time by long_sequence() */
			u64 cutm = wv_ballot(act0 && (lit > CAP_LEN || ml > CAP_LEN || is_last || !staged));
			u32 lo = 0;
			while (lo < k) {
				const u64 rest = cutm & ~((1ull << lo) - 1);
				u32 hi = rest ? (u32)wv_ffs(rest) - 1 : k;
				if (hi > lo) {
					/* ---------- ordinary sequences [lo, hi) ---------- */
					const u32 sub_start = wv_readlane(op, (int)lo);
					{
						/* at most SPAN_MAX bytes per step */
						u64 over = wv_ballot((u32)lane >= lo && (u32)lane < hi &&
								     op + len - sub_start > SPAN_MAX);
						if (over)
							hi = (u32)wv_ffs(over) - 1; /* > lo: one sequence is <= 128 B */
					}
					st.opos = sub_start; /* == end of whatever came before */
					win_reserve(st, win, wv_readlane(op + len, (int)(hi - 1)), lane);
					const u32 near_lo = st.valid_from > st.wbase ? st.valid_from : st.wbase;
					{
						u64 oddm = wv_ballot((u32)lane >= lo && (u32)lane < hi &&
								     src_pos < near_lo && src_pos + eff > st.flushed);
						if (oddm) {
							u32 l1 = (u32)wv_ffs(oddm) - 1;
							cutm |= 1ull << l1;
							if (l1 == lo)
								continue; /* handled as a single sequence below */
							hi = l1;
						}
					}
					const bool act = (u32)lane >= lo && (u32)lane < hi;
					const u32 sub_end = wv_readlane(op + len, (int)(hi - 1));
					u8 *const w0 = win - st.wbase;
					const bool is_far = act && src_pos < near_lo;
					PC(1);
					/* far sources: loads first, one round trip */
					u64 fv[8];
					const bool far_plain = is_far && off >= ml;
					u32 far_trips = 0;
					{
						u32 need = is_far ? src_pos + eff : 0;
						if (wv_any(need > st.fenced)) {
							wave_mem_fence();
							st.fenced = st.flushed;
						}
						if (wv_any(far_plain)) {
							/* widest far match in 8-byte units (1..8), by bisection on ballots */
							const u32 tr = far_plain ? (ml + 7) >> 3 : 0;
							u32 mxt = wv_any(tr > 4) ? 4 : 0;
							mxt += wv_any(tr > mxt + 2) ? 2 : 0;
							mxt += wv_any(tr > mxt + 1) ? 1 : 0;
							mxt += wv_any(tr > mxt) ? 1 : 0;
							far_trips = mxt;
							const u8 *g = out + src_pos;
							ZMT_UNROLL
							for (u32 t = 0; t < 8; t++) {
								fv[t] = 0;
								if (t < far_trips && far_plain && ml > 8 * t) {
									u32 o = 8 * t + 8 <= ml ? 8 * t : (ml >= 8 ? ml - 8 : 0);
									fv[t] = ld64u(g + o);
								}
							}
						}
					}
					PC(2);
					if (act)
						copy_units(w0 + op, cb + (lsrc - cs0), lit, true);
					PC(3);
					if (far_trips) {
						ZMT_UNROLL
						for (u32 t = 0; t < 8; t++) {
							if (t < far_trips && far_plain && ml > 8 * t) {
								u8 *d = w0 + mpos;
								if (ml >= 8) {
									u32 o = 8 * t + 8 <= ml ? 8 * t : ml - 8;
									st64u(d + o, fv[t]);
								} else {
									st32u(d, (u32)fv[0]);
									st32u(d + ml - 4, (u32)(fv[0] >> (8 * (ml - 4))));
								}
							}
						}
					}
					if (is_far && !far_plain) {
						copy_units(w0 + mpos, out + src_pos, off, false);
						win_match(w0 + mpos + off, off, ml - off);
					}
					wv_sync();
					PC(4);
					/* near matches, watermark rounds */
					{
						bool fin = !(act && !is_far);
						for (;;) {
							u64 unf = wv_ballot(!fin);
							if (!unf)
								break;
							u32 first = (u32)wv_ffs(unf) - 1;
							u32 W = wv_readlane(mpos, (int)first);
							bool ready = !fin && src_pos + eff <= W;
							if (ready) {
								win_match(w0 + mpos, off, ml);
								fin = true;
							}
							wv_sync();
						}
					}
					PC(5);
					st.opos = sub_end;
					{
						u32 end = sub_end & ~15u;
						if (end > st.flushed)
							flush_to(st, win, out, end, lane);
					}
					lo = hi;
				}
				if (lo < k && ((cutm >> lo) & 1)) {
					/* ---------- one long (or final literal-only) sequence ---------- */
					const u32 l_lit = wv_readlane(lit, (int)lo), l_ml = wv_readlane(ml, (int)lo);
					const u32 l_off = wv_readlane(off, (int)lo), l_src = wv_readlane(lsrc, (int)lo);
					st.opos = wv_readlane(op, (int)lo);
					long_sequence(src + l_src, l_lit, l_off, l_ml, out, st, win, lane);
					lo++;
				}
			}
		}
		if (stc == ST_OK && st.opos != bstart + olen)
			stc = ST_BAD_BLOCK;
	}
	PC(6);
	flush_to(st, win, out, st.opos, lane);
#ifndef ZMT_EMU
	if (PROF && prof && lane == 0) {
		for (int i = 0; i < (PROF ? 8 : 1); i++)
			atomicAdd(prof + i, (unsigned long long)pc[i]);
		atomicAdd(prof + 8, (unsigned long long)(KT() - t_begin3));
		atomicAdd(prof + 9, 1ull);
	}
#endif
	if (stc == ST_OK && st.opos != cap)
		stc = ST_SIZE_MISMATCH;
	if (lane == 0 && stc != ST_OK)
		status[rec] = stc;
}

#ifdef ZMT_EMU
#define COPY_ATTR
#else
#define COPY_ATTR __attribute__((amdgpu_waves_per_eu(COPY_WAVES_PER_SIMD, COPY_WAVES_PER_SIMD)))
#endif

extern "C" __global__ void __launch_bounds__(256) COPY_ATTR
zmt_dec_copy_kernel(const u8 *__restrict__ stream, u64 stream_bytes, u32 nrec, u8 *out_base,
		    const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
		    const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
		    const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
		    const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
		    const u32 *__restrict__ bidx, const u32 *__restrict__ blk_ntok,
		    const u32 *__restrict__ blk_olen, u32 *__restrict__ status)
{
	__shared__ __attribute__((aligned(16))) u8 lds[4 * (WIN + CSTAGE + CSLACK)];
	copy_body<false>(stream, stream_bytes, nrec, out_base, out_off, out_len, blk0, blk_coff,
			 blk_csize, rec_nblk, rec_flags, tok, bidx, blk_ntok, blk_olen, status, nullptr, lds);
}

#ifndef ZMT_EMU
/* same kernel with per-phase cycle counters (developer tool) */
extern "C" __global__ void __launch_bounds__(256)
zmt_dec_copy_kernel_prof(const u8 *__restrict__ stream, u64 stream_bytes, u32 nrec, u8 *out_base,
			 const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
			 const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
			 const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
			 const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
			 const u32 *__restrict__ bidx, const u32 *__restrict__ blk_ntok,
			 const u32 *__restrict__ blk_olen, u32 *__restrict__ status,
			 unsigned long long *prof)
{
	__shared__ __attribute__((aligned(16))) u8 lds[4 * (WIN + CSTAGE + CSLACK)];
	copy_body<true>(stream, stream_bytes, nrec, out_base, out_off, out_len, blk0, blk_coff,
			blk_csize, rec_nblk, rec_flags, tok, bidx, blk_ntok, blk_olen, status, prof, lds);
}
#endif
