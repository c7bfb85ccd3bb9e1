/*
 * lz4_dec_split.hip -- LZ4 frame decoder, first kernel of the three-kernel pipeline.
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
 *   K2 zmt_dec_parse4_kernel  (lz4_dec_parse4.hip) lane per block: serial token walk -> u16 token
 *                             positions (2 B per sequence), the batch list, block decoded sizes.
 *   K3 zmt_dec_copy3_*_kernel (lz4_dec_copy3.hip) wave per record: up to 64 sequences per step.
 *
 * Extra HBM traffic vs a fused kernel: the token list, 2 B per sequence written by K2 and read
 * by K3 (about +20 % of the algorithmic bytes on enwik-like text).
 * Frames whose block size exceeds 64 KiB are flagged for the serial kernel (never produced by
 * lz4-mt; LZ4F allows them).  (The round-2 parse / copy2 kernels are gone; DESIGN.md keeps their numbers.)
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define ST_NEEDS_SERIAL 100u /* internal: record is decoded by zmt_lz4_dec_serial afterwards */
#define BLK_STORED 0x80000000u
#define BLK_EMPTY 0xFFFFFFFFu


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
