/*
 * lz4_dec.hip -- LZ4 frame decoder, one wave per record (= one chunk).
 *
 * Replaces, per record, LZ4F_decompress (call site /root/reference/lib/lz4-mt_decompress.c:349-362)
 * together with the record checks of pt_read (:229-236): skippable magic and length, LZ4F magic /
 * version / reserved bits / header checksum, block walk (stored or LZ4 block with the chunk's
 * earlier output as prefix -- linked blocks), end mark, content size.  The XXH32 content checksum
 * is extracted here and verified by zmt_xxh32_kernel over the decoded bytes.
 *
 * Variant "serial" (this file, kernel zmt_lz4_dec_serial): the token stream is walked
 * wave-uniformly, the 64 lanes cooperate on each literal run and each match (byte i of the copy
 * is lane i mod 64).  Overlapping matches (offset < length) read the pattern's first period, so a
 * copy never depends on bytes written by the same sequence.  Output is written straight to HBM;
 * match sources are read back through L1/L2 (same wave, program order).
 */
#include "lz4_common.h"
#include "lz4_frame.h"

/*
 * Decode one LZ4 block [src, src+slen) appending at out[opos...]; matches may reach back to
 * out[low].  Returns new opos or 0xFFFFFFFF on malformed input.  limit = highest legal opos.
 */
static __device__ u32 decode_block_serial(const u8 *src, u32 slen, u8 *out, u32 opos, u32 low,
					  u32 limit, int lane)
{
	u32 ip = 0;
	if (slen == 0)
		return 0xFFFFFFFFu;
	for (;;) {
		u32 tok, lit, ml, off;
		if (ip >= slen)
			return 0xFFFFFFFFu;
		tok = uld8(src + ip++);
		lit = tok >> 4;
		if (lit == 15) {
			u32 b;
			do {
				if (ip >= slen)
					return 0xFFFFFFFFu;
				b = uld8(src + ip++);
				lit += b;
			} while (b == 255);
		}
		if (slen - ip < lit || limit - opos < lit)
			return 0xFFFFFFFFu;
		wave_copy(out + opos, src + ip, lit, lane);
		ip += lit;
		opos += lit;
		if (ip == slen)
			return opos;
		if (slen - ip < 2)
			return 0xFFFFFFFFu;
		off = uld16(src + ip);
		ip += 2;
		ml = tok & 15;
		if (ml == 15) {
			u32 b;
			do {
				if (ip >= slen)
					return 0xFFFFFFFFu;
				b = uld8(src + ip++);
				ml += b;
			} while (b == 255);
		}
		ml += 4;
		if (off == 0 || off > opos - low || limit - opos < ml)
			return 0xFFFFFFFFu;
		wave_mem_fence();
		{
			const u8 *m = out + opos - off;
			u8 *d = out + opos;
			if (off >= ml) {
				for (u32 i = (u32)lane; i < ml; i += 64)
					d[i] = m[i];
			} else {
				/* periodic fill: byte i of the match equals byte i mod off of the source */
				for (u32 i = (u32)lane; i < ml; i += 64)
					d[i] = m[i % off];
			}
		}
		opos += ml;
	}
}

/* XXH32 (seed 0) of [p, p + len) by one wave: lanes 0..3 carry the four accumulators over the
 * 16-byte stripes, lane 0 folds the tail; the result is wave-uniform */
static __device__ u32 wave_xxh32(const u8 *p, u32 len, int lane)
{
	u32 v = 0;
	if (len >= 16 && lane < 4) {
		u32 acc = (lane == 0) ? XP1 + XP2 : (lane == 1) ? XP2 : (lane == 2) ? 0u : 0u - XP1;
		const u8 *q = p + lane * 4;
		for (u32 s = 0, ns = len >> 4; s < ns; s++) {
			acc = xxh_round(acc, ld32u(q));
			q += 16;
		}
		v = rotl32(acc, (lane == 0) ? 1 : (lane == 1) ? 7 : (lane == 2) ? 12 : 18);
	}
	v += wv_shfl(v, lane ^ 1);
	v += wv_shfl(v, lane ^ 2);
	u32 h = (len >= 16 ? v : XP5) + len;
	if (lane == 0)
		h = xxh_tail(h, p + (len & ~15u), len & 15);
	return wv_readlane(h, 0);
}

extern "C" __global__ void __launch_bounds__(64)
zmt_lz4_dec_serial(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
		   const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base,
		   const u64 *__restrict__ out_off, u32 *__restrict__ out_len,
		   u32 *__restrict__ status, u32 *__restrict__ chk_expect,
		   u32 *__restrict__ chk_valid, u32 only_status)
{
	const u32 rec = blockIdx.x;
	const int lane = wv_lane();
	if (rec >= nrec)
		return;
	/* clean-up pass of the split decoder: only records it flagged for this kernel */
	if (only_status != 0xFFFFFFFFu && wv_readfirst(status[rec]) != only_status)
		return;
	const u8 *r = stream + rec_off[rec];
	const u32 rlen = rec_len[rec];
	u8 *out = out_base + out_off[rec];
	const u32 cap = out_len[rec];
	u32 st = ST_OK, opos = 0, ip, flen;
	FrameInfo fi;

	if (lane == 0) {
		chk_valid[rec] = 0;
		chk_expect[rec] = 0;
	}
	if (rlen < 12 || uld32(r) != ZMT_SKIP_MAGIC || uld32(r + 4) != 4 ||
	    uld32(r + 8) != rlen - 12) {
		st = ST_BAD_RECORD;
		goto done;
	}
	flen = rlen - 12;
	r += 12;
	st = parse_frame_header(r, flen, fi);
	if (st != ST_OK)
		goto done;
	ip = fi.hdr;
	for (;;) {
		u32 bh, bsz;
		if (flen - ip < 4) {
			st = ST_BAD_BLOCK;
			goto done;
		}
		bh = uld32(r + ip);
		ip += 4;
		if (bh == 0)
			break;
		bsz = bh & 0x7FFFFFFFu;
		if (bsz > fi.blkmax || flen - ip < bsz || (fi.has_bcheck && flen - ip - bsz < 4)) {
			st = ST_BAD_BLOCK;
			goto done;
		}
		if (fi.has_bcheck && wave_xxh32(r + ip, bsz, lane) != uld32(r + ip + bsz)) {
			st = ST_BAD_CHECKSUM; /* LZ4F_ERROR_blockChecksum_invalid */
			goto done;
		}
		if (bh & 0x80000000u) {
			if (cap - opos < bsz) {
				st = ST_BAD_BLOCK;
				goto done;
			}
			wave_copy(out + opos, r + ip, bsz, lane);
			opos += bsz;
		} else {
			u32 room = cap - opos < fi.blkmax ? cap - opos : fi.blkmax;
			u32 np = decode_block_serial(r + ip, bsz, out, opos, fi.indep ? opos : 0,
						     opos + room, lane);
			if (np == 0xFFFFFFFFu) {
				st = ST_BAD_BLOCK;
				goto done;
			}
			opos = np;
		}
		ip += bsz + (fi.has_bcheck ? 4u : 0u);
	}
	/* a frame that states its content size must produce exactly that (= out_len); one that does not
	 * (plain .lz4 files of the lz4 tool; never lz4-mt) was given a capacity and reports its size */
	if (fi.has_csize ? (fi.csize != (u64)opos || opos != cap) : opos > cap) {
		st = ST_SIZE_MISMATCH;
		goto done;
	}
	if (!fi.has_csize && lane == 0)
		out_len[rec] = opos;
	if (fi.has_ccheck) {
		if (flen - ip < 4) {
			st = ST_BAD_BLOCK;
			goto done;
		}
		if (lane == 0) {
			chk_expect[rec] = ld32u(r + ip);
			chk_valid[rec] = 1;
		}
		ip += 4;
	}
	if (ip != flen)
		st = ST_TRAILING;
done:
	if (lane == 0)
		status[rec] = st;
}
