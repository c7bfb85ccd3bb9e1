/*
 * lz4_frame.h -- LZ4F frame-header parsing shared by the decoder kernels
 * (what LZ4F_decompress checks first; reference call site lib/lz4-mt_decompress.c:349-362).
 */
#ifndef ZMT_LZ4_FRAME_H
#define ZMT_LZ4_FRAME_H

#include "lz4_common.h"

/* order this wave's earlier global stores before its later global loads */
static __device__ __forceinline__ void wave_mem_fence()
{
#ifdef ZMT_EMU
	wv_sync();
#else
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

struct FrameInfo {
	u32 hdr;       /* header bytes */
	u32 blkmax;
	u32 has_csize, has_ccheck, indep;
	u32 has_bcheck; /* every block is followed by the XXH32 of its stored bytes */
	u64 csize;
};

/* wave-uniform parse + validation of the LZ4F frame header at f[0..flen) */
static __device__ u32 parse_frame_header(const u8 *f, u32 flen, FrameInfo &fi)
{
	u32 flg, bd, hdr;
	if (flen < 7 || uld32(f) != ZMT_LZ4F_MAGIC)
		return ST_BAD_FRAME;
	flg = uld8(f + 4);
	bd = uld8(f + 5);
	if ((flg >> 6) != 1 || (flg & 0x02) || (bd & 0x8F) || (bd >> 4) < 4)
		return ST_BAD_FRAME;
	/* block checksums and a dictionary id are valid LZ4F that lz4-mt never writes; liblz4 (behind
	 * lib/lz4-mt_decompress.c:349-362) decodes both -- the id is informational when no dictionary is
	 * attached -- and so does the wave-per-record decoder */
	fi.has_bcheck = (flg >> 4) & 1;
	fi.indep = (flg >> 5) & 1;
	fi.has_csize = (flg >> 3) & 1;
	fi.has_ccheck = (flg >> 2) & 1;
	fi.blkmax = 1u << (8 + 2 * (bd >> 4));
	hdr = 7 + (fi.has_csize ? 8 : 0) + ((flg & 0x01) ? 4 : 0);
	if (flen < hdr)
		return ST_BAD_FRAME;
	{
		u8 d[14];
		for (u32 i = 0; i < hdr - 5; i++)
			d[i] = (u8)uld8(f + 4 + i);
		if (uld8(f + hdr - 1) != ((xxh32_short(d, hdr - 5) >> 8) & 0xFF))
			return ST_BAD_FRAME;
	}
	fi.csize = fi.has_csize ? ((u64)uld32(f + 6) | (u64)uld32(f + 10) << 32) : 0;
	fi.hdr = hdr;
	return ST_OK;
}

#endif
