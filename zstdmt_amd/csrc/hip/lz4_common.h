/*
 * lz4_common.h -- constants and small device helpers shared by the LZ4 / XXH32 kernels.
 * Wire format: SURVEY.md Appendix A (record = 12-byte skippable header + one LZ4 frame as
 * LZ4F_compressFrame emits it for the prefs of /root/reference/lib/lz4-mt_compress.c:141-146).
 */
#ifndef ZMT_LZ4_COMMON_H
#define ZMT_LZ4_COMMON_H

#include "wave.h"

#define ZMT_SKIP_MAGIC 0x184D2A50u
#define ZMT_LZ4F_MAGIC 0x184D2204u
#define ZMT_BLOCK 65536u

/* status codes == include/gpumt.h GPUMT_ST_* */
enum {
	ST_OK = 0, ST_BAD_RECORD = 1, ST_BAD_FRAME = 2, ST_BAD_BLOCK = 3, ST_SIZE_MISMATCH = 4,
	ST_BAD_CHECKSUM = 5, ST_TRAILING = 6, ST_UNSUPPORTED = 7
};

#define XP1 2654435761u
#define XP2 2246822519u
#define XP3 3266489917u
#define XP4 668265263u
#define XP5 374761393u

static __device__ __forceinline__ u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
static __device__ __forceinline__ u32 xxh_round(u32 acc, u32 in)
{
	return rotl32(acc + in * XP2, 13) * XP1;
}
static __device__ __forceinline__ u32 xxh_avalanche(u32 h)
{
	h ^= h >> 15;
	h *= XP2;
	h ^= h >> 13;
	h *= XP3;
	h ^= h >> 16;
	return h;
}
/* tail of XXH32: fold the last (len & 15) bytes at p into h (h already includes += len) */
static __device__ __forceinline__ u32 xxh_tail(u32 h, const u8 *p, u32 rem)
{
	while (rem >= 4) {
		h = rotl32(h + ld32u(p) * XP3, 17) * XP4;
		p += 4;
		rem -= 4;
	}
	while (rem) {
		h = rotl32(h + (u32)*p * XP5, 11) * XP1;
		p++;
		rem--;
	}
	return xxh_avalanche(h);
}
/* whole XXH32 for inputs shorter than 16 bytes (frame-descriptor checksum) */
static __device__ __forceinline__ u32 xxh32_short(const u8 *p, u32 len)
{
	return xxh_tail(XP5 + len, p, len);
}

/* wave-uniform loads: every lane issues the same access, the result is pinned to an SGPR */
static __device__ __forceinline__ u32 uld8(const u8 *p) { return wv_readfirst((u32)*p); }
static __device__ __forceinline__ u32 uld16(const u8 *p) { return wv_readfirst(ld16u(p)); }
static __device__ __forceinline__ u32 uld32(const u8 *p) { return wv_readfirst(ld32u(p)); }

/* cooperative forward copy by one wave, source and destination do not overlap */
static __device__ __forceinline__ void wave_copy(u8 *d, const u8 *s, u32 n, int lane)
{
	u32 i = 0;
	if (n >= 512) {
		u32 n4 = n & ~255u;
		for (i = (u32)lane * 4; i < n4; i += 256)
			st32u(d + i, ld32u(s + i));
		i = n4;
	}
	for (i += (u32)lane; i < n; i += 64)
		d[i] = s[i];
}

#endif
