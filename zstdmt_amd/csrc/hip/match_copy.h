/*
 * match_copy.h -- LZ77 copy helpers shared by the zstd and brotli decoders: single-lane copies of
 * short runs (8-byte unaligned global accesses) and whole-wave copies of long ones, for plain and
 * overlapping (offset < length) matches.
 */
#ifndef ZMT_MATCH_COPY_H
#define ZMT_MATCH_COPY_H

#include "lz4_common.h"
#include "lz4_frame.h"

static __device__ __forceinline__ void st64g(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }

static __device__ __forceinline__ void g_copy(u8 *d, const u8 *s, u32 len)
{
	if (len >= 8) {
		/* ends first: two loads cover up to 16 bytes, four up to 32 -- all issued before the
		 * first store, so the common lengths cost one memory round trip; the middle of longer
		 * runs (rare: Z_CAP is 64) goes piece by piece */
		const u64 a = ld64u(s), b = ld64u(s + len - 8);
		if (len > 16) {
			const u64 c = ld64u(s + 8), e = ld64u(s + len - 16);
			st64g(d + 8, c);
			st64g(d + len - 16, e);
			for (u32 i = 16; i + 16 < len; i += 8)
				st64g(d + i, ld64u(s + i));
		}
		st64g(d, a);
		st64g(d + len - 8, b);
	} else if (len >= 4) {
		const u32 a = ld32u(s), b = ld32u(s + len - 4);
		st32u(d, a);
		st32u(d + len - 4, b);
	} else {
		for (u32 i = 0; i < len; i++)
			d[i] = s[i];
	}
}

/* match copy by one lane: source [d-off, ...) fully written and visible */
static __device__ __forceinline__ void g_match(u8 *d, u32 off, u32 ml)
{
	const u8 *s = d - off;
	if (off >= ml) {
		g_copy(d, s, ml);
	} else if (off >= 8) {
		/* overlapping, period >= 8: forward 8-byte steps only ever read bytes already written */
		u32 i = 0;
		for (; i + 8 <= ml; i += 8)
			st64g(d + i, ld64u(s + i));
		for (; i < ml; i++)
			d[i] = s[i];
	} else {
		u32 j = 0;
		for (u32 i = 0; i < ml; i++) {
			d[i] = s[j];
			if (++j == off)
				j = 0;
		}
	}
}

/* whole-wave copies for long runs (wave-uniform arguments) */
static __device__ void wave_match(u8 *d, u32 off, u32 ml, int lane)
{
	const u8 *s = d - off;
	if (off >= ml) {
		wave_copy(d, s, ml, lane);
	} else if (off >= 64) {
		/* period >= 64: chunks of `off` bytes are independent of each other's output only one
		 * period back; copy period by period */
		for (u32 done = 0; done < ml; done += off) {
			const u32 n = ml - done < off ? ml - done : off;
			wave_copy(d + done, s + done, n, lane);
			wave_mem_fence();
		}
	} else {
		for (u32 i = (u32)lane; i < ml; i += 64)
			d[i] = s[i % off];
	}
}


#endif
