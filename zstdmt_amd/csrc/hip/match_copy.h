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

/* Two bodies of the single-lane copy.  ZMT_GCOPY_ONE_TRIP (defined by the translation unit before this header: the brotli
 * decoders) takes the one whose loads all come first -- one memory round trip per batch of copies; the zstd decoders measured
 * slower with every variant of it (reference-written streams 51.5-54.5 ms against 50.8) and keep the class-by-class body.
 * brotli dec4: 260.1 -> 252.3 ms.  (profiles/r06_sweeps/gcopy_one_trip.txt) */
#ifdef ZMT_GCOPY_ONE_TRIP
static __device__ __forceinline__ void g_copy(u8 *d, const u8 *s, u32 len)
{
	/* Lanes that take different branches run one after the other, and with in-order memory returns a branch that loads and
	 * stores pays its own load -> store round trip.  So ALL loads come first: every copy reads its first 8 bytes with the same
	 * instruction (a run shorter than 8 reads on behind itself: into output not written yet or the buffers' 64-byte slack),
	 * runs of 8 and more add three pieces, and the stores follow -- one round trip for the 64 copies of a batch whatever their
	 * lengths, in the registers the longest class needs anyway. */
	const bool big = len >= 8;
	const u32 o1 = len - 8 < 8 ? len - 8 : 8, o2 = (len > 16 ? len : 16) - 16, o3 = len - 8; /* (big only) */
	u64 a = 0, b = 0, c = 0, e = 0;
	if (len)
		a = ld64u(s);
	if (big) {
		/* 8..32 bytes: four 8-byte pieces at 0, min(8, len-8), max(len,16)-16, len-8 (they overlap for the shorter
		 * lengths); the middle of longer runs (rare: the caps are 64) piece by piece below */
		b = ld64u(s + o1);
		c = ld64u(s + o2);
		e = ld64u(s + o3);
	}
	if (big) {
		if (len > 32) {
			for (u32 i = 16; i + 16 < len; i += 8)
				st64g(d + i, ld64u(s + i));
		}
		st64g(d, a);
		st64g(d + o1, b);
		st64g(d + o2, c);
		st64g(d + o3, e);
	} else if (len >= 4) {
		st32u(d, (u32)a);
		st32u(d + len - 4, (u32)(a >> (8u * (len - 4u))));
	} else if (len) {
		d[0] = (u8)a;
		if (len > 1)
			d[1] = (u8)(a >> 8);
		if (len > 2)
			d[2] = (u8)(a >> 16);
	}
}

#else
static __device__ __forceinline__ void g_copy(u8 *d, const u8 *s, u32 len)
{
	/* Lanes that take different branches run one after the other, and with in-order memory returns
	 * every branch pays its own load -> store round trip: so three length classes only, each with
	 * all of its loads issued before its first store. */
	if (len >= 8) {
		/* 8..32 bytes: four 8-byte pieces at 0, min(8, len-8), max(len,16)-16, len-8 (they overlap
		 * for the shorter lengths); the middle of longer runs (rare: the caps are 64) piece by piece */
		const u32 o1 = len - 8 < 8 ? len - 8 : 8, o2 = (len > 16 ? len : 16) - 16, o3 = len - 8;
		const u64 a = ld64u(s), b = ld64u(s + o1), c = ld64u(s + o2), e = ld64u(s + o3);
		if (len > 32) {
			for (u32 i = 16; i + 16 < len; i += 8)
				st64g(d + i, ld64u(s + i));
		}
		st64g(d, a);
		st64g(d + o1, b);
		st64g(d + o2, c);
		st64g(d + o3, e);
	} else if (len >= 4) {
		const u32 a = ld32u(s), b = ld32u(s + len - 4);
		st32u(d, a);
		st32u(d + len - 4, b);
	} else if (len) {
		/* 1..3 bytes: first, middle, last */
		const u32 h = len >> 1;
		const u8 a = s[0], b = s[h], c = s[len - 1];
		d[0] = a;
		d[h] = b;
		d[len - 1] = c;
	}
}

#endif

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
