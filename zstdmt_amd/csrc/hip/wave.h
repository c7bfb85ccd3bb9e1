/*
 * wave.h -- wave64 cross-lane primitives used by the gfx950 kernels.
 *
 * All cross-lane operations are used in wave-uniform control flow only.  wv_sync() orders one
 * lane's LDS/global stores before another lane's later loads inside the same wave (the hardware
 * issues a wave's DS and VMEM operations in order; the fence keeps the compiler from moving them).
 *
 * ZMT_EMU selects the host fiber harness of tests/emu (test infrastructure, never shipped).
 */
#ifndef ZMT_WAVE_H
#define ZMT_WAVE_H

#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#ifdef ZMT_EMU
/* ------------------------------------------------------------------ host fiber harness */
#include "emu_runtime.h"

static inline int wv_lane() { return (int)(emu::tid() & 63); }
static inline void wv_sync() { emu::wave_barrier(); }

static inline u64 wv_xchg_(u64 v, int src)
{
	emu::Block *b = emu::g_blk;
	unsigned base = emu::tid() & ~63u;
	b->slot[emu::tid()] = v;
	emu::wave_barrier();
	u64 r = b->slot[base + ((unsigned)src & 63)];
	emu::wave_barrier();
	return r;
}
static inline u32 wv_shfl(u32 v, int src) { return (u32)wv_xchg_(v, src); }
static inline u32 wv_readlane(u32 v, int lane) { return (u32)wv_xchg_(v, lane); }
/* uniform code: every lane already holds the value; the strict mode checks that claim */
#ifdef ZMT_EMU_STRICT
static inline u32 wv_readfirst(u32 v)
{
	u32 r = (u32)wv_xchg_(v, 0);
	if (r != v) {
		fprintf(stderr, "emu: readfirstlane on a non-uniform value\n");
		abort();
	}
	return r;
}
#else
static inline u32 wv_readfirst(u32 v) { return v; }
#endif
static inline u64 wv_ballot(bool p)
{
	emu::Block *b = emu::g_blk;
	unsigned base = emu::tid() & ~63u;
	b->slot[emu::tid()] = p ? 1 : 0;
	emu::wave_barrier();
	u64 m = 0;
	for (unsigned i = 0; i < emu::wave_size_of(emu::tid() / 64); i++)
		m |= (u64)(b->slot[base + i] & 1) << i;
	emu::wave_barrier();
	return m;
}
static inline int wv_popc(u64 m) { return __builtin_popcountll(m); }
static inline int wv_ffs(u64 m) { return __builtin_ffsll((long long)m); } /* 1-based, 0 if none */
/* number of set bits of m below this lane */
static inline u32 wv_mbcnt(u64 m) { return (u32)__builtin_popcountll(m & ((1ull << wv_lane()) - 1)); }
static inline void wv_sleep() {}
/* ---- 16-lane groups (one stream per group, groups in divergent control flow: brotli_dec4.hip): every lane of the
 * group arrives, other groups need not */
static inline u64 grp_xchg_(u64 v, int src16)
{
	emu::Block *b = emu::g_blk;
	unsigned base = emu::tid() & ~15u;
	b->slot[emu::tid()] = v;
	emu::group_barrier16();
	u64 r = b->slot[base + ((unsigned)src16 & 15)];
	emu::group_barrier16();
	return r;
}
static inline u32 grp_shfl(u32 v, int src16) { return (u32)grp_xchg_(v, src16); }
static inline u32 grp_ballot(bool p)
{
	emu::Block *b = emu::g_blk;
	unsigned base = emu::tid() & ~15u;
	b->slot[emu::tid()] = p ? 1 : 0;
	emu::group_barrier16();
	u32 m = 0;
	for (unsigned i = 0; i < 16 && base + i < b->nthreads; i++)
		m |= (u32)(b->slot[base + i] & 1) << i;
	emu::group_barrier16();
	return m;
}
static inline void grp_sync() { emu::group_barrier16(); }
/* value of lane k (0..3) of this lane's group of four */
static inline u32 wv_quad(u32 v, int k) { return wv_shfl(v, (wv_lane() & ~3) + k); }
#define ZMT_UNROLL

#else
/* ------------------------------------------------------------------------------ gfx950 */
#include <hip/hip_runtime.h>

static __device__ __forceinline__ int wv_lane() { return (int)__lane_id(); }
static __device__ __forceinline__ void wv_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
static __device__ __forceinline__ u32 wv_shfl(u32 v, int src)
{
	return (u32)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
}
/* lane must be wave-uniform */
static __device__ __forceinline__ u32 wv_readlane(u32 v, int lane)
{
	return (u32)__builtin_amdgcn_readlane((int)v, lane);
}
static __device__ __forceinline__ u32 wv_readfirst(u32 v)
{
	return (u32)__builtin_amdgcn_readfirstlane((int)v);
}
/* (the i1 builtin: __ballot(int) would first widen the predicate to 0 / 1 and compare it again) */
static __device__ __forceinline__ u64 wv_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
static __device__ __forceinline__ int wv_popc(u64 m) { return __popcll(m); }
static __device__ __forceinline__ int wv_ffs(u64 m) { return __ffsll((unsigned long long)m); }
static __device__ __forceinline__ u32 wv_mbcnt(u64 m)
{
	return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
}
static __device__ __forceinline__ void wv_sleep() { __builtin_amdgcn_s_sleep(2); }
/* ---- 16-lane groups (one stream per group, groups in divergent control flow: brotli_dec4.hip).  The hardware
 * runs the lanes of a wave in lockstep, so a group needs no barrier of its own: a ballot sees the active lanes, each
 * group takes its 16 bits; ds_bpermute reads inside the group */
static __device__ __forceinline__ u32 grp_shfl(u32 v, int src16)
{
	return (u32)__builtin_amdgcn_ds_bpermute((int)((((u32)__lane_id() & 48u) + ((u32)src16 & 15u)) << 2), (int)v);
}
static __device__ __forceinline__ u32 grp_ballot(bool p)
{
	const u64 m = __builtin_amdgcn_ballot_w64(p);
	return (u32)(m >> ((u32)__lane_id() & 48u)) & 0xFFFFu;
}
static __device__ __forceinline__ void grp_sync() { wv_sync(); }
/* value of lane k (constant 0..3) of this lane's group of four: one DPP quad_perm move */
#define wv_quad(v, k) ((u32)__builtin_amdgcn_update_dpp(0, (int)(v), (k) * 0x55, 0xf, 0xf, true))
#define ZMT_UNROLL _Pragma("unroll")
#endif

/* ---------------------------------------------------------------- shared by both builds */
static __device__ __forceinline__ bool wv_any(bool p) { return wv_ballot(p) != 0; }
static __device__ __forceinline__ bool wv_all(bool p) { return wv_ballot(!p) == 0; }

/* a value the optimiser must take as it is.  A ballot of `c && v > k` compiles to mask algebra on SGPR pairs plus a select and a
 * second compare that re-materialise the mask; the condition folded into the compared value -- wv_ballot(wv_opaque(c ? v : 0) > k)
 * -- is one select and one compare, but the fold only survives instruction combining behind this */
#ifdef ZMT_EMU
static inline u32 wv_opaque(u32 v) { return v; }
#else
static __device__ __forceinline__ u32 wv_opaque(u32 v)
{
	asm volatile("" : "+v"(v));
	return v;
}
#endif

/* inclusive prefix sum over the 64 lanes */
#ifdef ZMT_EMU
static inline u32 wv_scan_incl(u32 v)
{
	int l = wv_lane();
	for (int d = 1; d < 64; d <<= 1) {
		u32 o = wv_shfl(v, l - d);
		if (l >= d)
			v += o;
	}
	return v;
}
#else
/* DPP row shifts inside each 16-lane row, then row broadcasts (gfx9 row_bcast15 / row_bcast31).  bound_ctrl on
 * the shifts (a lane without a source reads 0) lets the compiler fold each step into ONE v_add_u32_dpp; with
 * bound_ctrl off it emitted v_mov old + v_mov_dpp + v_add per step (18 instructions instead of 6) */
static __device__ __forceinline__ u32 wv_scan_incl(u32 v)
{
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); /* row_shr:1 */
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); /* row_shr:2 */
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); /* row_shr:4 */
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); /* row_shr:8 */
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 -> rows 1,3 */
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); /* row_bcast:31 -> rows 2,3 */
	return v;
}
#endif

/* inclusive prefix maximum (unsigned) over the 64 lanes; lane-1's value with `fill` for lane 0;
 * bytes sh.. of the 64-bit value hi:lo (sh = 0..3) */
#ifdef ZMT_EMU
static inline u32 wv_scan_max_incl(u32 v)
{
	int l = wv_lane();
	for (int d = 1; d < 64; d <<= 1) {
		u32 o = wv_shfl(v, l - d);
		if (l >= d && o > v)
			v = o;
	}
	return v;
}
static inline u32 wv_shr1(u32 v, u32 fill)
{
	int l = wv_lane();
	u32 o = wv_shfl(v, l - 1);
	return l ? o : fill;
}
/* lane l receives lane l+1's value, lane 63 `fill` */
static inline u32 wv_shl1(u32 v, u32 fill)
{
	int l = wv_lane();
	u32 o = wv_shfl(v, l + 1);
	return l < 63 ? o : fill;
}
static inline u32 wv_alignbyte(u32 hi, u32 lo, u32 sh) { return (u32)((((u64)hi << 32) | lo) >> (8 * (sh & 3))); }
#else
static __device__ __forceinline__ u32 wv_umax(u32 a, u32 b) { return a > b ? a : b; }
static __device__ __forceinline__ u32 wv_scan_max_incl(u32 v)
{
	v = wv_umax(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true)); /* row_shr:1 */
	v = wv_umax(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true)); /* row_shr:2 */
	v = wv_umax(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true)); /* row_shr:4 */
	v = wv_umax(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true)); /* row_shr:8 */
	v = wv_umax(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 */
	v = wv_umax(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 */
	return v;
}
/* wave_shr:1 (gfx9 DPP): lane l receives lane l-1's value, lane 0 keeps `fill` */
static __device__ __forceinline__ u32 wv_shr1(u32 v, u32 fill)
{
	return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}
/* wave_shl:1 (gfx9 DPP): lane l receives lane l+1's value, lane 63 keeps `fill` */
static __device__ __forceinline__ u32 wv_shl1(u32 v, u32 fill)
{
	return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
}
static __device__ __forceinline__ u32 wv_alignbyte(u32 hi, u32 lo, u32 sh)
{
	return (u32)__builtin_amdgcn_alignbyte(hi, lo, sh);
}
#endif

/* unaligned little-endian accesses (gfx950 global memory handles misaligned dwords natively) */
static __device__ __forceinline__ u32 ld32u(const u8 *p)
{
	u32 v;
	__builtin_memcpy(&v, p, 4);
	return v;
}
static __device__ __forceinline__ u64 ld64u(const u8 *p)
{
	u64 v;
	__builtin_memcpy(&v, p, 8);
	return v;
}
static __device__ __forceinline__ u32 ld16u(const u8 *p)
{
	u16 v;
	__builtin_memcpy(&v, p, 2);
	return v;
}
static __device__ __forceinline__ void st32u(u8 *p, u32 v) { __builtin_memcpy(p, &v, 4); }
static __device__ __forceinline__ void st16u(u8 *p, u32 v)
{
	u16 w = (u16)v;
	__builtin_memcpy(p, &w, 2);
}

#endif
