/*
 * emu_runtime.h -- TEST HARNESS ONLY: runs the repo's HIP kernels on the host CPU so their logic
 * can be debugged (gdb/ASan) in a container without a GPU.  Never part of the product: the shipped
 * library is built by hipcc for gfx950 and has no CPU path.
 *
 * Model: one workgroup at a time, every work-item is a fiber (its own stack, switched by a few
 * instructions of x86-64 assembly in emu_runtime.cpp) on one OS thread.  Fibers
 * run to their next synchronisation point (wave-level cross-lane op, wv_sync, __syncthreads) in
 * lane order, i.e. the *opposite* extreme of hardware lockstep: code that forgets a wv_sync()
 * between a store by one lane and a load by another gives wrong answers here, which is the point.
 * Cross-lane operations must sit in wave-uniform control flow (all 64 lanes arrive) -- the same
 * discipline the kernels follow on hardware; a violated barrier aborts with a message.
 */
#ifndef ZMT_EMU_RUNTIME_H
#define ZMT_EMU_RUNTIME_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __restrict__ __restrict

namespace emu {

struct dim3 {
	unsigned x, y, z;
};

struct Fiber {
	void *sp; /* saved stack pointer while the fiber is switched out */
	char *stack;
	bool done;
	unsigned tid;
};

struct Block {
	unsigned nthreads, nwaves;
	std::vector<Fiber> fib;
	void *sched_sp;
	unsigned cur;
	/* barrier state: index 0..nwaves-1 = wave barriers, nwaves = block barrier */
	std::vector<unsigned> arrived, gen;
	/* cross-lane exchange slots, one per thread */
	std::vector<uint64_t> slot;
	std::function<void()> body;
	dim3 block_idx, block_dim, grid_dim;
};

extern Block *g_blk;

/* save the callee-saved registers and the stack pointer at *save_sp, continue on load_sp */
extern "C" void emu_switch(void **save_sp, void *load_sp);

inline unsigned tid() { return g_blk->cur; }

inline void yield_to_sched()
{
	Block *b = g_blk;
	emu_switch(&b->fib[b->cur].sp, b->sched_sp);
}

/* generic barrier over `count` participants identified by barrier index bi */
inline void barrier(unsigned bi, unsigned count)
{
	Block *b = g_blk;
	unsigned g = b->gen[bi];
	if (++b->arrived[bi] == count) {
		b->arrived[bi] = 0;
		b->gen[bi]++;
		return;
	}
	while (b->gen[bi] == g)
		yield_to_sched();
}

inline unsigned wave_size_of(unsigned w)
{
	Block *b = g_blk;
	unsigned lo = w * 64, hi = lo + 64;
	if (hi > b->nthreads)
		hi = b->nthreads;
	return hi - lo;
}

inline void wave_barrier()
{
	unsigned w = tid() / 64;
	barrier(w, wave_size_of(w));
}
inline void block_barrier() { barrier(g_blk->nwaves, g_blk->nthreads); }
/* barrier over the 16 lanes of a lane group (kernels that run one stream per 16-lane group in divergent control
 * flow, brotli_dec4.hip): slots nwaves + 1 + group */
inline void group_barrier16()
{
	Block *b = g_blk;
	const unsigned g = tid() / 16, lo = g * 16;
	unsigned hi = lo + 16;
	if (hi > b->nthreads)
		hi = b->nthreads;
	barrier(b->nwaves + 1 + g, hi - lo);
}

void launch(dim3 grid, dim3 block, std::function<void()> body);

} /* namespace emu */

struct emu_idx {
	unsigned x, y, z;
};
#define threadIdx (emu_idx{emu::tid(), 0, 0})
#define blockIdx (emu_idx{emu::g_blk->block_idx.x, 0, 0})
#define blockDim (emu_idx{emu::g_blk->block_dim.x, 1, 1})
#define gridDim (emu_idx{emu::g_blk->grid_dim.x, 1, 1})

inline void __syncthreads() { emu::block_barrier(); }

template <typename T> inline T atomicAdd(T *p, T v)
{
	T o = *p;
	*p = o + v;
	return o;
}
template <typename T> inline T atomicOr(T *p, T v)
{
	T o = *p;
	*p = o | v;
	return o;
}
template <typename T> inline T atomicMax(T *p, T v)
{
	T o = *p;
	if (v > o)
		*p = v;
	return o;
}

#endif
