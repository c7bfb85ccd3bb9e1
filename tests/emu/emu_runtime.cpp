/* emu_runtime.cpp -- fiber scheduler of the host harness (see emu_runtime.h).  TEST HARNESS ONLY. */
#include "emu_runtime.h"

namespace emu {

Block *g_blk = nullptr;

static const size_t kStack = 256 * 1024;

static void trampoline()
{
	Block *b = g_blk;
	b->body();
	b->fib[b->cur].done = true;
	swapcontext(&b->fib[b->cur].ctx, &b->sched);
}

static void run_block(Block &b)
{
	g_blk = &b;
	unsigned n = b.nthreads;
	b.fib.resize(n);
	b.slot.assign(n, 0);
	b.nwaves = (n + 63) / 64;
	b.arrived.assign(b.nwaves + 1, 0);
	b.gen.assign(b.nwaves + 1, 0);
	for (unsigned i = 0; i < n; i++) {
		Fiber &f = b.fib[i];
		f.stack = (char *)malloc(kStack);
		f.done = false;
		f.tid = i;
		getcontext(&f.ctx);
		f.ctx.uc_stack.ss_sp = f.stack;
		f.ctx.uc_stack.ss_size = kStack;
		f.ctx.uc_link = nullptr;
		makecontext(&f.ctx, (void (*)())trampoline, 0);
	}
	unsigned live = n;
	unsigned long idle_rounds = 0;
	while (live) {
		unsigned progressed = 0;
		for (unsigned i = 0; i < n; i++) {
			if (b.fib[i].done)
				continue;
			b.cur = i;
			/* snapshot barrier generations to detect progress */
			swapcontext(&b.sched, &b.fib[i].ctx);
			if (b.fib[i].done) {
				live--;
				progressed++;
			}
		}
		/* deadlock detector: every live fiber is parked in a barrier that can never fill */
		unsigned waiting = 0;
		for (unsigned k = 0; k <= b.nwaves; k++)
			waiting += b.arrived[k];
		if (live && !progressed && waiting == live) {
			if (++idle_rounds > 4) {
				fprintf(stderr,
					"emu: deadlock -- cross-lane op or barrier in divergent control flow "
					"(block %u, %u fibers parked)\n",
					b.block_idx.x, live);
				abort();
			}
		} else {
			idle_rounds = 0;
		}
	}
	for (unsigned i = 0; i < n; i++)
		free(b.fib[i].stack);
	g_blk = nullptr;
}

void launch(dim3 grid, dim3 block, std::function<void()> body)
{
	for (unsigned bx = 0; bx < grid.x; bx++) {
		Block b;
		b.nthreads = block.x;
		b.body = body;
		b.block_idx = dim3{bx, 0, 0};
		b.block_dim = block;
		b.grid_dim = grid;
		run_block(b);
	}
}

} /* namespace emu */
