/* emu_runtime.cpp -- fiber scheduler of the host harness (see emu_runtime.h).  TEST HARNESS ONLY. */
#include "emu_runtime.h"

namespace emu {

Block *g_blk = nullptr;

static const size_t kStack = 256 * 1024;

#if !defined(__x86_64__)
#error "the fiber switch below is x86-64 System V only"
#endif
/* ucontext's swapcontext makes a signal-mask system call per switch, and the emulator switches at
 * every cross-lane operation of every lane: this one only moves the callee-saved registers. */
asm(".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "	pushq %rbp\n"
    "	pushq %rbx\n"
    "	pushq %r12\n"
    "	pushq %r13\n"
    "	pushq %r14\n"
    "	pushq %r15\n"
    "	movq %rsp, (%rdi)\n"
    "	movq %rsi, %rsp\n"
    "	popq %r15\n"
    "	popq %r14\n"
    "	popq %r13\n"
    "	popq %r12\n"
    "	popq %rbx\n"
    "	popq %rbp\n"
    "	ret\n"
    ".size emu_switch,.-emu_switch\n");

static void trampoline()
{
	Block *b = g_blk;
	b->body();
	b->fib[b->cur].done = true;
	emu_switch(&b->fib[b->cur].sp, b->sched_sp);
	abort(); /* a finished fiber is never resumed */
}

/* fiber stacks live as long as the process: a block of 1024 work-items would otherwise map and unmap
 * 256 MiB per launch */
static std::vector<char *> g_stacks;

static void run_block(Block &b)
{
	g_blk = &b;
	unsigned n = b.nthreads;
	b.fib.resize(n);
	b.slot.assign(n, 0);
	b.nwaves = (n + 63) / 64;
	const unsigned nbar = b.nwaves + 1 + (n + 15) / 16; /* wave barriers, the block barrier, 16-lane group barriers */
	b.arrived.assign(nbar, 0);
	b.gen.assign(nbar, 0);
	for (unsigned i = 0; i < n; i++) {
		Fiber &f = b.fib[i];
		if (g_stacks.size() <= i)
			g_stacks.push_back((char *)malloc(kStack));
		f.stack = g_stacks[i];
		f.done = false;
		f.tid = i;
		/* first switch "returns" into trampoline with the stack aligned as after a call */
		void **top = (void **)(((uintptr_t)f.stack + kStack) & ~(uintptr_t)15) - 2;
		top[0] = (void *)trampoline;
		top[1] = nullptr;
		for (int r = 1; r <= 6; r++)
			top[-r] = nullptr;
		f.sp = top - 6;
	}
	unsigned live = n;
	unsigned long idle_rounds = 0;
	/* EMU_REVERSE=1: lanes take their turns from the highest down.  Between two synchronisation points a lane
	 * then sees the stores of the lanes *above* it instead of those below: code that passes both ways does not
	 * depend on which neighbour happened to run first. */
	static const int order = getenv("EMU_REVERSE") ? atoi(getenv("EMU_REVERSE")) : 0;
	/* EMU_REVERSE=2: a fresh pseudo-random order every round (deterministic: the seed is the launch's shape) */
	std::vector<unsigned> perm(n);
	for (unsigned k = 0; k < n; k++)
		perm[k] = order == 1 ? n - 1 - k : k;
	unsigned long long rng = 0x9E3779B97F4A7C15ull ^ ((unsigned long long)b.block_idx.x << 20) ^ n;
	while (live) {
		unsigned progressed = 0;
		if (order == 2)
			for (unsigned k = n - 1; k > 0; k--) {
				rng = rng * 6364136223846793005ull + 1442695040888963407ull;
				const unsigned j = (unsigned)((rng >> 33) % (k + 1));
				const unsigned t = perm[k];
				perm[k] = perm[j];
				perm[j] = t;
			}
		for (unsigned k = 0; k < n; k++) {
			const unsigned i = perm[k];
			if (b.fib[i].done)
				continue;
			b.cur = i;
			/* snapshot barrier generations to detect progress */
			emu_switch(&b.sched_sp, b.fib[i].sp);
			if (b.fib[i].done) {
				live--;
				progressed++;
			}
		}
		/* deadlock detector: every live fiber is parked in a barrier that can never fill */
		unsigned waiting = 0;
		for (unsigned k = 0; k < nbar; k++)
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
