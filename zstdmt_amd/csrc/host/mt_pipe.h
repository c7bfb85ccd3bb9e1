/*
 * mt_pipe.h -- the three-role batch pipeline shared by the host engines.
 *
 * The reference runs T worker threads that each read a chunk, code it and write it
 * (lib/lz4-mt_compress.c:207-310); callbacks come from library threads, never two reads or two
 * writes at a time, but a read and a write may overlap (SURVEY.md 8b "Threading").  Here the work
 * moves in device batches through three roles on three threads:
 *
 *     reader thread : fill(slot)      fn_read into the slot's pinned input
 *     calling thread: launch(slot)    H2D + kernels + D2H, asynchronous
 *                     complete(slot)  wait until the slot's results are in pinned memory
 *     writer thread : drain(slot)     fn_write from the slot's pinned output, in order
 *
 * over n slots (4 by default, see mt_nslot_for): up to n - 1 batches are launched and not yet completed
 * (every slot has its own kernel stream, so their kernels overlap on the device -- or on several
 * devices, mt_host.h), the remaining slot is being read into or written out.  Order is positional
 * (batch b lives in slot b % n); the first error stops all roles and is returned.
 */
#ifndef ZMT_MT_PIPE_H
#define ZMT_MT_PIPE_H

#include <stddef.h>

#define MT_NSLOT 32 /* slots a context owns; the pipeline uses the first mt_nslot_for() of them */

/* slots in use: GPUMT_SLOTS (2..MT_NSLOT); default 4 on one device -- more slots keep more batches on
 * the device at once (the wave-per-record decoders are latency-bound per record) at the price of pinned
 * memory -- and 2 per device + 2 when a context spreads its slots over several devices (GPUMT_DEVICES) */
int mt_nslot(void);
int mt_nslot_for(int ndevices);

typedef struct {
	/* fill: *has_data = 0 when the input ended before anything was read into the slot; *eof = 1
	 * when no further batch follows this one.  All return 0 or a library error code. */
	size_t (*fill)(void *arg, int slot, int *has_data, int *eof);
	size_t (*launch)(void *arg, int slot);
	size_t (*complete)(void *arg, int slot);
	size_t (*drain)(void *arg, int slot);
	/* called once on the reader thread and once on the writer thread before their first batch (may be NULL): the
	 * engines bind the two threads to the CPUs next to the device's pinned memory (mt_bind_near, mt_host.h) */
	void (*role_start)(void *arg);
} mt_pipe_ops;

size_t mt_pipe_run(const mt_pipe_ops *ops, void *arg);             /* mt_nslot() slots */
size_t mt_pipe_run_n(const mt_pipe_ops *ops, void *arg, int nslot); /* nslot in 2..MT_NSLOT */

/* Binds the calling thread to the CPUs of NUMA node `node` of the host (those of them its affinity mask already allows; nothing
 * happens when there are none, when node < 0, or with GPUMT_NUMA=0 in the environment).  Returns 1 when the mask changed. */
int mt_bind_to_node(int node);

/* The same batches, every role on the calling thread, one batch at a time: what the reference does
 * for a decompress context with threads == 1 (lib/lz4-mt_decompress.c:528-534 calls pt_decompress
 * directly, so fn_read and fn_write run on the caller's thread). */
size_t mt_pipe_run_inline(const mt_pipe_ops *ops, void *arg);

#endif
