/* CPU harness for zstdmt_amd/csrc/host/mt_pipe.c (tests/test_host_pipe.py): the three-role pipeline with
 * synthetic roles -- no device.  fill produces numbered batches, launch / complete only flip flags,
 * drain checks that batches leave strictly in order and that a slot is never refilled before it was
 * drained; errors can be injected into any role at any batch. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "mt_pipe.h"

typedef struct {
	long nbatch, filled, drained;
	long fail_batch;
	int fail_role; /* 0 none, 1 fill, 2 launch, 3 complete, 4 drain */
	long slot_batch[MT_NSLOT];
	int slot_state[MT_NSLOT]; /* 0 free, 1 filled, 2 launched, 3 completed */
	int violation;
	int jitter;
} sim;

static void nap(sim *s, int role, long b)
{
	if (s->jitter)
		usleep((useconds_t)(((b * 7 + role * 13) % 5) * 200));
}
static size_t s_fill(void *a, int slot, int *has_data, int *eof)
{
	sim *s = (sim *)a;
	if (s->slot_state[slot] != 0)
		s->violation |= 1; /* refilled before it was drained */
	if (s->filled >= s->nbatch) {
		*has_data = 0;
		*eof = 1;
		return 0;
	}
	if (s->fail_role == 1 && s->filled == s->fail_batch)
		return (size_t)-5;
	nap(s, 1, s->filled);
	s->slot_batch[slot] = s->filled++;
	s->slot_state[slot] = 1;
	*has_data = 1;
	*eof = s->filled >= s->nbatch;
	return 0;
}
static size_t s_launch(void *a, int slot)
{
	sim *s = (sim *)a;
	if (s->slot_state[slot] != 1)
		s->violation |= 2;
	if (s->fail_role == 2 && s->slot_batch[slot] == s->fail_batch)
		return (size_t)-6;
	s->slot_state[slot] = 2;
	return 0;
}
static size_t s_complete(void *a, int slot)
{
	sim *s = (sim *)a;
	if (s->slot_state[slot] != 2)
		s->violation |= 4;
	if (s->fail_role == 3 && s->slot_batch[slot] == s->fail_batch)
		return (size_t)-7;
	nap(s, 3, s->slot_batch[slot]);
	s->slot_state[slot] = 3;
	return 0;
}
static size_t s_drain(void *a, int slot)
{
	sim *s = (sim *)a;
	if (s->slot_state[slot] != 3 || s->slot_batch[slot] != s->drained)
		s->violation |= 8; /* out of order, or not completed */
	if (s->fail_role == 4 && s->slot_batch[slot] == s->fail_batch)
		return (size_t)-8;
	nap(s, 4, s->drained);
	s->drained++;
	s->slot_state[slot] = 0;
	return 0;
}

/* argv: nbatch nslot inline(0/1) fail_role fail_batch jitter -> prints "rv drained violation" */
int main(int argc, char **argv)
{
	static const mt_pipe_ops ops = {s_fill, s_launch, s_complete, s_drain};
	sim s;
	size_t rv;
	if (argc < 7)
		return 2;
	memset(&s, 0, sizeof s);
	s.nbatch = atol(argv[1]);
	s.fail_role = atoi(argv[4]);
	s.fail_batch = atol(argv[5]);
	s.jitter = atoi(argv[6]);
	rv = atoi(argv[3]) ? mt_pipe_run_inline(&ops, &s) : mt_pipe_run_n(&ops, &s, atoi(argv[2]));
	printf("%ld %ld %d %d\n", (long)rv, s.drained, s.violation, mt_nslot_for(atoi(argv[2])));
	return 0;
}
