/* mt_pipe.c -- see mt_pipe.h */
#define _GNU_SOURCE /* cpu_set_t, pthread_setaffinity_np */
#include "mt_pipe.h"

#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/*
 * The reader's and the writer's work is a memcpy between the caller's memory and the pinned batch buffers, which the HIP runtime
 * places on the NUMA node of the device.  [MI355X box, 2 nodes] bound to that node the LZ4MT legs run at 20.9 / 25.5 GB/s, bound to
 * the other one at 16.5 / 19.5 (profiles/r06_sweeps/api_numa.txt) -- the two are the library's own threads (as the reference's
 * workers are, lib/lz4-mt_compress.c:207), so they go where their buffers are.
 */
int mt_bind_to_node(int node)
{
	const char *e = getenv("GPUMT_NUMA");
	char path[96], list[4096];
	cpu_set_t have, want;
	FILE *f;
	int any = 0, differs = 0;
	if (node < 0 || (e && *e == '0'))
		return 0;
	snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
	f = fopen(path, "r");
	if (!f)
		return 0;
	if (!fgets(list, sizeof list, f))
		list[0] = 0;
	fclose(f);
	if (pthread_getaffinity_np(pthread_self(), sizeof have, &have))
		return 0;
	CPU_ZERO(&want);
	for (const char *c = list; *c;) { /* "0-63,128-191" */
		char *end;
		long a = strtol(c, &end, 10), b;
		if (end == c)
			break;
		b = a;
		if (*end == '-') {
			c = end + 1;
			b = strtol(c, &end, 10);
			if (end == c)
				break;
		}
		for (long i = a; i <= b && i < CPU_SETSIZE; i++)
			if (i >= 0 && CPU_ISSET((int)i, &have)) {
				CPU_SET((int)i, &want);
				any = 1;
			}
		c = *end == ',' ? end + 1 : end;
		if (*end != ',')
			break;
	}
	if (!any)
		return 0;
	for (int i = 0; i < CPU_SETSIZE; i++)
		if (CPU_ISSET(i, &have) != CPU_ISSET(i, &want))
			differs = 1;
	if (!differs)
		return 0;
	return pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
}

int mt_nslot_for(int ndevices)
{
	const char *e = getenv("GPUMT_SLOTS");
	int v = e && *e ? atoi(e) : (ndevices > 1 ? 2 * ndevices + 2 : 4);
	if (ndevices < 1)
		ndevices = 1;
	if (v > 12 * ndevices) /* a device serves its slots on 12 kernel streams / completion marks */
		v = 12 * ndevices;
	return v < 2 ? 2 : v > MT_NSLOT ? MT_NSLOT : v;
}
int mt_nslot(void) { return mt_nslot_for(1); }

enum { S_FREE, S_FILLED, S_DONE };

typedef struct {
	const mt_pipe_ops *ops;
	void *arg;
	pthread_mutex_t mu;
	pthread_cond_t cv;
	int state[MT_NSLOT];
	int nslot;
	size_t err;        /* first error, sticky */
	long n_filled;     /* batches the reader has produced */
	long n_done;       /* batches whose results are in host memory */
	int reader_over;   /* the reader will produce no further batch */
	int device_over;   /* the device role will complete no further batch */
	/* GPUMT_TRACE=1: seconds each role spent working / waiting for a slot (stderr at the end) */
	double t_fill, t_fill_wait, t_drain, t_drain_wait, t_launch, t_complete, t_dev_wait;
} pipe_t;

static void fail(pipe_t *p, size_t err)
{
	pthread_mutex_lock(&p->mu);
	if (!p->err)
		p->err = err;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
}

static void *reader_main(void *a)
{
	pipe_t *p = (pipe_t *)a;
	if (p->ops->role_start)
		p->ops->role_start(p->arg);
	for (long b = 0;; b++) {
		const int s = (int)(b % p->nslot);
		int has_data = 0, eof = 0;
		size_t err;
		double t0 = now_s(), t1;
		pthread_mutex_lock(&p->mu);
		while (p->state[s] != S_FREE && !p->err)
			pthread_cond_wait(&p->cv, &p->mu);
		err = p->err;
		pthread_mutex_unlock(&p->mu);
		if (err)
			break;
		t1 = now_s();
		p->t_fill_wait += t1 - t0;
		err = p->ops->fill(p->arg, s, &has_data, &eof);
		p->t_fill += now_s() - t1;
		if (err) {
			fail(p, err);
			break;
		}
		pthread_mutex_lock(&p->mu);
		if (has_data) {
			p->state[s] = S_FILLED;
			p->n_filled++;
		}
		if (eof || !has_data)
			p->reader_over = 1;
		pthread_cond_broadcast(&p->cv);
		pthread_mutex_unlock(&p->mu);
		if (eof || !has_data)
			break;
	}
	pthread_mutex_lock(&p->mu);
	p->reader_over = 1;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
	return NULL;
}

static void *writer_main(void *a)
{
	pipe_t *p = (pipe_t *)a;
	if (p->ops->role_start)
		p->ops->role_start(p->arg);
	for (long b = 0;; b++) {
		const int s = (int)(b % p->nslot);
		size_t err;
		int stop;
		double t0 = now_s(), t1;
		pthread_mutex_lock(&p->mu);
		while (p->state[s] != S_DONE && !p->err && !(p->device_over && b >= p->n_done))
			pthread_cond_wait(&p->cv, &p->mu);
		stop = p->err || p->state[s] != S_DONE;
		pthread_mutex_unlock(&p->mu);
		if (stop)
			break;
		t1 = now_s();
		p->t_drain_wait += t1 - t0;
		err = p->ops->drain(p->arg, s);
		p->t_drain += now_s() - t1;
		if (err) {
			fail(p, err);
			break;
		}
		pthread_mutex_lock(&p->mu);
		p->state[s] = S_FREE;
		pthread_cond_broadcast(&p->cv);
		pthread_mutex_unlock(&p->mu);
	}
	return NULL;
}

/* wait until the slot is done being filled; 1 = go, 0 = no more batches / error */
static int wait_filled(pipe_t *p, long b)
{
	const int s = (int)(b % p->nslot);
	int go;
	pthread_mutex_lock(&p->mu);
	while (!(p->state[s] == S_FILLED && b < p->n_filled) && !p->err && !(p->reader_over && b >= p->n_filled))
		pthread_cond_wait(&p->cv, &p->mu);
	go = !p->err && p->state[s] == S_FILLED && b < p->n_filled;
	pthread_mutex_unlock(&p->mu);
	return go;
}

static void mark_done(pipe_t *p, long b)
{
	pthread_mutex_lock(&p->mu);
	p->state[b % p->nslot] = S_DONE;
	p->n_done = b + 1;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
}

size_t mt_pipe_run(const mt_pipe_ops *ops, void *arg) { return mt_pipe_run_n(ops, arg, mt_nslot()); }

size_t mt_pipe_run_n(const mt_pipe_ops *ops, void *arg, int nslot)
{
	pipe_t p;
	pthread_t rt, wt;
	long b = 0;
	size_t err;

	p.ops = ops;
	p.arg = arg;
	p.nslot = nslot < 2 ? 2 : nslot > MT_NSLOT ? MT_NSLOT : nslot;
	p.err = 0;
	p.n_filled = p.n_done = 0;
	p.reader_over = p.device_over = 0;
	p.t_fill = p.t_fill_wait = p.t_drain = p.t_drain_wait = p.t_launch = p.t_complete = p.t_dev_wait = 0;
	const double t_begin = now_s();
	const char *trace_env = getenv("GPUMT_TRACE");
	const int trace = trace_env && *trace_env ? atoi(trace_env) : 0;
	for (int i = 0; i < MT_NSLOT; i++)
		p.state[i] = S_FREE;
	pthread_mutex_init(&p.mu, NULL);
	pthread_cond_init(&p.cv, NULL);
	if (pthread_create(&rt, NULL, reader_main, &p)) {
		pthread_cond_destroy(&p.cv);
		pthread_mutex_destroy(&p.mu);
		return (size_t)-1; /* memory_allocation in every codec's enum */
	}
	if (pthread_create(&wt, NULL, writer_main, &p)) {
		fail(&p, (size_t)-1);
		pthread_join(rt, NULL);
		pthread_cond_destroy(&p.cv);
		pthread_mutex_destroy(&p.mu);
		return (size_t)-1;
	}
	/* device role on the calling thread: launch every batch as soon as it is filled and complete the
	 * oldest one once MT_NSLOT - 1 are in flight (the remaining slot is the one being filled or
	 * drained), so that the device always has several batches to overlap */
	{
		long done = 0; /* batches [done, b) are launched and not completed */
		for (;;) {
			double t0 = now_s(), t1;
			const int go = wait_filled(&p, b);
			t1 = now_s();
			p.t_dev_wait += t1 - t0;
			if (go) {
				err = ops->launch(arg, (int)(b % p.nslot));
				p.t_launch += now_s() - t1;
				if (trace > 1)
					fprintf(stderr, "[mt_pipe] %8.3f ms launch(%ld) took %.3f ms, waited %.3f ms for input\n",
						(t1 - t_begin) * 1e3, b, (now_s() - t1) * 1e3, (t1 - t0) * 1e3);
				if (err) {
					fail(&p, err);
					break;
				}
				b++;
			}
			err = 0;
			while (done < b && (!go || b - done >= p.nslot - 1)) {
				t0 = now_s();
				err = ops->complete(arg, (int)(done % p.nslot));
				p.t_complete += now_s() - t0;
				if (trace > 1)
					fprintf(stderr, "[mt_pipe] %8.3f ms complete(%ld) took %.3f ms\n", (t0 - t_begin) * 1e3, done,
						(now_s() - t0) * 1e3);
				if (err)
					break;
				mark_done(&p, done);
				done++;
			}
			if (err) {
				fail(&p, err);
				break;
			}
			if (!go)
				break;
		}
	}
	pthread_mutex_lock(&p.mu);
	p.device_over = 1;
	pthread_cond_broadcast(&p.cv);
	pthread_mutex_unlock(&p.mu);
	pthread_join(rt, NULL);
	pthread_join(wt, NULL);
	err = p.err;
	{
		if (trace)
			fprintf(stderr,
				"[mt_pipe] %ld batches in %.3f s | reader: fill %.3f wait %.3f | device: launch %.3f "
				"complete %.3f wait-for-input %.3f | writer: drain %.3f wait %.3f\n",
				b, now_s() - t_begin, p.t_fill, p.t_fill_wait, p.t_launch, p.t_complete, p.t_dev_wait,
				p.t_drain, p.t_drain_wait);
	}
	pthread_cond_destroy(&p.cv);
	pthread_mutex_destroy(&p.mu);
	return err;
}

size_t mt_pipe_run_inline(const mt_pipe_ops *ops, void *arg)
{
	for (long b = 0;; b++) {
		const int s = (int)(b % 2);
		int has_data = 0, eof = 0;
		size_t err = ops->fill(arg, s, &has_data, &eof);
		if (err)
			return err;
		if (has_data) {
			if ((err = ops->launch(arg, s)) != 0 || (err = ops->complete(arg, s)) != 0 ||
			    (err = ops->drain(arg, s)) != 0)
				return err;
		}
		if (eof || !has_data)
			break;
	}
	return 0;
}
