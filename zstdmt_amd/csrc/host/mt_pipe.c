/* mt_pipe.c -- see mt_pipe.h */
#include "mt_pipe.h"

#include <pthread.h>
#include <stdlib.h>

int mt_nslot(void)
{
	static int n;
	if (!n) {
		const char *e = getenv("GPUMT_SLOTS");
		int v = e && *e ? atoi(e) : 4;
		n = v < 2 ? 2 : v > MT_NSLOT ? MT_NSLOT : v;
	}
	return n;
}

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
	for (long b = 0;; b++) {
		const int s = (int)(b % p->nslot);
		int has_data = 0, eof = 0;
		size_t err;
		pthread_mutex_lock(&p->mu);
		while (p->state[s] != S_FREE && !p->err)
			pthread_cond_wait(&p->cv, &p->mu);
		err = p->err;
		pthread_mutex_unlock(&p->mu);
		if (err)
			break;
		err = p->ops->fill(p->arg, s, &has_data, &eof);
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
	for (long b = 0;; b++) {
		const int s = (int)(b % p->nslot);
		size_t err;
		int stop;
		pthread_mutex_lock(&p->mu);
		while (p->state[s] != S_DONE && !p->err && !(p->device_over && b >= p->n_done))
			pthread_cond_wait(&p->cv, &p->mu);
		stop = p->err || p->state[s] != S_DONE;
		pthread_mutex_unlock(&p->mu);
		if (stop)
			break;
		err = p->ops->drain(p->arg, s);
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

size_t mt_pipe_run(const mt_pipe_ops *ops, void *arg)
{
	pipe_t p;
	pthread_t rt, wt;
	long b = 0;
	size_t err;

	p.ops = ops;
	p.arg = arg;
	p.nslot = mt_nslot();
	p.err = 0;
	p.n_filled = p.n_done = 0;
	p.reader_over = p.device_over = 0;
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
			const int go = wait_filled(&p, b);
			if (go) {
				err = ops->launch(arg, (int)(b % p.nslot));
				if (err) {
					fail(&p, err);
					break;
				}
				b++;
			}
			err = 0;
			while (done < b && (!go || b - done >= p.nslot - 1)) {
				err = ops->complete(arg, (int)(done % p.nslot));
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
