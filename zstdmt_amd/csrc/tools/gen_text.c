/*
 * gen_text.c -- seeded "enwik-style" synthetic text.
 *
 * No corpus can be shipped, so BASELINE.json's "8 GiB enwik-style synthetic" buffer is generated:
 * Zipf(s=1.07) over a 50 000-word vocabulary, word lengths ~Poisson(5.2) clipped to 1..14 from a
 * frequency-skewed 26-letter alphabet, n-gram redundancy (75 % of words are one of the previous
 * word's three habitual successors -- tuned so LZ4-1 lands at enwik-like ratios), separators {space 80 %, ", " 7 %, ". " 6 %, "\n" 2 %,
 * [[word]] 2 %, &quot; 1.5 %, decimal number 1 %, XML boilerplate 0.5 %}.
 *
 * The buffer is produced in independent 64 KiB segments, segment k seeded from (seed, k), so any
 * sub-range can be regenerated (CPU checker vs GPU-resident data) and generation threads scale.
 * Acceptance band: LZ4 level-1 ratio 1.7-1.9 at 128 KiB chunks (checked in tests/test_gen_text.py).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VOCAB   50000
#define SEG     65536
#define WMAX    14
#ifndef NSUCC
#define NSUCC 3
#endif
#ifndef BIGRAM_PM
#define BIGRAM_PM 750
#endif

typedef struct {
	uint8_t  len[VOCAB];
	char     w[VOCAB][16];
	uint32_t alias[VOCAB];
	uint32_t prob[VOCAB]; /* threshold scaled to 2^32 */
	int      ready;
} vocab_t;

static vocab_t g_vocab;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static inline uint64_t splitmix(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

/* English-like letter frequencies (per mille), cumulative table built at init */
static const uint16_t letter_pm[26] = { 82, 15, 28, 43, 127, 22, 20, 61, 70, 2, 8, 40, 24,
					 67, 75, 19, 1, 60, 63, 91, 28, 10, 24, 2, 20, 1 };

static void vocab_init(void)
{
	vocab_t *v = &g_vocab;
	uint64_t s = 0x5EEDF00DCAFEULL;
	uint16_t cum[26];
	double *p = (double *)malloc(sizeof(double) * VOCAB);
	uint32_t *small = (uint32_t *)malloc(sizeof(uint32_t) * VOCAB);
	uint32_t *large = (uint32_t *)malloc(sizeof(uint32_t) * VOCAB);
	double sum = 0;
	int i, ns = 0, nl = 0, acc = 0;

	for (i = 0; i < 26; i++) {
		acc += letter_pm[i];
		cum[i] = (uint16_t)acc;
	}
	for (i = 0; i < VOCAB; i++) {
		/* Poisson(5.2) by Knuth's product method, clipped to 1..14 */
		double L = exp(-5.2), pr = 1.0;
		int k = 0, j;
		do {
			k++;
			pr *= (double)(splitmix(&s) >> 11) * (1.0 / 9007199254740992.0);
		} while (pr > L);
		k -= 1;
		if (k < 1)
			k = 1;
		if (k > WMAX)
			k = WMAX;
		/* frequent (low-rank) words are short, as in natural language */
		if (i < 64 && k > 4)
			k = 1 + (int)(splitmix(&s) % 4);
		v->len[i] = (uint8_t)k;
		for (j = 0; j < k; j++) {
			unsigned r = (unsigned)(splitmix(&s) % (unsigned)acc), c = 0;
			while (cum[c] <= r)
				c++;
			v->w[i][j] = (char)('a' + c);
		}
		p[i] = 1.0 / pow((double)(i + 1), 1.07);
		sum += p[i];
	}
	/* Vose alias table */
	for (i = 0; i < VOCAB; i++) {
		p[i] = p[i] / sum * VOCAB;
		if (p[i] < 1.0)
			small[ns++] = (uint32_t)i;
		else
			large[nl++] = (uint32_t)i;
	}
	while (ns && nl) {
		uint32_t a = small[--ns], b = large[--nl];
		v->prob[a] = (uint32_t)(p[a] * 4294967295.0);
		v->alias[a] = b;
		p[b] = (p[b] + p[a]) - 1.0;
		if (p[b] < 1.0)
			small[ns++] = b;
		else
			large[nl++] = b;
	}
	while (nl) {
		uint32_t b = large[--nl];
		v->prob[b] = 0xFFFFFFFFu;
		v->alias[b] = b;
	}
	while (ns) {
		uint32_t a = small[--ns];
		v->prob[a] = 0xFFFFFFFFu;
		v->alias[a] = a;
	}
	free(p);
	free(small);
	free(large);
	v->ready = 1;
}

static inline uint32_t zipf_word(uint64_t r)
{
	uint32_t slot = (uint32_t)(((r >> 32) * (uint64_t)VOCAB) >> 32);
	return ((uint32_t)r <= g_vocab.prob[slot]) ? slot : g_vocab.alias[slot];
}

static inline size_t put(uint8_t *d, size_t o, size_t cap, const char *s, size_t n)
{
	if (n > cap - o)
		n = cap - o;
	memcpy(d + o, s, n);
	return o + n;
}

/* one 64 KiB segment (or the ragged tail), fully determined by (seed, segment index) */
static void gen_segment(uint8_t *d, size_t n, uint64_t seed, uint64_t seg)
{
	/* decorrelate segments: the per-segment state is a hash of (seed, seg), not an offset
	 * into one shared splitmix sequence */
	uint64_t k0 = seed ^ (seg * 0xD1342543DE82EF95ULL + 0x2545F4914F6CDD1DULL);
	uint64_t s = splitmix(&k0) ^ (splitmix(&k0) << 1);
	size_t o = 0;
	int cap_next = 1;
	uint32_t prev = (uint32_t)(splitmix(&s) % VOCAB);

	while (o < n) {
		uint64_t r = splitmix(&s);
		uint32_t wi = zipf_word(r);
		/* n-gram redundancy of natural text: with probability BIGRAM_PM/1000 the next word is
		 * one of the previous word's four habitual successors */
		if ((r >> 20) % 1000 < BIGRAM_PM) {
			uint64_t hk = (uint64_t)prev * 8 + ((r >> 40) % NSUCC) + 0x1234567ULL;
			wi = zipf_word(splitmix(&hk));
		}
		prev = wi;
		unsigned sel = (unsigned)(splitmix(&s) % 1000);
		const char *w = g_vocab.w[wi];
		size_t wl = g_vocab.len[wi], o0 = o;

		if (sel < 800 || sel >= 995) {
			o = put(d, o, n, w, wl);
			if (cap_next && o > o0) {
				d[o0] = (uint8_t)(d[o0] - 32);
				cap_next = 0;
			}
			if (sel >= 995) { /* XML boilerplate 0.5 % */
				static const char bp[] = "</text>\n</page>\n<page>\n<title>";
				static const char bq[] = "</title>\n<text xml:space=\"preserve\">";
				uint32_t t = zipf_word(splitmix(&s));
				o = put(d, o, n, bp, sizeof bp - 1);
				o = put(d, o, n, g_vocab.w[t], g_vocab.len[t]);
				o = put(d, o, n, bq, sizeof bq - 1);
				cap_next = 1;
			} else {
				o = put(d, o, n, " ", 1);
			}
		} else if (sel < 870) {
			o = put(d, o, n, w, wl);
			o = put(d, o, n, ", ", 2);
		} else if (sel < 930) {
			o = put(d, o, n, w, wl);
			o = put(d, o, n, ". ", 2);
			cap_next = 1;
		} else if (sel < 950) {
			o = put(d, o, n, w, wl);
			o = put(d, o, n, "\n", 1);
			cap_next = 1;
		} else if (sel < 970) {
			o = put(d, o, n, "[[", 2);
			o = put(d, o, n, w, wl);
			o = put(d, o, n, "]] ", 3);
		} else if (sel < 985) {
			o = put(d, o, n, "&quot;", 6);
			o = put(d, o, n, w, wl);
			o = put(d, o, n, "&quot; ", 7);
		} else { /* decimal number 1 % */
			char num[16];
			unsigned v = (unsigned)(r % 100000), k = 0, i;
			char tmp[8];
			do {
				tmp[k++] = (char)('0' + v % 10);
				v /= 10;
			} while (v);
			for (i = 0; i < k; i++)
				num[i] = tmp[k - 1 - i];
			num[k++] = ' ';
			o = put(d, o, n, num, k);
		}
	}
}

struct job {
	uint8_t *dst;
	size_t n;
	uint64_t seed, seg0;
	size_t next;
	pthread_mutex_t mu;
};

static void *worker(void *arg)
{
	struct job *j = (struct job *)arg;
	size_t nseg = (j->n + SEG - 1) / SEG;
	for (;;) {
		size_t k, off, len;
		pthread_mutex_lock(&j->mu);
		k = j->next;
		j->next += 16;
		pthread_mutex_unlock(&j->mu);
		if (k >= nseg)
			break;
		for (size_t e = k + 16 < nseg ? k + 16 : nseg; k < e; k++) {
			off = k * SEG;
			len = j->n - off < SEG ? j->n - off : SEG;
			gen_segment(j->dst + off, len, j->seed, j->seg0 + k);
		}
	}
	return NULL;
}

/*
 * Fill dst[0..n) with the synthetic text whose absolute byte offset starts at `offset`
 * (must be a multiple of 64 KiB), using `threads` generator threads.
 */
int zmt_gen_text(uint8_t *dst, size_t n, uint64_t seed, uint64_t offset, int threads)
{
	struct job j;
	pthread_t th[256];
	int t;

	if (offset % SEG)
		return -1;
	pthread_once(&g_once, vocab_init);
	memset(&j, 0, sizeof j);
	j.dst = dst;
	j.n = n;
	j.seed = seed;
	j.seg0 = offset / SEG;
	pthread_mutex_init(&j.mu, NULL);
	if (threads < 1)
		threads = 1;
	if (threads > 256)
		threads = 256;
	if (threads == 1) {
		worker(&j);
		return 0;
	}
	for (t = 0; t < threads; t++)
		pthread_create(&th[t], NULL, worker, &j);
	for (t = 0; t < threads; t++)
		pthread_join(th[t], NULL);
	return 0;
}

/* seeded PRNG bytes standing in for /dev/urandom (BASELINE config 1) */
int zmt_gen_random(uint8_t *dst, size_t n, uint64_t seed)
{
	uint64_t s = seed ^ 0xA5A5A5A55A5A5A5AULL;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) {
		uint64_t r = splitmix(&s);
		memcpy(dst + i, &r, 8);
	}
	if (i < n) {
		uint64_t r = splitmix(&s);
		memcpy(dst + i, &r, n - i);
	}
	return 0;
}
