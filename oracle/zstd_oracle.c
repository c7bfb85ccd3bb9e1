/*
 * zstd_oracle.c -- CPU restatement of the zstd-mt decode path.  TEST INFRASTRUCTURE ONLY
 * (same rule as zmt_oracle.h: nothing under zstdmt_amd/ or include/ may link or call this).
 *
 * What it restates (reference = /root/reference, mcmilk/zstdmt @ 2025-10-17):
 *   - the record parser of lib/zstd-mt_decompress.c:209-369 (pt_read, "pzstd style": a 12-byte
 *     0x184D2A50 skippable frame whose 4-byte payload is the size of the zstd frame that follows,
 *     which is what lib/zstd-mt_compress.c:296-302 emits) and the per-record decode of
 *     pt_decompress :371-549 (exactly one zstd frame per record);
 *   - the third-party arithmetic that path calls, which is NOT in the reference tree: zstd v1.5.6
 *     (pinned at programs/Makefile:11) ZSTD_decompressStream (lib/zstd-mt_decompress.c:464).
 *     The published format is restated from RFC 8878 (Zstandard frame format): frame header,
 *     raw / RLE / compressed blocks, Huffman literals (direct and FSE-compressed weights, 1 or 4
 *     streams, treeless), FSE sequence tables (predefined / RLE / compressed / repeat), repeat
 *     offsets, XXH64 content checksum.  No dictionaries.
 *
 * The compress side of zstd-mt has no byte-level oracle: zstd's compressed bytes differ between
 * versions (1.4.9 in this image, 1.5.6 pinned) and the bar for this codec is decompress-identical
 * (SURVEY 8a row C4): a frame written by the HIP encoder is correct iff this decoder AND the
 * reference build (oracle/_ref/libzstdmt_ref.so -> libzstd) decode it to the original bytes.
 *
 * Parity pinning: checked against streams produced by the reference's own lib/zstd-mt_*.c built in
 * place against the image's libzstd 1.4.9 (oracle/ref/Makefile), levels 1..19, see
 * tests/golden/zstd/ and tests/test_zstd_oracle.py.
 */
#include "zmt_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ZERR ((size_t)-1)

static inline uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static inline uint32_t rd24(const uint8_t *p) { return rd16(p) | ((uint32_t)p[2] << 16); }
static inline uint32_t rd32(const uint8_t *p) { return rd16(p) | (rd16(p + 2) << 16); }
static inline uint64_t rd64(const uint8_t *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static inline int highbit(uint32_t v) { return 31 - __builtin_clz(v); }

/* ------------------------------------------------------------------ XXH64 (content checksum) */
#define P64_1 0x9E3779B185EBCA87ull
#define P64_2 0xC2B2AE3D27D4EB4Full
#define P64_3 0x165667B19E3779F9ull
#define P64_4 0x85EBCA77C2B2AE63ull
#define P64_5 0x27D4EB2F165667C5ull
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t x64_round(uint64_t acc, uint64_t in)
{
	acc += in * P64_2;
	return rotl64(acc, 31) * P64_1;
}
static inline uint64_t x64_merge(uint64_t acc, uint64_t v)
{
	acc ^= x64_round(0, v);
	return acc * P64_1 + P64_4;
}

uint64_t zo_xxh64(const void *data, size_t len, uint64_t seed)
{
	const uint8_t *p = (const uint8_t *)data, *end = p + len;
	uint64_t h;
	if (len >= 32) {
		uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
		do {
			v1 = x64_round(v1, rd64(p));
			v2 = x64_round(v2, rd64(p + 8));
			v3 = x64_round(v3, rd64(p + 16));
			v4 = x64_round(v4, rd64(p + 24));
			p += 32;
		} while (p + 32 <= end);
		h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
		h = x64_merge(h, v1);
		h = x64_merge(h, v2);
		h = x64_merge(h, v3);
		h = x64_merge(h, v4);
	} else {
		h = seed + P64_5;
	}
	h += (uint64_t)len;
	while (p + 8 <= end) {
		h ^= x64_round(0, rd64(p));
		h = rotl64(h, 27) * P64_1 + P64_4;
		p += 8;
	}
	if (p + 4 <= end) {
		h ^= (uint64_t)rd32(p) * P64_1;
		h = rotl64(h, 23) * P64_2 + P64_3;
		p += 4;
	}
	while (p < end) {
		h ^= (uint64_t)(*p++) * P64_5;
		h = rotl64(h, 11) * P64_1;
	}
	h ^= h >> 33;
	h *= P64_2;
	h ^= h >> 29;
	h *= P64_3;
	h ^= h >> 32;
	return h;
}

/* ------------------------------------------------------------------ bit readers */
/* forward reader (FSE table descriptions): bits are consumed from bit 0 of byte 0 upwards */
typedef struct {
	const uint8_t *p;
	size_t len;
	size_t bit;
} fwd_bits;

static uint32_t fwd_peek(const fwd_bits *b, int n) /* n <= 24; bytes past the end read as 0 */
{
	uint64_t v = 0;
	size_t byte = b->bit >> 3;
	for (int i = 0; i < 5; i++)
		if (byte + i < b->len)
			v |= (uint64_t)b->p[byte + i] << (8 * i);
	return (uint32_t)(v >> (b->bit & 7)) & ((1u << n) - 1);
}

/* backward reader (FSE / Huffman payloads): the last byte carries a 1-bit end mark above the
 * first bit to read; bits are consumed from there downwards.  pos = number of unread bits;
 * it may go negative (reads past the start deliver zeros), which callers treat as the end or as
 * corruption depending on context (RFC 8878 4.1). */
typedef struct {
	const uint8_t *p;
	int64_t pos;
} rev_bits;

static int rev_init(rev_bits *b, const uint8_t *p, size_t len)
{
	if (len == 0 || p[len - 1] == 0)
		return -1;
	b->p = p;
	b->pos = (int64_t)8 * (int64_t)(len - 1) + highbit(p[len - 1]);
	return 0;
}

static uint64_t rev_window(const rev_bits *b, int64_t lo, int n) /* bits [lo, lo+n), n <= 56 */
{
	uint64_t v = 0;
	if (n == 0)
		return 0;
	int64_t shift = 0;
	if (lo < 0) { /* missing low bits are zeros */
		shift = -lo;
		if (shift >= n)
			return 0;
		n -= (int)shift;
		lo = 0;
	}
	int64_t byte = lo >> 3;
	int64_t last = (lo + n - 1) >> 3;
	for (int64_t i = byte; i <= last; i++)
		v |= (uint64_t)b->p[i] << (8 * (i - byte));
	v = (v >> (lo & 7)) & (((uint64_t)1 << n) - 1);
	return v << shift;
}

static uint64_t rev_read(rev_bits *b, int n)
{
	b->pos -= n;
	return rev_window(b, b->pos, n);
}
static uint64_t rev_peek(const rev_bits *b, int n) { return rev_window(b, b->pos - n, n); }

/* ------------------------------------------------------------------ FSE decoding tables */
typedef struct {
	uint8_t sym, nb;
	uint16_t base; /* next state = base + read(nb) */
} fse_cell;
typedef struct {
	fse_cell cell[512];
	int log;
	int valid;
} fse_table;

/* RFC 8878 4.1.1: spread symbols over the table, then number the occurrences of each symbol */
static int fse_build(fse_table *t, const int16_t *norm, int nsym, int log)
{
	const uint32_t size = 1u << log, mask = size - 1;
	uint16_t next[256];
	uint32_t high = size - 1, pos = 0;
	const uint32_t step = (size >> 1) + (size >> 3) + 3;
	for (int s = 0; s < nsym; s++) {
		if (norm[s] == -1) {
			t->cell[high--].sym = (uint8_t)s;
			next[s] = 1;
		} else {
			next[s] = (uint16_t)norm[s];
		}
	}
	for (int s = 0; s < nsym; s++) {
		for (int i = 0; i < norm[s]; i++) {
			t->cell[pos].sym = (uint8_t)s;
			do
				pos = (pos + step) & mask;
			while (pos > high);
		}
	}
	if (pos != 0)
		return -1;
	for (uint32_t u = 0; u < size; u++) {
		const uint32_t x = next[t->cell[u].sym]++;
		const int nb = log - highbit(x);
		t->cell[u].nb = (uint8_t)nb;
		t->cell[u].base = (uint16_t)((x << nb) - size);
	}
	t->log = log;
	t->valid = 1;
	return 0;
}

static void fse_rle(fse_table *t, int sym)
{
	t->cell[0].sym = (uint8_t)sym;
	t->cell[0].nb = 0;
	t->cell[0].base = 0;
	t->log = 0;
	t->valid = 1;
}

/* RFC 8878 4.1.1 table description.  Returns bytes consumed or -1. */
static int fse_read_ncount(const uint8_t *p, size_t len, int16_t *norm, int *nsym_io, int *log_out,
			   int max_log)
{
	fwd_bits b = {p, len, 0};
	const int max_sym = *nsym_io; /* number of symbols the alphabet allows */
	if (len < 1)
		return -1;
	int log = (int)fwd_peek(&b, 4) + 5;
	b.bit += 4;
	if (log > max_log)
		return -1;
	int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
	int prev0 = 0;
	memset(norm, 0, sizeof(int16_t) * (size_t)max_sym);
	while (remaining > 1 && sym < max_sym) {
		if (prev0) {
			/* 2-bit repeat flags: number of further zero-probability symbols */
			for (;;) {
				int r = (int)fwd_peek(&b, 2);
				b.bit += 2;
				sym += r;
				if (r != 3)
					break;
			}
			if (sym >= max_sym)
				return -1;
			prev0 = 0;
			continue; /* a regular value follows */
		}
		{
			const int max = (2 * threshold - 1) - remaining;
			int count;
			uint32_t v = fwd_peek(&b, nbits);
			if ((int)(v & (uint32_t)(threshold - 1)) < max) {
				count = (int)(v & (uint32_t)(threshold - 1));
				b.bit += (size_t)(nbits - 1);
			} else {
				count = (int)(v & (uint32_t)(2 * threshold - 1));
				if (count >= threshold)
					count -= max;
				b.bit += (size_t)nbits;
			}
			count--; /* value 0 means "less than one": probability -1 */
			remaining -= count < 0 ? -count : count;
			norm[sym++] = (int16_t)count;
			prev0 = (count == 0);
			while (remaining < threshold) {
				nbits--;
				threshold >>= 1;
			}
		}
	}
	if (remaining != 1)
		return -1;
	if ((b.bit + 7) / 8 > len)
		return -1;
	*nsym_io = sym;
	*log_out = log;
	return (int)((b.bit + 7) / 8);
}

/* ------------------------------------------------------------------ Huffman literals */
typedef struct {
	uint8_t sym[2048], nb[2048];
	int log;
	int valid;
} huf_table;

/* RFC 8878 4.2.1: weights -> prefix code table.  nw explicit weights, the last one is implied. */
static int huf_build(huf_table *h, uint8_t *w, int nw)
{
	uint32_t total = 0;
	for (int i = 0; i < nw; i++) {
		if (w[i] > 11)
			return -1;
		total += w[i] ? (1u << (w[i] - 1)) : 0;
	}
	if (total == 0)
		return -1;
	const int log = highbit(total) + 1;
	if (log > 11)
		return -1;
	const uint32_t rest = (1u << log) - total;
	if (rest & (rest - 1))
		return -1; /* must be a power of two */
	w[nw] = (uint8_t)(highbit(rest) + 1);
	nw++;
	uint32_t start[13] = {0}, cnt[13] = {0};
	for (int i = 0; i < nw; i++)
		cnt[w[i]]++;
	if (cnt[1] < 2 || (cnt[1] & 1))
		return -1; /* as upstream: at least two, and an even number of, longest codes */
	uint32_t at = 0;
	for (int r = 1; r <= log; r++) {
		start[r] = at;
		at += cnt[r] << (r - 1);
	}
	for (int s = 0; s < nw; s++) {
		const int r = w[s];
		if (!r)
			continue;
		const uint32_t n = 1u << (r - 1);
		for (uint32_t i = 0; i < n; i++) {
			h->sym[start[r] + i] = (uint8_t)s;
			h->nb[start[r] + i] = (uint8_t)(log + 1 - r);
		}
		start[r] += n;
	}
	h->log = log;
	h->valid = 1;
	return 0;
}

/* tree description: returns bytes consumed or -1 */
static int huf_read_tree(huf_table *h, const uint8_t *p, size_t len)
{
	uint8_t w[256];
	int nw = 0;
	if (len < 1)
		return -1;
	const int hb = p[0];
	if (hb >= 128) {
		/* direct: 4-bit weights, high nibble first */
		nw = hb - 127;
		const int bytes = (nw + 1) / 2;
		if ((size_t)(1 + bytes) > len)
			return -1;
		for (int i = 0; i < nw; i++)
			w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
		if (huf_build(h, w, nw))
			return -1;
		return 1 + bytes;
	}
	/* FSE-compressed weights: table (accuracy <= 6), then two interleaved states */
	if ((size_t)(1 + hb) > len || hb < 1)
		return -1;
	int16_t norm[13];
	int nsym = 13, log;
	int used = fse_read_ncount(p + 1, (size_t)hb, norm, &nsym, &log, 6);
	if (used < 0 || used >= hb)
		return -1;
	fse_table t;
	if (fse_build(&t, norm, nsym, log))
		return -1;
	rev_bits b;
	if (rev_init(&b, p + 1 + used, (size_t)(hb - used)))
		return -1;
	uint32_t s1 = (uint32_t)rev_read(&b, log), s2 = (uint32_t)rev_read(&b, log);
	if (b.pos < 0)
		return -1;
	for (;;) {
		if (nw > 253)
			return -1;
		w[nw++] = t.cell[s1].sym;
		s1 = t.cell[s1].base + (uint32_t)rev_read(&b, t.cell[s1].nb);
		if (b.pos < 0) {
			w[nw++] = t.cell[s2].sym;
			break;
		}
		if (nw > 253)
			return -1;
		w[nw++] = t.cell[s2].sym;
		s2 = t.cell[s2].base + (uint32_t)rev_read(&b, t.cell[s2].nb);
		if (b.pos < 0) {
			w[nw++] = t.cell[s1].sym;
			break;
		}
	}
	if (huf_build(h, w, nw))
		return -1;
	return 1 + hb;
}

static int huf_stream(const huf_table *h, const uint8_t *p, size_t len, uint8_t *dst, size_t n)
{
	rev_bits b;
	if (rev_init(&b, p, len))
		return -1;
	for (size_t i = 0; i < n; i++) {
		const uint32_t v = (uint32_t)rev_peek(&b, h->log);
		dst[i] = h->sym[v];
		b.pos -= h->nb[v];
	}
	return b.pos == 0 ? 0 : -1; /* every stream must end exactly on its first bit */
}

/* ------------------------------------------------------------------ sequences */
static const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
				       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
				       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
				       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
				       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,   9,   10,  11,
				     12, 13, 14, 15, 16, 18, 20, 22, 24,  28,  32,  40,
				     48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
				    1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16,
				     17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30,
				     31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83,
				     99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
				    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
				    2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

typedef struct {
	huf_table huf;
	fse_table ll, of, ml;
	uint32_t rep[3];
	uint8_t *lit; /* 128 KiB + slack */
} frame_state;

/* one sequences-section table: mode from the Symbol_Compression_Modes byte */
static int seq_table(fse_table *t, int mode, const uint8_t **pp, const uint8_t *end, const int16_t *def,
		     int def_n, int def_log, int max_sym, int max_log)
{
	const uint8_t *p = *pp;
	switch (mode) {
	case 0:
		return fse_build(t, def, def_n, def_log);
	case 1:
		if (p >= end || *p >= max_sym)
			return -1;
		fse_rle(t, *p);
		*pp = p + 1;
		return 0;
	case 2: {
		int16_t norm[64];
		int nsym = max_sym, log;
		int used = fse_read_ncount(p, (size_t)(end - p), norm, &nsym, &log, max_log);
		if (used < 0 || fse_build(t, norm, nsym, log))
			return -1;
		*pp = p + used;
		return 0;
	}
	default:
		return t->valid ? 0 : -1; /* repeat: the table of the previous compressed block */
	}
}

/* one compressed block -> dst[opos...]; returns new opos or ZERR */
static size_t block_decode(frame_state *fs, const uint8_t *src, size_t n, uint8_t *dst, size_t opos,
			   size_t cap, size_t block_max)
{
	const uint8_t *p = src, *end = src + n;
	if (n < 2)
		return ZERR;
	/* ---- literals section (RFC 8878 3.1.1.3.1) ---- */
	const int ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
	size_t regen, csize = 0;
	int streams = 1;
	const uint8_t *lit;
	if (ltype < 2) {
		if (sf == 0 || sf == 2) {
			regen = p[0] >> 3;
			p += 1;
		} else if (sf == 1) {
			regen = rd16(p) >> 4;
			p += 2;
		} else {
			if (n < 3)
				return ZERR;
			regen = rd24(p) >> 4;
			p += 3;
		}
		if (regen > block_max)
			return ZERR;
		if (ltype == 0) {
			if ((size_t)(end - p) < regen)
				return ZERR;
			lit = p;
			p += regen;
		} else {
			if (p >= end)
				return ZERR;
			memset(fs->lit, *p, regen);
			lit = fs->lit;
			p += 1;
		}
	} else {
		if (n < 5)
			return ZERR;
		if (sf < 2) {
			const uint32_t v = rd24(p);
			regen = (v >> 4) & 1023;
			csize = v >> 14;
			streams = sf == 0 ? 1 : 4;
			p += 3;
		} else if (sf == 2) {
			const uint32_t v = rd32(p);
			regen = (v >> 4) & 16383;
			csize = v >> 18;
			streams = 4;
			p += 4;
		} else {
			const uint64_t v = (uint64_t)rd32(p) | ((uint64_t)p[4] << 32);
			regen = (size_t)((v >> 4) & 262143);
			csize = (size_t)(v >> 22);
			streams = 4;
			p += 5;
		}
		if (regen > block_max || (size_t)(end - p) < csize || regen == 0)
			return ZERR;
		const uint8_t *lp = p, *lend = p + csize;
		p = lend;
		if (ltype == 2) {
			int used = huf_read_tree(&fs->huf, lp, (size_t)(lend - lp));
			if (used < 0)
				return ZERR;
			lp += used;
		} else if (!fs->huf.valid) {
			return ZERR; /* treeless without a previous tree */
		}
		if (streams == 1) {
			if (huf_stream(&fs->huf, lp, (size_t)(lend - lp), fs->lit, regen))
				return ZERR;
		} else {
			if (lend - lp < 10)
				return ZERR; /* 6-byte jump table + at least one byte per stream */
			const size_t s1 = rd16(lp), s2 = rd16(lp + 2), s3 = rd16(lp + 4);
			lp += 6;
			if (s1 + s2 + s3 >= (size_t)(lend - lp))
				return ZERR;
			const size_t s4 = (size_t)(lend - lp) - s1 - s2 - s3;
			const size_t q = (regen + 3) / 4;
			if (3 * q > regen)
				return ZERR;
			if (huf_stream(&fs->huf, lp, s1, fs->lit, q) ||
			    huf_stream(&fs->huf, lp + s1, s2, fs->lit + q, q) ||
			    huf_stream(&fs->huf, lp + s1 + s2, s3, fs->lit + 2 * q, q) ||
			    huf_stream(&fs->huf, lp + s1 + s2 + s3, s4, fs->lit + 3 * q, regen - 3 * q))
				return ZERR;
		}
		lit = fs->lit;
	}
	/* ---- sequences section (RFC 8878 3.1.1.3.2) ---- */
	if (p >= end)
		return ZERR;
	uint32_t nseq = *p++;
	if (nseq >= 128) {
		if (nseq == 255) {
			if (end - p < 2)
				return ZERR;
			nseq = rd16(p) + 0x7F00;
			p += 2;
		} else {
			if (p >= end)
				return ZERR;
			nseq = ((nseq - 128) << 8) + *p++;
		}
	}
	const size_t bstart = opos;
	size_t lpos = 0;
	if (nseq) {
		if (p >= end)
			return ZERR;
		const int modes = *p++;
		if (modes & 3)
			return ZERR;
		if (seq_table(&fs->ll, modes >> 6, &p, end, LL_DEFAULT, 36, 6, 36, 9) ||
		    seq_table(&fs->of, (modes >> 4) & 3, &p, end, OF_DEFAULT, 29, 5, 32, 8) ||
		    seq_table(&fs->ml, (modes >> 2) & 3, &p, end, ML_DEFAULT, 53, 6, 53, 9))
			return ZERR;
		rev_bits b;
		if (rev_init(&b, p, (size_t)(end - p)))
			return ZERR;
		uint32_t sl = (uint32_t)rev_read(&b, fs->ll.log);
		uint32_t so = (uint32_t)rev_read(&b, fs->of.log);
		uint32_t sm = (uint32_t)rev_read(&b, fs->ml.log);
		if (b.pos < 0)
			return ZERR;
		for (uint32_t i = 0; i < nseq; i++) {
			const fse_cell cl = fs->ll.cell[sl], co = fs->of.cell[so], cm = fs->ml.cell[sm];
			if (co.sym > 31)
				return ZERR;
			const uint32_t ofv = (1u << co.sym) + (uint32_t)rev_read(&b, co.sym);
			const uint32_t ml = ML_BASE[cm.sym] + (uint32_t)rev_read(&b, ML_BITS[cm.sym]);
			const uint32_t ll = LL_BASE[cl.sym] + (uint32_t)rev_read(&b, LL_BITS[cl.sym]);
			if (i + 1 < nseq) {
				sl = cl.base + (uint32_t)rev_read(&b, cl.nb);
				sm = cm.base + (uint32_t)rev_read(&b, cm.nb);
				so = co.base + (uint32_t)rev_read(&b, co.nb);
			}
			if (b.pos < 0)
				return ZERR;
			/* repeat offsets (RFC 8878 3.1.1.5) */
			uint32_t off;
			if (ofv > 3) {
				off = ofv - 3;
				fs->rep[2] = fs->rep[1];
				fs->rep[1] = fs->rep[0];
				fs->rep[0] = off;
			} else {
				const uint32_t idx = ofv - 1 + (ll == 0);
				if (idx == 0) {
					off = fs->rep[0];
				} else {
					off = idx == 3 ? fs->rep[0] - 1 : fs->rep[idx];
					if (off == 0)
						return ZERR;
					if (idx > 1)
						fs->rep[2] = fs->rep[1];
					fs->rep[1] = fs->rep[0];
					fs->rep[0] = off;
				}
			}
			if (ll > regen - lpos || opos + ll + ml > cap || opos + ll + ml - bstart > block_max)
				return ZERR;
			memcpy(dst + opos, lit + lpos, ll);
			opos += ll;
			lpos += ll;
			if (off > opos)
				return ZERR;
			for (uint32_t k = 0; k < ml; k++) /* byte-wise: overlapping matches replicate */
				dst[opos + k] = dst[opos + k - off];
			opos += ml;
		}
		if (b.pos != 0)
			return ZERR;
	} else if (p != end) {
		return ZERR;
	}
	/* literals left over after the last sequence */
	if (opos + (regen - lpos) > cap || opos + (regen - lpos) - bstart > block_max)
		return ZERR;
	memcpy(dst + opos, lit + lpos, regen - lpos);
	return opos + (regen - lpos);
}

/* ------------------------------------------------------------------ frames */
typedef struct {
	size_t hdr_len;
	uint64_t window, content; /* content = ZO_ZSTD_UNKNOWN when absent */
	int has_checksum, single;
} zhdr;

static int frame_header(const uint8_t *src, size_t n, zhdr *h)
{
	if (n < 6 || rd32(src) != ZO_ZSTD_MAGIC)
		return -1;
	const int fhd = src[4], fcs = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
	if (fhd & 0x08)
		return -1; /* reserved bit */
	static const int did_len[4] = {0, 1, 2, 4}, fcs_len[4] = {0, 2, 4, 8};
	size_t pos = 5;
	const size_t need = 5 + (single ? 0 : 1) + (size_t)did_len[did] + (size_t)(fcs ? fcs_len[fcs] : single);
	if (n < need)
		return -1;
	h->window = 0;
	if (!single) {
		const int wd = src[pos++], e = wd >> 3, m = wd & 7;
		const uint64_t base = 1ull << (10 + e);
		h->window = base + (base >> 3) * (uint64_t)m;
	}
	if (did) {
		uint32_t id = 0;
		for (int i = 0; i < did_len[did]; i++)
			id |= (uint32_t)src[pos + i] << (8 * i);
		if (id)
			return -1; /* dictionaries are not supported */
		pos += (size_t)did_len[did];
	}
	h->content = ZO_ZSTD_UNKNOWN;
	if (fcs == 0 && single)
		h->content = src[pos++];
	else if (fcs == 1) {
		h->content = rd16(src + pos) + 256;
		pos += 2;
	} else if (fcs == 2) {
		h->content = rd32(src + pos);
		pos += 4;
	} else if (fcs == 3) {
		h->content = rd64(src + pos);
		pos += 8;
	}
	if (single)
		h->window = h->content;
	h->has_checksum = (fhd >> 2) & 1;
	h->single = single;
	h->hdr_len = pos;
	return 0;
}

uint64_t zo_zstd_frame_content_size(const uint8_t *frame, size_t slen)
{
	zhdr h;
	if (frame_header(frame, slen, &h))
		return ZO_ZSTD_ERROR;
	return h.content;
}

size_t zo_zstd_decompress_frame(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap,
				size_t *consumed)
{
	zhdr h;
	if (frame_header(src, slen, &h))
		return ZERR;
	const size_t block_max = h.window < 131072 ? (size_t)h.window : 131072;
	frame_state *fs = (frame_state *)calloc(1, sizeof(*fs));
	if (!fs)
		return ZERR;
	fs->lit = (uint8_t *)malloc(131072 + 64);
	fs->rep[0] = 1;
	fs->rep[1] = 4;
	fs->rep[2] = 8;
	size_t ip = h.hdr_len, opos = 0, ret = ZERR;
	for (;;) {
		if (slen - ip < 3)
			goto out;
		const uint32_t bh = rd24(src + ip);
		const int last = bh & 1, type = (bh >> 1) & 3;
		const size_t bsize = bh >> 3;
		ip += 3;
		if (type == 0) {
			if (bsize > block_max || slen - ip < bsize || cap - opos < bsize)
				goto out;
			memcpy(dst + opos, src + ip, bsize);
			opos += bsize;
			ip += bsize;
		} else if (type == 1) {
			if (bsize > block_max || slen - ip < 1 || cap - opos < bsize)
				goto out;
			memset(dst + opos, src[ip], bsize);
			opos += bsize;
			ip += 1;
		} else if (type == 2) {
			if (bsize > block_max || slen - ip < bsize)
				goto out;
			opos = block_decode(fs, src + ip, bsize, dst, opos, cap, block_max);
			if (opos == ZERR)
				goto out;
			ip += bsize;
		} else {
			goto out;
		}
		if (last)
			break;
	}
	if (h.content != ZO_ZSTD_UNKNOWN && h.content != opos)
		goto out;
	if (h.has_checksum) {
		if (slen - ip < 4 || rd32(src + ip) != (uint32_t)zo_xxh64(dst, opos, 0))
			goto out;
		ip += 4;
	}
	if (consumed)
		*consumed = ip;
	ret = opos;
out:
	free(fs->lit);
	free(fs);
	return ret;
}

/* Whole MT stream as lib/zstd-mt_compress.c writes it: records of a 12-byte skippable header
 * (magic, 4, csize) + exactly one zstd frame of csize bytes (lib/zstd-mt_decompress.c:300-369). */
size_t zo_zstdmt_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap)
{
	size_t ip = 0, op = 0;
	if (slen < 16 || rd32(src) != ZO_SKIP_MAGIC || rd32(src + 12) != ZO_ZSTD_MAGIC)
		return ZERR; /* only the layout the compressor emits; legacy layouts: SURVEY 8f-2 */
	while (ip < slen) {
		if (slen - ip < 12 || rd32(src + ip) != ZO_SKIP_MAGIC || rd32(src + ip + 4) != 4)
			return ZERR;
		const size_t csize = rd32(src + ip + 8);
		ip += 12;
		if (slen - ip < csize)
			return ZERR;
		size_t used = 0;
		const size_t got = zo_zstd_decompress_frame(src + ip, csize, dst + op, cap - op, &used);
		if (got == ZERR || used != csize)
			return ZERR;
		op += got;
		ip += csize;
	}
	return op;
}
