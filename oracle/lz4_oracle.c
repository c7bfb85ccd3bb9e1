/*
 * lz4_oracle.c -- scalar CPU restatement of the lz4-mt hot path.  TEST INFRASTRUCTURE ONLY
 * (see zmt_oracle.h for the rules and for what is being restated).
 *
 * Follows:
 *   record framing ........ /root/reference/lib/lz4-mt_compress.c:279-298, lz4-mt_decompress.c:192-388
 *   frame container ....... LZ4 frame format spec v1.6.x as emitted by LZ4F_compressFrame with the
 *                           prefs of lz4-mt_compress.c:141-146
 *   block encoder ......... LZ4 "fast" greedy parser, acceleration 1
 *   block decoder ......... LZ4 block format spec
 *   XXH32 ................. xxHash spec, 32-bit variant
 */
#include "zmt_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------- little-endian helpers (reference equivalents: lib/memmt.h:230-292) ---------- */
static inline uint32_t rd32(const uint8_t *p)
{
	return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
static inline uint64_t rd64(const uint8_t *p)
{
	return (uint64_t)rd32(p) | (uint64_t)rd32(p + 4) << 32;
}
static inline void wr16(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v;
	p[1] = (uint8_t)(v >> 8);
}
static inline void wr32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v;
	p[1] = (uint8_t)(v >> 8);
	p[2] = (uint8_t)(v >> 16);
	p[3] = (uint8_t)(v >> 24);
}
static inline void wr64(uint8_t *p, uint64_t v)
{
	wr32(p, (uint32_t)v);
	wr32(p + 4, (uint32_t)(v >> 32));
}

/* ------------------------------------ XXH32 ------------------------------------ */
#define XP1 2654435761u
#define XP2 2246822519u
#define XP3 3266489917u
#define XP4 668265263u
#define XP5 374761393u
static inline uint32_t rotl(uint32_t x, int r)
{
	return (x << r) | (x >> (32 - r));
}
static inline uint32_t xround(uint32_t acc, uint32_t in)
{
	return rotl(acc + in * XP2, 13) * XP1;
}

uint32_t zo_xxh32(const void *data, size_t len, uint32_t seed)
{
	const uint8_t *p = (const uint8_t *)data, *end = p + len;
	uint32_t h;

	if (len >= 16) {
		uint32_t a = seed + XP1 + XP2, b = seed + XP2, c = seed, d = seed - XP1;
		const uint8_t *lim = end - 16;
		do {
			a = xround(a, rd32(p));
			b = xround(b, rd32(p + 4));
			c = xround(c, rd32(p + 8));
			d = xround(d, rd32(p + 12));
			p += 16;
		} while (p <= lim);
		h = rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18);
	} else {
		h = seed + XP5;
	}
	h += (uint32_t)len;
	while (p + 4 <= end) {
		h = rotl(h + rd32(p) * XP3, 17) * XP4;
		p += 4;
	}
	while (p < end) {
		h = rotl(h + (uint32_t)*p * XP5, 11) * XP1;
		p++;
	}
	h ^= h >> 15;
	h *= XP2;
	h ^= h >> 13;
	h *= XP3;
	h ^= h >> 16;
	return h;
}

/* ------------------------------- LZ4 block encoder ------------------------------- */
enum { MINMATCH = 4, MFLIMIT = 12, LASTLITERALS = 5, DIST_MAX = 65535, SKIP_TRIGGER = 6 };

static inline uint32_t hash_u32tab(const uint8_t *p) /* byU32: 5-byte hash, 12 bits */
{
	return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> 52);
}
static inline uint32_t hash_u16tab(const uint8_t *p) /* byU16: 4-byte hash, 13 bits */
{
	return (rd32(p) * 2654435761u) >> 19;
}

static inline uint32_t tab_get(const zo_lz4_table *t, uint32_t h, int u16)
{
	return u16 ? ((const uint16_t *)t->tab)[h] : t->tab[h];
}
static inline void tab_put(zo_lz4_table *t, uint32_t h, uint32_t v, int u16)
{
	if (u16)
		((uint16_t *)t->tab)[h] = (uint16_t)v;
	else
		t->tab[h] = v;
}
#define HASH(p) (u16 ? hash_u16tab(p) : hash_u32tab(p))

size_t zo_lz4_block_encode(zo_lz4_table *t, const uint8_t *chunk, size_t pos, size_t len,
			   uint8_t *dst, size_t cap, int u16)
{
	const uint8_t *const src = chunk + pos;
	const uint8_t *const iend = src + len;
	const uint8_t *const mflimit_p1 = iend - MFLIMIT + 1;
	const uint8_t *const matchlimit = iend - LASTLITERALS;
	/* catch-up floor: chunk start for linked blocks (prefix = all earlier blocks), block start
	 * for the single independent block */
	const uint8_t *const low = u16 ? src : chunk;
	const uint8_t *ip = src, *anchor = src;
	uint8_t *op = dst, *const olimit = dst + cap;
	uint32_t fwd_h;

	if (len < MFLIMIT + 1)
		goto last_literals;

	tab_put(t, HASH(ip), (uint32_t)(ip - chunk), u16);
	ip++;
	fwd_h = HASH(ip);

	for (;;) {
		const uint8_t *match;
		uint8_t *token;
		{
			const uint8_t *fwd = ip;
			unsigned step = 1, nb = 1u << SKIP_TRIGGER;
			for (;;) {
				uint32_t h = fwd_h;
				uint32_t cur = (uint32_t)(fwd - chunk);
				uint32_t midx = tab_get(t, h, u16);
				ip = fwd;
				fwd += step;
				step = nb++ >> SKIP_TRIGGER;
				if (fwd > mflimit_p1)
					goto last_literals;
				match = chunk + midx;
				fwd_h = HASH(fwd);
				tab_put(t, h, cur, u16);
				if (!u16 && midx + DIST_MAX < cur)
					continue;
				if (rd32(match) == rd32(ip))
					break;
			}
		}
		while (ip > anchor && match > low && ip[-1] == match[-1]) {
			ip--;
			match--;
		}
		{
			size_t lit = (size_t)(ip - anchor);
			token = op++;
			if (op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > olimit)
				return 0;
			if (lit >= 15) {
				size_t r = lit - 15;
				*token = 15 << 4;
				for (; r >= 255; r -= 255)
					*op++ = 255;
				*op++ = (uint8_t)r;
			} else {
				*token = (uint8_t)(lit << 4);
			}
			memcpy(op, anchor, lit);
			op += lit;
		}
next_match:
		wr16(op, (uint32_t)(ip - match));
		op += 2;
		{
			const uint8_t *a = ip + MINMATCH, *b = match + MINMATCH;
			size_t mc;
			while (a < matchlimit && *a == *b) {
				a++;
				b++;
			}
			mc = (size_t)(a - (ip + MINMATCH));
			ip += mc + MINMATCH;
			if (op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit)
				return 0;
			if (mc >= 15) {
				*token += 15;
				mc -= 15;
				for (; mc >= 255; mc -= 255)
					*op++ = 255;
				*op++ = (uint8_t)mc;
			} else {
				*token += (uint8_t)mc;
			}
		}
		anchor = ip;
		if (ip >= mflimit_p1)
			break;
		tab_put(t, HASH(ip - 2), (uint32_t)(ip - 2 - chunk), u16);
		{
			uint32_t h = HASH(ip);
			uint32_t cur = (uint32_t)(ip - chunk);
			uint32_t midx = tab_get(t, h, u16);
			match = chunk + midx;
			tab_put(t, h, cur, u16);
			if ((u16 || midx + DIST_MAX >= cur) && rd32(match) == rd32(ip)) {
				token = op++;
				*token = 0;
				goto next_match;
			}
		}
		fwd_h = HASH(++ip);
	}

last_literals:
	{
		size_t run = (size_t)(iend - anchor);
		if (op + run + 1 + (run + 255 - 15) / 255 > olimit)
			return 0;
		if (run >= 15) {
			size_t r = run - 15;
			*op++ = 15 << 4;
			for (; r >= 255; r -= 255)
				*op++ = 255;
			*op++ = (uint8_t)r;
		} else {
			*op++ = (uint8_t)(run << 4);
		}
		memcpy(op, anchor, run);
		op += run;
	}
	return (size_t)(op - dst);
}

/* ------------------------------- LZ4 block decoder ------------------------------- */
size_t zo_lz4_block_decode(const uint8_t *src, size_t slen, uint8_t *out, size_t opos,
			   size_t out_limit)
{
	const uint8_t *ip = src, *const iend = src + slen;

	if (slen == 0)
		return (size_t)-1;
	for (;;) {
		unsigned tok;
		size_t lit, ml, off;

		if (ip >= iend)
			return (size_t)-1;
		tok = *ip++;
		lit = tok >> 4;
		if (lit == 15) {
			unsigned b;
			do {
				if (ip >= iend)
					return (size_t)-1;
				b = *ip++;
				lit += b;
			} while (b == 255);
		}
		if ((size_t)(iend - ip) < lit || out_limit - opos < lit)
			return (size_t)-1;
		memcpy(out + opos, ip, lit);
		ip += lit;
		opos += lit;
		if (ip == iend)
			return opos; /* last sequence: literals only */
		if (iend - ip < 2)
			return (size_t)-1;
		off = (size_t)ip[0] | (size_t)ip[1] << 8;
		ip += 2;
		ml = tok & 15;
		if (ml == 15) {
			unsigned b;
			do {
				if (ip >= iend)
					return (size_t)-1;
				b = *ip++;
				ml += b;
			} while (b == 255);
		}
		ml += MINMATCH;
		if (off == 0 || off > opos || out_limit - opos < ml)
			return (size_t)-1;
		{
			size_t i;
			const uint8_t *m = out + opos - off;
			uint8_t *d = out + opos;
			for (i = 0; i < ml; i++) /* byte-serial: offset < length replicates */
				d[i] = m[i];
		}
		opos += ml;
	}
}

/* --------------------------------- LZ4F container --------------------------------- */
size_t zo_lz4f_bound(size_t n)
{
	/* 19-byte max header + 4 per block + data + 4 endmark + 4 content checksum */
	size_t full = n / ZO_BLOCK_MAX, part = n % ZO_BLOCK_MAX;
	size_t nblk = full + (part ? 1 : 0);
	return 19 + 4 * nblk + full * ZO_BLOCK_MAX + part + 8;
}

size_t zo_lz4f_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
	uint8_t *op = dst;
	zo_lz4_table tab;
	size_t pos;
	/* LZ4F_compressFrame flips to independent blocks when the input fits one block. */
	int single = (n <= ZO_BLOCK_MAX);
	uint8_t flg = 0x40 | (single ? 0x20 : 0) | (n ? 0x08 : 0) | 0x04;
	size_t hdr = n ? 15 : 7;

	if (cap < zo_lz4f_bound(n))
		return (size_t)-1;
	wr32(op, ZO_LZ4F_MAGIC);
	op[4] = flg;
	op[5] = 0x40; /* BD: 64 KiB blocks */
	if (n)
		wr64(op + 6, (uint64_t)n);
	op[hdr - 1] = (uint8_t)(zo_xxh32(op + 4, hdr - 5, 0) >> 8);
	op += hdr;

	memset(&tab, 0, sizeof tab);
	for (pos = 0; pos < n; pos += ZO_BLOCK_MAX) {
		size_t len = n - pos < ZO_BLOCK_MAX ? n - pos : ZO_BLOCK_MAX;
		size_t c = zo_lz4_block_encode(&tab, src, pos, len, op + 4, len - 1, single);
		if (c == 0) {
			wr32(op, (uint32_t)len | 0x80000000u);
			memcpy(op + 4, src + pos, len);
			c = len;
		} else {
			wr32(op, (uint32_t)c);
		}
		op += 4 + c;
	}
	wr32(op, 0);
	wr32(op + 4, zo_xxh32(src, n, 0));
	op += 8;
	return (size_t)(op - dst);
}

uint64_t zo_lz4f_content_size(const uint8_t *frame, size_t slen)
{
	if (slen < 14)
		return 0;
	return rd64(frame + 6);
}

size_t zo_lz4f_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap)
{
	const uint8_t *ip = src, *const iend = src + slen;
	uint8_t flg, bd;
	size_t hdr, opos = 0, blkmax;
	uint64_t csize = 0;
	int has_csize, has_ccheck, has_bcheck, has_dict, indep;

	if (slen < 7 || rd32(ip) != ZO_LZ4F_MAGIC)
		return (size_t)-1;
	flg = ip[4];
	bd = ip[5];
	if ((flg >> 6) != 1 || (flg & 0x02) || (bd & 0x8F))
		return (size_t)-1;
	indep = (flg >> 5) & 1;
	has_bcheck = (flg >> 4) & 1;
	has_csize = (flg >> 3) & 1;
	has_ccheck = (flg >> 2) & 1;
	has_dict = flg & 1;
	if ((bd >> 4) < 4)
		return (size_t)-1;
	blkmax = (size_t)1 << (8 + 2 * (bd >> 4));
	hdr = 7 + (has_csize ? 8 : 0) + (has_dict ? 4 : 0);
	if (slen < hdr)
		return (size_t)-1;
	if (ip[hdr - 1] != (uint8_t)(zo_xxh32(ip + 4, hdr - 5, 0) >> 8))
		return (size_t)-1;
	if (has_csize)
		csize = rd64(ip + 6);
	ip += hdr;

	for (;;) {
		uint32_t bh;
		size_t bsz;
		if (iend - ip < 4)
			return (size_t)-1;
		bh = rd32(ip);
		ip += 4;
		if (bh == 0)
			break;
		bsz = bh & 0x7FFFFFFFu;
		if (bsz > blkmax || (size_t)(iend - ip) < bsz + (has_bcheck ? 4 : 0))
			return (size_t)-1;
		if (bh & 0x80000000u) {
			if (cap - opos < bsz)
				return (size_t)-1;
			memcpy(dst + opos, ip, bsz);
			opos += bsz;
		} else {
			/* independent blocks may not reference earlier blocks */
			size_t base = indep ? opos : 0;
			size_t room = cap - opos < blkmax ? cap - opos : blkmax;
			size_t np = zo_lz4_block_decode(ip, bsz, dst + base, opos - base,
							opos - base + room);
			if (np == (size_t)-1)
				return (size_t)-1;
			opos = np + base;
		}
		if (has_bcheck && rd32(ip + bsz) != zo_xxh32(ip, bsz, 0))
			return (size_t)-1;
		ip += bsz + (has_bcheck ? 4 : 0);
	}
	if (has_csize && csize != (uint64_t)opos)
		return (size_t)-1;
	if (has_ccheck) {
		if (iend - ip < 4 || rd32(ip) != zo_xxh32(dst, opos, 0))
			return (size_t)-1;
		ip += 4;
	}
	if (ip != iend) /* lz4-mt_decompress.c:359-362: the record must be exactly one frame */
		return (size_t)-1;
	return opos;
}

/* ----------------------------------- MT stream ----------------------------------- */
size_t zo_lz4mt_compress_bound(size_t n, size_t chunk)
{
	size_t full = n / chunk, part = n % chunk;
	size_t b = full * (zo_lz4f_bound(chunk) + 12);
	if (part || n == 0)
		b += zo_lz4f_bound(part) + 12;
	return b;
}

size_t zo_lz4mt_compress(const uint8_t *src, size_t n, size_t chunk, uint8_t *dst, size_t cap)
{
	size_t pos = 0, out = 0;
	int first = 1;

	/* pt_compress: stop at the first zero-length read unless no frame was written yet */
	while (pos < n || first) {
		size_t len = n - pos < chunk ? n - pos : chunk;
		size_t c;
		if (cap - out < zo_lz4f_bound(len) + 12)
			return (size_t)-1;
		c = zo_lz4f_compress(src + pos, len, dst + out + 12, cap - out - 12);
		wr32(dst + out, ZO_SKIP_MAGIC);
		wr32(dst + out + 4, 4);
		wr32(dst + out + 8, (uint32_t)c);
		out += 12 + c;
		pos += len;
		first = 0;
	}
	return out;
}

size_t zo_lz4mt_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap)
{
	size_t ip = 0, out = 0;

	while (ip < slen) {
		uint32_t csz;
		size_t want, got;
		if (slen - ip < 12 || rd32(src + ip) != ZO_SKIP_MAGIC || rd32(src + ip + 4) != 4)
			return (size_t)-1;
		csz = rd32(src + ip + 8);
		ip += 12;
		if (slen - ip < csz)
			return (size_t)-1;
		/* lz4-mt_decompress.c:329-334: output size comes from the content-size field;
		 * a frame without one (only the empty frame is emitted that way) gets 64 KiB */
		if (csz >= 15 && (src[ip + 4] & 0x08)) {
			uint64_t cs = zo_lz4f_content_size(src + ip, csz);
			if (cs > (uint64_t)(cap - out))
				return (size_t)-2;
			want = (size_t)cs;
		} else {
			want = cap - out < 65536 ? cap - out : 65536;
		}
		got = zo_lz4f_decompress(src + ip, csz, dst + out, want);
		if (got == (size_t)-1)
			return (size_t)-1;
		out += got;
		ip += csz;
	}
	return out;
}

/* ------------------------- T-thread variants (CPU baseline) ------------------------- */
struct mtjob {
	const uint8_t *src;
	uint8_t *dst;
	size_t n, chunk, nchunks;
	size_t *csz;      /* per-chunk record size */
	size_t slot;      /* per-chunk slot stride in dst scratch */
	uint8_t *scratch;
	size_t next;
	pthread_mutex_t mu;
	int fail;
	/* decompress */
	const size_t *roff; /* record offsets */
	const size_t *ooff; /* output offsets */
};

static void *c_worker(void *arg)
{
	struct mtjob *j = (struct mtjob *)arg;
	for (;;) {
		size_t i, pos, len, c;
		uint8_t *o;
		pthread_mutex_lock(&j->mu);
		i = j->next++;
		pthread_mutex_unlock(&j->mu);
		if (i >= j->nchunks)
			break;
		pos = i * j->chunk;
		len = j->n - pos < j->chunk ? j->n - pos : j->chunk;
		o = j->scratch + i * j->slot;
		c = zo_lz4f_compress(j->src + pos, len, o + 12, j->slot - 12);
		wr32(o, ZO_SKIP_MAGIC);
		wr32(o + 4, 4);
		wr32(o + 8, (uint32_t)c);
		j->csz[i] = c + 12;
	}
	return NULL;
}

size_t zo_lz4mt_compress_mt(const uint8_t *src, size_t n, size_t chunk, uint8_t *dst, size_t cap,
			    int threads)
{
	struct mtjob j;
	pthread_t th[256];
	size_t i, out = 0;
	int t;

	memset(&j, 0, sizeof j);
	j.src = src;
	j.n = n;
	j.chunk = chunk;
	j.nchunks = n ? (n + chunk - 1) / chunk : 1;
	j.slot = zo_lz4f_bound(chunk) + 12;
	j.scratch = (uint8_t *)malloc(j.nchunks * j.slot);
	j.csz = (size_t *)calloc(j.nchunks, sizeof(size_t));
	if (!j.scratch || !j.csz)
		return (size_t)-1;
	pthread_mutex_init(&j.mu, NULL);
	if (threads > 256)
		threads = 256;
	for (t = 0; t < threads; t++)
		pthread_create(&th[t], NULL, c_worker, &j);
	for (t = 0; t < threads; t++)
		pthread_join(th[t], NULL);
	for (i = 0; i < j.nchunks; i++) {
		if (cap - out < j.csz[i]) {
			out = (size_t)-1;
			break;
		}
		memcpy(dst + out, j.scratch + i * j.slot, j.csz[i]);
		out += j.csz[i];
	}
	free(j.scratch);
	free(j.csz);
	return out;
}

static void *d_worker(void *arg)
{
	struct mtjob *j = (struct mtjob *)arg;
	for (;;) {
		size_t i, got, want;
		pthread_mutex_lock(&j->mu);
		i = j->next++;
		pthread_mutex_unlock(&j->mu);
		if (i >= j->nchunks)
			break;
		want = j->ooff[i + 1] - j->ooff[i];
		got = zo_lz4f_decompress(j->src + j->roff[i] + 12, j->roff[i + 1] - j->roff[i] - 12,
					 j->dst + j->ooff[i], want);
		if (got != want)
			j->fail = 1;
	}
	return NULL;
}

size_t zo_lz4mt_decompress_mt(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap, int threads)
{
	struct mtjob j;
	pthread_t th[256];
	size_t ip = 0, out = 0, nrec = 0, capn = 1024;
	size_t *roff = (size_t *)malloc(capn * sizeof(size_t));
	size_t *ooff = (size_t *)malloc(capn * sizeof(size_t));
	int t;

	while (ip < slen) { /* serial header walk = pt_read under the read mutex */
		uint32_t csz;
		if (slen - ip < 12 || rd32(src + ip) != ZO_SKIP_MAGIC || rd32(src + ip + 4) != 4)
			goto bad;
		csz = rd32(src + ip + 8);
		if (slen - ip - 12 < csz)
			goto bad;
		if (nrec + 2 > capn) {
			capn *= 2;
			roff = (size_t *)realloc(roff, capn * sizeof(size_t));
			ooff = (size_t *)realloc(ooff, capn * sizeof(size_t));
		}
		roff[nrec] = ip;
		ooff[nrec] = out;
		if (csz >= 15 && (src[ip + 12 + 4] & 0x08))
			out += (size_t)zo_lz4f_content_size(src + ip + 12, csz);
		if (out > cap)
			goto bad;
		ip += 12 + csz;
		nrec++;
	}
	roff[nrec] = ip;
	ooff[nrec] = out;
	memset(&j, 0, sizeof j);
	j.src = src;
	j.dst = dst;
	j.nchunks = nrec;
	j.roff = roff;
	j.ooff = ooff;
	pthread_mutex_init(&j.mu, NULL);
	if (threads > 256)
		threads = 256;
	for (t = 0; t < threads; t++)
		pthread_create(&th[t], NULL, d_worker, &j);
	for (t = 0; t < threads; t++)
		pthread_join(th[t], NULL);
	free(roff);
	free(ooff);
	return j.fail ? (size_t)-1 : out;
bad:
	free(roff);
	free(ooff);
	return (size_t)-1;
}
