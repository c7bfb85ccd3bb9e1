/*
 * lz4hc_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the LZ4 "HC" block encoder the
 * reference reaches for levels 3..12 (LZ4F_compressFrame with compressionLevel >= 3, call site
 * /root/reference/lib/lz4-mt_compress.c:141-146 + :281; the CLI's default level is 3,
 * /root/reference/programs/lz4-mt.c:19).
 *
 * The algorithm lives in liblz4's lz4hc.c, a third-party dependency that is NOT in /root/reference
 * (programs/Makefile:9 pins v1.9.4; this image carries the binary of v1.9.3 only).  What follows
 * restates that file's published parsers for every HC level: 3..8 the hash-chain walk (4, 8, 16, 32, 64,
 * 128 attempts), 9 the same with the repeated-pattern analysis (256 attempts), 10..12 the optimal
 * parser (price table, chain swap; below) -- and is pinned byte for byte against the reference build
 * (oracle/_ref/liblz4mt_ref.so, tests/test_oracle_vs_ref.py) and the committed fixtures
 * (tests/golden/lz4hc).  Frame container: lz4_oracle.c.
 */
#include <stdlib.h>
#include <string.h>

#include "zmt_oracle.h"

#define HC_HASH_LOG 15
#define HC_MAXD 65536u
#define HC_DIST_MAX 65535u
#define HC_MINMATCH 4
#define HC_MFLIMIT 12
#define HC_LASTLITERALS 5
#define HC_OPTIMAL_ML 18 /* (ML_MASK - 1) + MINMATCH */
#define HC_BASE 65536u   /* index of the chunk's first byte (LZ4HC_init_internal starts at 64 KB) */

typedef struct {
	uint32_t hash[1u << HC_HASH_LOG];
	uint16_t chain[HC_MAXD];
	const uint8_t *src;     /* chunk start: byte at index i is src[i - HC_BASE] */
	uint32_t next_to_update;
} zo_hc;

static uint32_t rd32(const uint8_t *p)
{
	return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
static uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
static uint32_t hc_hash(const uint8_t *p) { return (rd32(p) * 2654435761u) >> (32 - HC_HASH_LOG); }

static unsigned hc_count(const uint8_t *a, const uint8_t *b, const uint8_t *alimit)
{
	const uint8_t *s = a;
	while (a < alimit && *a == *b) {
		a++;
		b++;
	}
	return (unsigned)(a - s);
}

/* LZ4HC_Insert: every position below `target` enters the hash chains */
static void hc_insert(zo_hc *h, uint32_t target)
{
	uint32_t idx = h->next_to_update;
	while (idx < target) {
		const uint32_t hv = hc_hash(h->src + (idx - HC_BASE));
		uint32_t delta = idx - h->hash[hv];
		if (delta > HC_DIST_MAX)
			delta = HC_DIST_MAX;
		h->chain[idx & (HC_MAXD - 1)] = (uint16_t)delta;
		h->hash[hv] = idx;
		idx++;
	}
	h->next_to_update = target;
}

/* LZ4HC_countPattern / LZ4HC_reverseCountPattern: bytes after p (before p) that continue the
 * repetition of the 4-byte pattern, phase 0 at p */
static unsigned hc_count_pattern(const uint8_t *p, const uint8_t *end, uint32_t pattern)
{
	const uint8_t *s = p;
	unsigned k = 0;
	while (p < end && *p == (uint8_t)(pattern >> (8 * (k & 3)))) {
		p++;
		k++;
	}
	return (unsigned)(p - s);
}
static unsigned hc_reverse_count_pattern(const uint8_t *p, const uint8_t *low, uint32_t pattern)
{
	const uint8_t *s = p;
	unsigned k = 3;
	while (p > low && p[-1] == (uint8_t)(pattern >> (8 * (k & 3)))) {
		p--;
		k--;
	}
	return (unsigned)(s - p);
}

/* LZ4HC_InsertAndGetWiderMatch without chain swap (levels 3..9); `pattern_analysis` is the
 * repeated-pattern shortcut liblz4 enables for more than 128 attempts (level 9).
 * ip, low, high are positions in the chunk; returns the longest length found (> longest on entry),
 * match position in *mpos and the (possibly moved back) start in *spos. */
static int hc_wider_core(zo_hc *h, uint32_t ip, uint32_t low_limit, uint32_t high_limit, int longest,
			 uint32_t *mpos, uint32_t *spos, int max_attempts, int pattern_analysis, int chain_swap)
{
	uint32_t match_chain_pos = 0;
	int repeat = 0; /* 0 untested, 1 not a repetition, 2 confirmed */
	size_t src_pattern_len = 0;
	const uint8_t *const s = h->src;
	const uint32_t ip_index = ip + HC_BASE;
	const uint32_t lowest = (HC_BASE + HC_DIST_MAX + 1 > ip_index) ? HC_BASE : ip_index - HC_DIST_MAX;
	const int look_back = (int)(ip - low_limit);
	const uint32_t pattern = rd32(s + ip);
	int attempts = max_attempts;
	uint32_t match_index;

	hc_insert(h, ip_index);
	match_index = h->hash[hc_hash(s + ip)];
	while (match_index >= lowest && attempts > 0) {
		const uint32_t m = match_index - HC_BASE; /* chunk position of the candidate */
		int match_len = 0;
		attempts--;
		if (rd16(s + low_limit + longest - 1) == rd16(s + m - look_back + longest - 1)) {
			if (rd32(s + m) == pattern) {
				int back = 0, ml;
				if (look_back) {
					/* LZ4HC_countBack: not before the search start, not before the chunk */
					const int min_i = (int)low_limit - (int)ip;
					const int min_m = -(int)m;
					const int min = min_i > min_m ? min_i : min_m;
					while (back > min && s[ip + back - 1] == s[m + back - 1])
						back--;
				}
				ml = HC_MINMATCH + (int)hc_count(s + ip + HC_MINMATCH, s + m + HC_MINMATCH, s + high_limit);
				ml -= back;
				match_len = ml;
				if (ml > longest) {
					longest = ml;
					*mpos = (uint32_t)((int)m + back);
					*spos = (uint32_t)((int)ip + back);
				}
			}
		}
		if (chain_swap && match_len == longest) {
			/* a match as long as the best one: continue on the chain of the position inside it whose
			 * next candidate is farthest away (the optimal parser only; search forward only) */
			if (match_index + (uint32_t)longest <= ip_index) {
				const int k_trigger = 4;
				uint32_t dist_to_next = 1;
				const int end = longest - HC_MINMATCH + 1;
				int step = 1, accel = 1 << k_trigger, pos;
				for (pos = 0; pos < end; pos += step) {
					const uint32_t cand_dist = h->chain[(match_index + (uint32_t)pos) & (HC_MAXD - 1)];
					step = (accel++ >> k_trigger);
					if (cand_dist > dist_to_next) {
						dist_to_next = cand_dist;
						match_chain_pos = (uint32_t)pos;
						accel = 1 << k_trigger;
					}
				}
				if (dist_to_next > 1) {
					if (dist_to_next > match_index)
						break;
					match_index -= dist_to_next;
					continue;
				}
			}
		}
		if (pattern_analysis && match_chain_pos == 0 && h->chain[match_index & (HC_MAXD - 1)] == 1) {
			/* the candidate sits inside a run of one byte value: jump to where the run can
			 * match the run at ip in full instead of walking it link by link */
			const uint32_t cand = match_index - 1;
			if (repeat == 0) {
				if (((pattern & 0xFFFF) == (pattern >> 16)) & ((pattern & 0xFF) == (pattern >> 24))) {
					repeat = 2;
					src_pattern_len = hc_count_pattern(s + ip + 4, s + high_limit, pattern) + 4;
				} else {
					repeat = 1;
				}
			}
			if (repeat == 2 && cand >= lowest) {
				const uint8_t *mp = s + (cand - HC_BASE);
				if (rd32(mp) == pattern) {
					const size_t fwd = hc_count_pattern(mp + 4, s + high_limit, pattern) + 4;
					size_t back = hc_reverse_count_pattern(mp, s, pattern), cur;
					{
						const uint32_t far = cand - (uint32_t)back;
						back = cand - (far > lowest ? far : lowest);
					}
					cur = back + fwd;
					if (cur >= src_pattern_len && fwd <= src_pattern_len) {
						match_index = cand + (uint32_t)fwd - (uint32_t)src_pattern_len;
					} else {
						match_index = cand - (uint32_t)back;
						if (look_back == 0) {
							const size_t max_ml = cur < src_pattern_len ? cur : src_pattern_len;
							if ((size_t)longest < max_ml) {
								if (ip_index - match_index > HC_DIST_MAX)
									break;
								longest = (int)max_ml;
								*mpos = match_index - HC_BASE;
								*spos = ip;
							}
							{
								const uint32_t d = h->chain[match_index & (HC_MAXD - 1)];
								if (d > match_index)
									break;
								match_index -= d;
							}
						}
					}
					continue;
				}
			}
		}
		match_index -= h->chain[(match_index + match_chain_pos) & (HC_MAXD - 1)];
	}
	return longest;
}
/* the hash-chain levels: pattern analysis beyond 128 attempts (level 9), never a chain swap */
static int hc_wider(zo_hc *h, uint32_t ip, uint32_t low_limit, uint32_t high_limit, int longest,
		    uint32_t *mpos, uint32_t *spos, int max_attempts)
{
	return hc_wider_core(h, ip, low_limit, high_limit, longest, mpos, spos, max_attempts, max_attempts > 128, 0);
}

/* LZ4HC_encodeSequence; returns 1 when the output limit is hit */
static int hc_encode(const uint8_t *s, uint32_t *ip, uint8_t **op, uint32_t *anchor, int ml, uint32_t mpos,
		     uint8_t *oend)
{
	uint8_t *const token = (*op)++;
	size_t length = (size_t)(*ip - *anchor);
	if (*op + (length / 255) + length + (2 + 1 + HC_LASTLITERALS) > oend)
		return 1;
	if (length >= 15) {
		size_t len = length - 15;
		*token = 15 << 4;
		for (; len >= 255; len -= 255)
			*(*op)++ = 255;
		*(*op)++ = (uint8_t)len;
	} else {
		*token = (uint8_t)(length << 4);
	}
	memcpy(*op, s + *anchor, length);
	*op += length;
	(*op)[0] = (uint8_t)(*ip - mpos);
	(*op)[1] = (uint8_t)((*ip - mpos) >> 8);
	*op += 2;
	length = (size_t)ml - HC_MINMATCH;
	if (*op + (length / 255) + (1 + HC_LASTLITERALS) > oend)
		return 1;
	if (length >= 15) {
		*token += 15;
		length -= 15;
		for (; length >= 510; length -= 510) {
			*(*op)++ = 255;
			*(*op)++ = 255;
		}
		if (length >= 255) {
			length -= 255;
			*(*op)++ = 255;
		}
		*(*op)++ = (uint8_t)length;
	} else {
		*token += (uint8_t)length;
	}
	*ip += (uint32_t)ml;
	*anchor = *ip;
	return 0;
}

/* LZ4HC_compress_hashChain over chunk positions [start, start + n), output capacity cap
 * (limitedOutput): compressed size or 0 */
static size_t hc_block(zo_hc *h, uint32_t start, uint32_t n, uint8_t *dst, size_t cap, int max_attempts)
{
	const uint8_t *const s = h->src;
	uint32_t ip = start, anchor = start;
	const uint32_t iend = start + n;
	const uint32_t mflimit = iend - HC_MFLIMIT, matchlimit = iend - HC_LASTLITERALS;
	uint8_t *op = dst, *const oend = dst + cap;
	int ml0, ml, ml2, ml3;
	uint32_t start0, ref0, ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0;

	if (n < HC_MFLIMIT + 1)
		goto last_literals;
	while (ip <= mflimit) {
		{
			uint32_t dummy = ip;
			ml = hc_wider(h, ip, ip, matchlimit, HC_MINMATCH - 1, &ref, &dummy, max_attempts);
		}
		if (ml < HC_MINMATCH) {
			ip++;
			continue;
		}
		start0 = ip;
		ref0 = ref;
		ml0 = ml;
search2:
		if (ip + (uint32_t)ml <= mflimit)
			ml2 = hc_wider(h, ip + (uint32_t)ml - 2, ip + 0, matchlimit, ml, &ref2, &start2, max_attempts);
		else
			ml2 = ml;
		if (ml2 == ml) { /* no better match: encode ML1 */
			if (hc_encode(s, &ip, &op, &anchor, ml, ref, oend))
				return 0;
			continue;
		}
		if (start0 < ip) { /* first match was skipped at least once */
			if (start2 < ip + (uint32_t)ml0) { /* squeezing ML1 between ML0 (original ML1) and ML2 */
				ip = start0;
				ref = ref0;
				ml = ml0;
			}
		}
		if (start2 - ip < 3) { /* first match too small: removed */
			ml = ml2;
			ip = start2;
			ref = ref2;
			goto search2;
		}
search3:
		if (start2 - ip < HC_OPTIMAL_ML) {
			int correction;
			int new_ml = ml;
			if (new_ml > HC_OPTIMAL_ML)
				new_ml = HC_OPTIMAL_ML;
			if (ip + (uint32_t)new_ml > start2 + (uint32_t)ml2 - HC_MINMATCH)
				new_ml = (int)(start2 - ip) + ml2 - HC_MINMATCH;
			correction = new_ml - (int)(start2 - ip);
			if (correction > 0) {
				start2 += (uint32_t)correction;
				ref2 += (uint32_t)correction;
				ml2 -= correction;
			}
		}
		if (start2 + (uint32_t)ml2 <= mflimit)
			ml3 = hc_wider(h, start2 + (uint32_t)ml2 - 3, start2, matchlimit, ml2, &ref3, &start3, max_attempts);
		else
			ml3 = ml2;
		if (ml3 == ml2) { /* no better match: encode ML1 and ML2 */
			if (start2 < ip + (uint32_t)ml)
				ml = (int)(start2 - ip);
			if (hc_encode(s, &ip, &op, &anchor, ml, ref, oend))
				return 0;
			ip = start2;
			if (hc_encode(s, &ip, &op, &anchor, ml2, ref2, oend))
				return 0;
			continue;
		}
		if (start3 < ip + (uint32_t)ml + 3) { /* not enough space for match 2: remove it */
			if (start3 >= ip + (uint32_t)ml) { /* Seq1 can be written now; Seq3 becomes Seq1 */
				if (start2 < ip + (uint32_t)ml) {
					const int correction = (int)(ip + (uint32_t)ml - start2);
					start2 += (uint32_t)correction;
					ref2 += (uint32_t)correction;
					ml2 -= correction;
					if (ml2 < HC_MINMATCH) {
						start2 = start3;
						ref2 = ref3;
						ml2 = ml3;
					}
				}
				if (hc_encode(s, &ip, &op, &anchor, ml, ref, oend))
					return 0;
				ip = start3;
				ref = ref3;
				ml = ml3;
				start0 = start2;
				ref0 = ref2;
				ml0 = ml2;
				goto search2;
			}
			start2 = start3;
			ref2 = ref3;
			ml2 = ml3;
			goto search3;
		}
		/* three ascending matches: write the first one */
		if (start2 < ip + (uint32_t)ml) {
			if (start2 - ip < HC_OPTIMAL_ML) {
				int correction;
				if (ml > HC_OPTIMAL_ML)
					ml = HC_OPTIMAL_ML;
				if (ip + (uint32_t)ml > start2 + (uint32_t)ml2 - HC_MINMATCH)
					ml = (int)(start2 - ip) + ml2 - HC_MINMATCH;
				correction = ml - (int)(start2 - ip);
				if (correction > 0) {
					start2 += (uint32_t)correction;
					ref2 += (uint32_t)correction;
					ml2 -= correction;
				}
			} else {
				ml = (int)(start2 - ip);
			}
		}
		if (hc_encode(s, &ip, &op, &anchor, ml, ref, oend))
			return 0;
		ip = start2;
		ref = ref2;
		ml = ml2;
		start2 = start3;
		ref2 = ref3;
		ml2 = ml3;
		goto search3;
	}
last_literals:
	{
		const size_t run = (size_t)(iend - anchor);
		const size_t lit_len = (run + 255 - 15) / 255;
		if (op + 1 + lit_len + run > oend)
			return 0;
		if (run >= 15) {
			size_t acc = run - 15;
			*op++ = 15 << 4;
			for (; acc >= 255; acc -= 255)
				*op++ = 255;
			*op++ = (uint8_t)acc;
		} else {
			*op++ = (uint8_t)(run << 4);
		}
		memcpy(op, s + anchor, run);
		op += run;
	}
	return (size_t)(op - dst);
}


/* ---- levels 10..12: LZ4HC_compress_optimal ------------------------------------------------
 * Prices are bytes of output.  From a position with a match the parser fills a table of the cheapest
 * way to reach each of the next positions (literals or a match ending there), searching again at
 * every position where that can pay, then walks the table backwards from the last position reached
 * and emits the chosen sequences.  nb_searches / sufficient_len / full_update per level:
 * 96 / 64 / 0, 512 / 128 / 0, 16384 / 4096 / 1. */
#define HC_OPT_NUM 4096
#define HC_TRAILING 3
typedef struct {
	int price, off, mlen, litlen;
} hc_opt_t;

static int hc_lit_price(int litlen)
{
	int price = litlen;
	if (litlen >= 15)
		price += 1 + (litlen - 15) / 255;
	return price;
}
static int hc_seq_price(int litlen, int mlen)
{
	int price = 1 + 2 + hc_lit_price(litlen);
	if (mlen >= 15 + HC_MINMATCH)
		price += 1 + (mlen - (15 + HC_MINMATCH)) / 255;
	return price;
}
/* LZ4HC_FindLongerMatch: pattern analysis and chain swap on, forward only */
static void hc_find_longer(zo_hc *h, uint32_t ip, uint32_t high_limit, int min_len, int nb_searches, int *len, int *off)
{
	uint32_t mpos = 0, spos = ip;
	const int ml = hc_wider_core(h, ip, ip, high_limit, min_len, &mpos, &spos, nb_searches, 1, 1);
	*len = 0;
	*off = 0;
	if (ml <= min_len)
		return;
	*len = ml;
	*off = (int)(ip - mpos);
}

static size_t hc_opt_block(zo_hc *h, uint32_t start, uint32_t n, uint8_t *dst, size_t cap, int nb_searches,
			   size_t sufficient_len, int full_update)
{
	const uint8_t *const s = h->src;
	hc_opt_t *const opt = (hc_opt_t *)malloc(sizeof(hc_opt_t) * (HC_OPT_NUM + HC_TRAILING + 8));
	size_t result = 0;
	uint32_t ip = start, anchor = start;
	const uint32_t iend = start + n;
	const uint32_t mflimit = iend - HC_MFLIMIT, matchlimit = iend - HC_LASTLITERALS;
	uint8_t *op = dst, *const oend = dst + cap;

	if (!opt)
		return 0;
	if (sufficient_len >= HC_OPT_NUM)
		sufficient_len = HC_OPT_NUM - 1;
	if (n < HC_MFLIMIT + 1)
		goto last_literals;
	while (ip <= mflimit) {
		const int llen = (int)(ip - anchor);
		int best_mlen, best_off, cur, last_match_pos = 0, first_len, first_off;

		hc_find_longer(h, ip, matchlimit, HC_MINMATCH - 1, nb_searches, &first_len, &first_off);
		if (first_len == 0) {
			ip++;
			continue;
		}
		if ((size_t)first_len > sufficient_len) {
			/* good enough: immediate encoding */
			if (hc_encode(s, &ip, &op, &anchor, first_len, ip - (uint32_t)first_off, oend))
				goto done;
			continue;
		}
		/* prices of the first positions (literals) and of the first match */
		for (int r = 0; r < HC_MINMATCH; r++) {
			opt[r].mlen = 1;
			opt[r].off = 0;
			opt[r].litlen = llen + r;
			opt[r].price = hc_lit_price(llen + r);
		}
		for (int ml = HC_MINMATCH; ml <= first_len; ml++) {
			opt[ml].mlen = ml;
			opt[ml].off = first_off;
			opt[ml].litlen = llen;
			opt[ml].price = hc_seq_price(llen, ml);
		}
		last_match_pos = first_len;
		for (int a = 1; a <= HC_TRAILING; a++) {
			opt[last_match_pos + a].mlen = 1;
			opt[last_match_pos + a].off = 0;
			opt[last_match_pos + a].litlen = a;
			opt[last_match_pos + a].price = opt[last_match_pos].price + hc_lit_price(a);
		}
		/* further positions */
		for (cur = 1; cur < last_match_pos; cur++) {
			const uint32_t cur_pos = ip + (uint32_t)cur;
			int new_len, new_off;
			if (cur_pos > mflimit)
				break;
			if (full_update) {
				if (opt[cur + 1].price <= opt[cur].price && opt[cur + HC_MINMATCH].price < opt[cur].price + 3)
					continue;
			} else {
				if (opt[cur + 1].price <= opt[cur].price)
					continue;
			}
			if (full_update)
				hc_find_longer(h, cur_pos, matchlimit, HC_MINMATCH - 1, nb_searches, &new_len, &new_off);
			else
				hc_find_longer(h, cur_pos, matchlimit, last_match_pos - cur, nb_searches, &new_len, &new_off);
			if (!new_len)
				continue;
			if ((size_t)new_len > sufficient_len || new_len + cur >= HC_OPT_NUM) {
				best_mlen = new_len;
				best_off = new_off;
				last_match_pos = cur + 1;
				goto encode;
			}
			/* before the match: literals */
			{
				const int base_litlen = opt[cur].litlen;
				for (int litlen = 1; litlen < HC_MINMATCH; litlen++) {
					const int price = opt[cur].price - hc_lit_price(base_litlen) + hc_lit_price(base_litlen + litlen);
					const int pos = cur + litlen;
					if (price < opt[pos].price) {
						opt[pos].mlen = 1;
						opt[pos].off = 0;
						opt[pos].litlen = base_litlen + litlen;
						opt[pos].price = price;
					}
				}
			}
			/* the match at cur */
			for (int ml = HC_MINMATCH; ml <= new_len; ml++) {
				const int pos = cur + ml;
				int price, ll;
				if (opt[cur].mlen == 1) {
					ll = opt[cur].litlen;
					price = ((cur > ll) ? opt[cur - ll].price : 0) + hc_seq_price(ll, ml);
				} else {
					ll = 0;
					price = opt[cur].price + hc_seq_price(0, ml);
				}
				if (pos > last_match_pos + HC_TRAILING || price <= opt[pos].price) {
					if (ml == new_len && last_match_pos < pos)
						last_match_pos = pos;
					opt[pos].mlen = ml;
					opt[pos].off = new_off;
					opt[pos].litlen = ll;
					opt[pos].price = price;
				}
			}
			for (int a = 1; a <= HC_TRAILING; a++) {
				opt[last_match_pos + a].mlen = 1;
				opt[last_match_pos + a].off = 0;
				opt[last_match_pos + a].litlen = a;
				opt[last_match_pos + a].price = opt[last_match_pos].price + hc_lit_price(a);
			}
		}
		best_mlen = opt[last_match_pos].mlen;
		best_off = opt[last_match_pos].off;
		cur = last_match_pos - best_mlen;
encode:
		/* reverse traversal: the chosen sequences, first to last */
		{
			int candidate_pos = cur, selected_ml = best_mlen, selected_off = best_off;
			for (;;) {
				const int next_ml = opt[candidate_pos].mlen, next_off = opt[candidate_pos].off;
				opt[candidate_pos].mlen = selected_ml;
				opt[candidate_pos].off = selected_off;
				selected_ml = next_ml;
				selected_off = next_off;
				if (next_ml > candidate_pos)
					break;
				candidate_pos -= next_ml;
			}
		}
		{
			int r = 0;
			while (r < last_match_pos) {
				const int ml = opt[r].mlen, offset = opt[r].off;
				if (ml == 1) {
					ip++;
					r++;
					continue;
				}
				r += ml;
				if (hc_encode(s, &ip, &op, &anchor, ml, ip - (uint32_t)offset, oend))
					goto done;
			}
		}
	}
last_literals:
	{
		const size_t run = (size_t)(iend - anchor);
		const size_t lit_len = (run + 255 - 15) / 255;
		if (op + 1 + lit_len + run > oend)
			goto done;
		if (run >= 15) {
			size_t acc = run - 15;
			*op++ = 15 << 4;
			for (; acc >= 255; acc -= 255)
				*op++ = 255;
			*op++ = (uint8_t)acc;
		} else {
			*op++ = (uint8_t)(run << 4);
		}
		memcpy(op, s + anchor, run);
		op += run;
	}
	result = (size_t)(op - dst);
done:
	free(opt);
	return result;
}

static int hc_attempts(int level)
{
	static const int a[] = {2, 2, 2, 4, 8, 16, 32, 64, 128, 256, 96, 512, 16384};
	return level >= 0 && level <= 12 ? a[level] : 0;
}

int zo_lz4hc_level_supported(int level) { return level >= 3 && level <= 12; }

static size_t hc_any_block(zo_hc *h, uint32_t start, uint32_t n, uint8_t *dst, size_t cap, int level)
{
	if (level >= 10)
		return hc_opt_block(h, start, n, dst, cap, hc_attempts(level), level == 10 ? 64 : level == 11 ? 128 : HC_OPT_NUM,
				    level == 12);
	return hc_block(h, start, n, dst, cap, hc_attempts(level));
}

/* one LZ4 frame as LZ4F_compressFrame writes it for the prefs of lz4-mt at an HC level: same
 * container as zo_lz4f_compress (lz4_oracle.c), blocks from the HC parser with one context per frame
 * (linked blocks: the chains run on across the chunk's blocks) */
size_t zo_lz4f_compress_hc(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int level)
{
	uint8_t *op = dst;
	size_t pos;
	const int single = (n <= ZO_BLOCK_MAX);
	const uint8_t flg = (uint8_t)(0x40 | (single ? 0x20 : 0) | (n ? 0x08 : 0) | 0x04);
	const size_t hdr = n ? 15 : 7;
	zo_hc *h;

	if (!zo_lz4hc_level_supported(level) || cap < zo_lz4f_bound(n))
		return (size_t)-1;
	h = (zo_hc *)calloc(1, sizeof *h);
	if (!h)
		return (size_t)-1;
	h->src = src;
	h->next_to_update = HC_BASE;
	op[0] = 0x04; op[1] = 0x22; op[2] = 0x4D; op[3] = 0x18;
	op[4] = flg;
	op[5] = 0x40;
	if (n) {
		uint64_t v = n;
		for (int i = 0; i < 8; i++)
			op[6 + i] = (uint8_t)(v >> (8 * i));
	}
	op[hdr - 1] = (uint8_t)(zo_xxh32(op + 4, hdr - 5, 0) >> 8);
	op += hdr;
	for (pos = 0; pos < n; pos += ZO_BLOCK_MAX) {
		const size_t len = n - pos < ZO_BLOCK_MAX ? n - pos : ZO_BLOCK_MAX;
		const size_t c = hc_any_block(h, (uint32_t)pos, (uint32_t)len, op + 4, len - 1, level);
		uint32_t bh;
		if (c == 0) { /* did not shrink: stored block; the chains keep what the attempt inserted */
			bh = (uint32_t)len | 0x80000000u;
			memcpy(op + 4, src + pos, len);
		} else {
			bh = (uint32_t)c;
		}
		op[0] = (uint8_t)bh; op[1] = (uint8_t)(bh >> 8); op[2] = (uint8_t)(bh >> 16); op[3] = (uint8_t)(bh >> 24);
		op += 4 + (bh & 0x7FFFFFFFu);
	}
	op[0] = op[1] = op[2] = op[3] = 0; /* end mark */
	op += 4;
	{
		const uint32_t x = zo_xxh32(src, n, 0);
		op[0] = (uint8_t)x; op[1] = (uint8_t)(x >> 8); op[2] = (uint8_t)(x >> 16); op[3] = (uint8_t)(x >> 24);
		op += 4;
	}
	free(h);
	return (size_t)(op - dst);
}

/* the MT stream at an HC level (record framing as zo_lz4mt_compress) */
size_t zo_lz4mt_compress_level(const uint8_t *src, size_t n, size_t chunk, uint8_t *dst, size_t cap, int level)
{
	size_t pos = 0, out = 0;
	int first = 1;
	if (level < 3)
		return zo_lz4mt_compress(src, n, chunk, dst, cap);
	while (pos < n || first) {
		const size_t len = n - pos < chunk ? n - pos : chunk;
		size_t c;
		if (cap - out < zo_lz4f_bound(len) + 12)
			return (size_t)-1;
		c = zo_lz4f_compress_hc(src + pos, len, dst + out + 12, cap - out - 12, level);
		if (c == (size_t)-1)
			return (size_t)-1;
		dst[out + 0] = 0x50; dst[out + 1] = 0x2A; dst[out + 2] = 0x4D; dst[out + 3] = 0x18;
		dst[out + 4] = 4; dst[out + 5] = dst[out + 6] = dst[out + 7] = 0;
		dst[out + 8] = (uint8_t)c; dst[out + 9] = (uint8_t)(c >> 8);
		dst[out + 10] = (uint8_t)(c >> 16); dst[out + 11] = (uint8_t)(c >> 24);
		out += 12 + c;
		pos += len;
		first = 0;
	}
	return out;
}
