/*
 * snappy_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the snappy raw format for the
 * snappy-mt path (decoder + the snappy-mt record walk).  Never linked into the product.
 *
 * Where the algorithm lives.  lib/snappy-mt_compress.c:264-276 calls snappy_compress(&env, ...) and
 * lib/snappy-mt_decompress.c:262-267,350 calls snappy_uncompressed_length / snappy_uncompress of
 * "snappy.h" with a struct snappy_env: the C port of snappy (Andi Kleen's snappy-c), which the
 * zstdmt repository vendors in a directory that is not part of /root/reference.  What is restated
 * here is the published format (google/snappy format_description.txt): a varint with the
 * uncompressed length, then literals (tag & 3 == 0: length - 1 in the upper six bits, 60..63 = that
 * many + 1 - 60 little-endian length bytes follow) and copies (1: length 4..11, 11-bit offset; 2:
 * length 1..64, 16-bit offset; 3: length 1..64, 32-bit offset); a copy may overlap its own output.
 *
 * Pinning.  The image holds libsnappy 1.1.8 (/opt/conda/lib/libsnappy.so.1, snappy-c.h API), a
 * different implementation of the same format: tests/test_snappy_oracle.py checks this decoder
 * against it both ways (streams written by libsnappy decode to the input; damaged streams get the
 * verdict of snappy_uncompress) and against the committed vectors of tests/golden/snappy.  The
 * reference's own encoder bytes cannot be pinned (its library is absent), so the bar for
 * compression is decompress-identical, as for zstd and brotli.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "zmt_oracle.h"

#define SERR ((size_t)-1)

/* varint preamble: returns its length in bytes (1..5) or 0 */
size_t zo_snappy_uncompressed_length(const uint8_t *src, size_t n, uint32_t *out)
{
	uint32_t v = 0;
	for (size_t i = 0; i < 5 && i < n; i++) {
		const uint32_t b = src[i];
		if (i == 4 && b > 15)
			return 0; /* more than 32 bits */
		v |= (b & 127u) << (7 * i);
		if (b < 128) {
			*out = v;
			return i + 1;
		}
	}
	return 0;
}

/* one raw snappy stream -> dst; returns the decoded length (== the preamble) or SERR */
size_t zo_snappy_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
	uint32_t want;
	size_t ip = zo_snappy_uncompressed_length(src, n, &want), op = 0;
	if (!ip || want > cap)
		return SERR;
	while (ip < n) {
		const uint32_t tag = src[ip++];
		uint32_t len, off;
		switch (tag & 3) {
		case 0:
			len = tag >> 2;
			if (len >= 60) {
				const uint32_t nb = len - 59;
				if (n - ip < nb)
					return SERR;
				len = 0;
				for (uint32_t k = 0; k < nb; k++)
					len |= (uint32_t)src[ip + k] << (8 * k);
				ip += nb;
			}
			if (len == 0xFFFFFFFFu)
				return SERR;
			len += 1;
			if (len > n - ip || len > want - op)
				return SERR;
			memcpy(dst + op, src + ip, len);
			ip += len;
			op += len;
			continue;
		case 1:
			if (n - ip < 1)
				return SERR;
			len = 4 + ((tag >> 2) & 7);
			off = (tag >> 5) << 8 | src[ip];
			ip += 1;
			break;
		case 2:
			if (n - ip < 2)
				return SERR;
			len = 1 + (tag >> 2);
			off = (uint32_t)src[ip] | (uint32_t)src[ip + 1] << 8;
			ip += 2;
			break;
		default:
			if (n - ip < 4)
				return SERR;
			len = 1 + (tag >> 2);
			off = (uint32_t)src[ip] | (uint32_t)src[ip + 1] << 8 | (uint32_t)src[ip + 2] << 16 |
			      (uint32_t)src[ip + 3] << 24;
			ip += 4;
			break;
		}
		if (off == 0 || off > op || len > want - op)
			return SERR;
		for (uint32_t k = 0; k < len; k++) /* byte-wise: an overlapping copy replicates */
			dst[op + k] = dst[op + k - off];
		op += len;
	}
	return op == want ? op : SERR;
}

/* snappy-mt record walk (lib/snappy-mt_decompress.c:185-290 pt_read, :292-377 pt_decompress): 16-byte
 * headers -- skippable magic, 8, compressed size, "SP", a hint the decoder ignores -- each followed by
 * one raw snappy stream whose preamble sizes the output.  Returns total decoded bytes or SERR. */
size_t zo_snappymt_decompress(const uint8_t *src, size_t slen, uint8_t *dst, size_t cap)
{
	size_t ip = 0, op = 0;
	if (slen < 4)
		return SERR;
	while (ip < slen) {
		if (slen - ip < 16)
			return SERR;
		const uint32_t magic = (uint32_t)src[ip] | (uint32_t)src[ip + 1] << 8 | (uint32_t)src[ip + 2] << 16 |
				       (uint32_t)src[ip + 3] << 24;
		const uint32_t eight = (uint32_t)src[ip + 4] | (uint32_t)src[ip + 5] << 8 | (uint32_t)src[ip + 6] << 16 |
				       (uint32_t)src[ip + 7] << 24;
		const size_t csize = (uint32_t)src[ip + 8] | (uint32_t)src[ip + 9] << 8 | (uint32_t)src[ip + 10] << 16 |
				     (uint32_t)src[ip + 11] << 24;
		if (magic != ZO_SKIP_MAGIC || eight != 8 || ((uint32_t)src[ip + 12] | (uint32_t)src[ip + 13] << 8) != 0x5053u)
			return SERR;
		ip += 16;
		if (csize > slen - ip)
			return SERR;
		const size_t r = zo_snappy_decompress(src + ip, csize, dst + op, cap - op);
		if (r == SERR)
			return SERR;
		op += r;
		ip += csize;
	}
	return op;
}
