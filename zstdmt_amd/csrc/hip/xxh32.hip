/*
 * xxh32.hip -- batched XXH32 (seed 0) over many independent items (chunk contents).
 *
 * Replaces the XXH32 that LZ4F_compressFrame / LZ4F_decompress run over every chunk for the
 * content checksum (reference call sites lib/lz4-mt_compress.c:281, lib/lz4-mt_decompress.c:350;
 * flag set at lib/lz4-mt_compress.c:145).
 *
 * XXH32 is four independent serial accumulator chains per item (rotl/mul, not associative), so
 * the parallelism is across items: a quad of lanes owns one item, lane a of the quad runs
 * accumulator a.  A wave therefore streams 16 items; every wave-level load fetches 16 B from each
 * of them (the four lanes of a quad read one 16-byte stripe).
 * HBM-bound: algorithmic bytes = item length, read once.
 */
#include "lz4_common.h"

#ifndef XXH_BLOCK
#define XXH_BLOCK 256
#endif

extern "C" __global__ void __launch_bounds__(XXH_BLOCK)
zmt_xxh32_kernel(const u8 *__restrict__ base, const u64 *__restrict__ off,
		 const u32 *__restrict__ len, u32 n, u32 *__restrict__ out,
		 const u32 *__restrict__ expect, const u32 *__restrict__ expect_valid,
		 u32 *__restrict__ status)
{
	const u32 gtid = blockIdx.x * XXH_BLOCK + threadIdx.x;
	const u32 item = gtid >> 2;
	const u32 a = gtid & 3;
	const bool live = item < n;
	const u8 *p = live ? base + off[item] : base;
	const u32 L = live ? len[item] : 0;
	u32 v = 0;

	if (L >= 16) {
		u32 acc = (a == 0) ? XP1 + XP2 : (a == 1) ? XP2 : (a == 2) ? 0u : 0u - XP1;
		const u8 *q = p + a * 4;
		u32 ns = L >> 4, s = 0;
		for (; s + 4 <= ns; s += 4) {
			u32 x0 = ld32u(q), x1 = ld32u(q + 16), x2 = ld32u(q + 32), x3 = ld32u(q + 48);
			acc = xxh_round(acc, x0);
			acc = xxh_round(acc, x1);
			acc = xxh_round(acc, x2);
			acc = xxh_round(acc, x3);
			q += 64;
		}
		for (; s < ns; s++) {
			acc = xxh_round(acc, ld32u(q));
			q += 16;
		}
		v = rotl32(acc, (a == 0) ? 1 : (a == 1) ? 7 : (a == 2) ? 12 : 18);
	}
	/* quad reduction, wave-uniform */
	{
		int l = wv_lane();
		v += wv_shfl(v, l ^ 1);
		v += wv_shfl(v, l ^ 2);
	}
	if (live && a == 0) {
		u32 h = (L >= 16 ? v : XP5) + L;
		h = xxh_tail(h, p + (L & ~15u), L & 15);
		if (out)
			out[item] = h;
		if (expect && expect_valid[item] && expect[item] != h && status[item] == ST_OK)
			status[item] = ST_BAD_CHECKSUM;
	}
}
