/*
 * snappy.hip -- snappy-mt on gfx950: raw snappy streams, one wave per record.
 *
 * Replaces snappy_compress(&env, chunk, n, dst, &len) as called per chunk by the reference
 * (/root/reference/lib/snappy-mt_compress.c:264-276) together with the 16-byte record header emit
 * (:280-300), and snappy_uncompressed_length + snappy_uncompress as called per record
 * (/root/reference/lib/snappy-mt_decompress.c:262-267, :350).  The reference's snappy library is a C
 * port the zstdmt repository vendors outside /root/reference, so its bytes cannot be pinned: the bar
 * for the encoder is decompress-identical (valid raw snappy that decodes to the chunk), for the
 * decoder byte-identical output and the verdicts of a snappy decoder (oracle/snappy_oracle.c, pinned
 * against the image's libsnappy 1.1.8).  Format: google/snappy format_description.txt.
 *
 *   zmt_snappy_enc_kernel   persistent waves, one record (chunk) at a time, its 64 KiB blocks in turn
 *                           (matches stay inside a block, so every copy has a 16-bit offset):
 *     match finding         64 positions per step, one per lane: hash of 6 bytes, 4096-entry u16 LDS
 *                           table (newest position of a step wins), candidates verified against a
 *                           24-byte window (4 bytes in front, 20 from the candidate on), loads two
 *                           steps ahead, long matches extended 512 bytes per step by the wave -- the
 *                           match finder of zstd_enc.hip;
 *     parse                 greedy, leftmost match first, resolved with ballots, matches grow
 *                           backwards into the literals in front of them;
 *     elements              64 sequences (literal run + match) at a time: every lane sizes its own
 *                           elements, a prefix sum places them, the lane writes them (literal tag,
 *                           literal bytes, copies of at most 64 bytes: 1-byte-offset form for 4..11
 *                           bytes below 2 KiB, else 2-byte-offset form); long literal runs by the
 *                           whole wave.  Then the record header.
 *   zmt_snappy_dec_kernel   one wave per record: preamble, then element by element -- the tag and its
 *                           fields are wave-uniform, the bytes are moved by the wave (overlapping
 *                           copies as pattern repeats); every length and offset is checked against
 *                           the input left, the output produced and the size the preamble states.
 */
#include "lz4_common.h"
#include "lz4_frame.h"
#include "match_copy.h"

#define SN_BLOCK 65536u
#define SN_HLOG 12
#ifndef SN_MINMATCH
#define SN_MINMATCH 6u
#endif
#ifndef SN_HBYTES
#define SN_HBYTES 6
#endif
#ifndef SN_LAZYW
#define SN_LAZYW 4u /* positions the greedy parse looks ahead for a longer match (0 = none) */
#endif
#define SN_FWD 20u /* bytes of a match measured by the lane that found it; longer ones by the whole wave */
#define SN_CAP 64u /* longer literal runs are copied by the whole wave */
#define SN_HDR 16u /* record header: skippable magic, 8, payload size, "SP", hint */
#define SN_MAGIC_SKIPPABLE 0x184D2A50u
#define SN_MAGICNUMBER 0x5053u /* lib/snappy-mt.h:16 */
#define SN_HASH(v) ((u32)((((v) << (64 - 8 * SN_HBYTES)) * 0x9E3779B185EBCA87ull) >> (64 - SN_HLOG)))

/* status codes of include/gpumt.h */
#define SN_ST_OK 0u
#define SN_ST_BAD_BLOCK 3u
#define SN_ST_SIZE_MISMATCH 4u

struct SnEncLds {
	u16 table[1u << SN_HLOG]; /* low 16 bits (= all, inside a 64 KiB block) of the newest position of a hash */
};

/* elements of up to 64 sequences, lane i holding sequence i: literals src[lp, lp + ll), then a match of ml
 * bytes at distance of (ml == 0: literals only).  Returns the bytes written at o. */
static __device__ __forceinline__ u32 sn_emit(u8 *o, const u8 *src, u32 count, u32 lp, u32 ll, u32 ml, u32 of, int lane)
{
	const bool on = (u32)lane < count;
	const u32 lh = !on || ll == 0 ? 0u : ll <= 60 ? 1u : ll <= 256 ? 2u : 3u;
	const u32 pieces = on ? (ml + 63) / 64 : 0u;
	const u32 last = ml - 64u * (pieces ? pieces - 1 : 0u); /* 1..64 */
	const bool c1 = pieces && last >= 4 && last <= 11 && of < 2048;
	const u32 size = on ? lh + ll + 3u * pieces - (c1 ? 1u : 0u) : 0u;
	const u32 incl = wv_scan_incl(size);
	u8 *p = o + (incl - size);
	if (lh == 1) {
		p[0] = (u8)((ll - 1) << 2);
	} else if (lh == 2) {
		p[0] = 60u << 2;
		p[1] = (u8)(ll - 1);
	} else if (lh == 3) {
		p[0] = 61u << 2;
		p[1] = (u8)(ll - 1);
		p[2] = (u8)((ll - 1) >> 8);
	}
	if (on && ll && ll <= SN_CAP)
		g_copy(p + lh, src + lp, ll);
	u64 lm = wv_ballot(on && ll > SN_CAP);
	while (lm) {
		const int j = wv_ffs(lm) - 1;
		lm &= lm - 1;
		wave_copy(o + wv_readlane(incl - size + lh, j), src + wv_readlane(lp, j), wv_readlane(ll, j), lane);
	}
	if (pieces) {
		u8 *q = p + lh + ll;
		for (u32 k = 1; k < pieces; k++) {
			q[0] = (u8)(63u << 2 | 2u);
			q[1] = (u8)of;
			q[2] = (u8)(of >> 8);
			q += 3;
		}
		if (c1) {
			q[0] = (u8)(1u | (last - 4) << 2 | (of >> 8) << 5);
			q[1] = (u8)of;
		} else {
			q[0] = (u8)((last - 1) << 2 | 2u);
			q[1] = (u8)of;
			q[2] = (u8)(of >> 8);
		}
	}
	return wv_readlane(incl, 63);
}

/* 128 VGPRs = 4 waves per SIMD; with 8 KiB of LDS a CU holds 16 of these latency-bound waves */
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
zmt_snappy_enc_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nrec, u8 *__restrict__ slots, u64 stride,
		      u32 *__restrict__ rec_len)
{
	__shared__ SnEncLds L;
	const int lane = wv_lane();
	for (u32 rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
		const u64 cstart = (u64)rec * chunk;
		const u32 clen = n > cstart ? (u32)(n - cstart < chunk ? n - cstart : chunk) : 0u;
		u8 *const out = slots + (u64)rec * stride;
		/* preamble: the uncompressed length as a varint */
		u32 op = SN_HDR;
		{
			u32 v = clen;
			while (v >= 128) {
				if (lane == 0)
					out[op] = (u8)(v | 128u);
				op++;
				v >>= 7;
			}
			if (lane == 0)
				out[op] = (u8)v;
			op++;
		}
		for (u32 bstart = 0; bstart < clen; bstart += SN_BLOCK) {
			const u32 bsize = clen - bstart < SN_BLOCK ? clen - bstart : SN_BLOCK;
			const u8 *src = in + cstart + bstart;
			for (u32 i = (u32)lane; i < (1u << SN_HLOG); i += 64)
				L.table[i] = 0;
			wv_sync();
			u32 ns = 0, anchor = 0, cursor = 0;
			u32 r_lp = 0, r_ll = 0, r_ml = 0, r_of = 0;
			const u32 steps = bsize >= SN_MINMATCH ? (bsize - SN_MINMATCH) / 64 + 1 : 0;
/* All loads of the pipeline are unconditional (addresses clamped, results of invalid lanes ignored), as in
 * zstd_enc.hip: a load under an exec mask needs its destination initialised first. */
#define SN_LOADV(t, V)                                                                             \
	do {                                                                                       \
		const u32 p_ = (t) * 64u + (u32)lane;                                              \
		(V) = ld64u(src + (p_ < bsize ? p_ : bsize - 1));                                  \
	} while (0)
#define SN_LOOKUP(t, V, Cc, M)                                                                     \
	do {                                                                                       \
		const u32 p_ = (t) * 64u + (u32)lane;                                              \
		const bool ok_ = (t) < steps && p_ + SN_MINMATCH <= bsize;                         \
		const u32 h_ = SN_HASH(V);                                                         \
		const u32 e_ = ok_ ? L.table[h_] : 0;                                              \
		wv_sync();                                                                         \
		if (ok_)                                                                           \
			L.table[h_] = (u16)p_;                                                     \
		wv_sync();                                                                         \
		/* equal hashes inside one step: the highest position must stay */                     \
		while (wv_any(ok_ && L.table[h_] < (u16)p_)) {                                     \
			if (ok_ && L.table[h_] < (u16)p_)                                          \
				L.table[h_] = (u16)p_;                                             \
			wv_sync();                                                                 \
		}                                                                                  \
		/* an empty slot reads as position 0: a real candidate, verified like any other; the    \
		 * first four positions of a block are none (the window starts 4 bytes in front) */    \
		(Cc) = (ok_ && e_ < p_ && e_ >= 4u) ? e_ : 0xFFFFFFFFu;                            \
		{                                                                                  \
			const bool v_ = (Cc) != 0xFFFFFFFFu;                                       \
			const u8 *cp_ = src + (v_ ? (Cc) - 4u : 0u);                               \
			const u8 *ip_ = src + (v_ ? p_ : 0u);                                      \
			(M).v = (V);                                                               \
			(M).a = ld64u(cp_);                                                        \
			(M).b = ld64u(cp_ + 8);                                                    \
			(M).c = ld64u(cp_ + 16);                                                   \
			(M).d = ld64u(ip_ + 8);                                                    \
			(M).e = ld32u(ip_ + 16);                                                   \
			const u32 n_ = wv_shfl((u32)(V), lane - 4);                                \
			(M).pb = lane >= 4 ? n_ : ~(u32)(M).a;                                     \
		}                                                                                  \
	} while (0)
			struct Cmp {
				u64 v, a, b, c, d; /* input bytes 0..7, candidate window (bytes -4..19), input bytes 8..15 */
				u32 e, pb;         /* input bytes 16..19, input bytes -4..-1 (lanes 4..63) */
			};
			Cmp M[3];
			u32 Cn[3];
			u64 V[3];
			SN_LOADV(0u, V[0]);
			SN_LOADV(1u, V[1]);
			SN_LOADV(2u, V[2]);
			SN_LOOKUP(0u, V[0], Cn[0], M[0]);
			SN_LOOKUP(1u, V[1], Cn[1], M[1]);
			for (u32 t0 = 0; t0 < steps; t0 += 3) {
				ZMT_UNROLL
				for (int k = 0; k < 3; k++) {
					const u32 t = t0 + (u32)k;
					if (t >= steps)
						break;
					SN_LOADV(t + 3, V[k]);
					SN_LOOKUP(t + 2, V[(k + 2) % 3], Cn[(k + 2) % 3], M[(k + 2) % 3]);
					const Cmp &m0 = M[k];
					const u32 c0 = Cn[k];
					const u32 p0 = t * 64u, p = p0 + (u32)lane;
					if (p0 + 64 <= cursor)
						continue; /* the whole step lies inside the previous match */
					const u64 x0 = m0.v ^ (m0.a >> 32 | m0.b << 32), x1 = m0.d ^ (m0.b >> 32 | m0.c << 32);
					const u32 x2 = m0.e ^ (u32)(m0.c >> 32);
					u32 m = x0   ? (u32)__builtin_ctzll(x0) >> 3
						: x1 ? 8u + ((u32)__builtin_ctzll(x1) >> 3)
						: x2 ? 16u + ((u32)__builtin_ctz(x2) >> 3)
						     : SN_FWD;
					const bool cand = c0 != 0xFFFFFFFFu && p >= cursor;
					if (cand && m > bsize - p)
						m = bsize - p;
					const u32 xb = m0.pb ^ (u32)m0.a;
					const u32 back = !cand ? 0u : xb ? (u32)__builtin_clz(xb) >> 3 : 4u;
					u64 mask = wv_ballot(cand && m >= SN_MINMATCH);
					/* look-ahead, as in zstd_enc.hip (round 5): a match that starts d <= SN_LAZYW bytes further on wins
					 * when it is longer by more than the d literals it adds (bytes it reaches backwards count for it) */
					if (SN_LAZYW) {
						u32 w = (cand && m >= SN_MINMATCH) ? (m | back << 8) : 0u;
						bool lz = false;
						ZMT_UNROLL
						for (u32 d = 1; d <= SN_LAZYW; d++) {
							w = wv_shl1(w, 0u);
							const u32 bq = (w >> 8) < d ? (w >> 8) : d;
							lz = lz || (w != 0u && (w & 255u) + bq >= m + d + 1u);
						}
						mask &= ~wv_ballot(lz && m < SN_FWD);
					}
					while (mask) {
						const int j = wv_ffs(mask) - 1;
						mask &= mask - 1;
						const u32 pj = p0 + (u32)j;
						if (pj < cursor)
							continue;
						const u32 cj = wv_readlane(c0, j);
						u32 ml = wv_readlane(m, j);
						if (ml == SN_FWD) {
							for (u32 base = SN_FWD;; base += 512) {
								const u32 o = base + 8u * (u32)lane;
								u32 kk = 0;
								bool stop = true;
								if (pj + o < bsize) {
									const u64 y = ld64u(src + pj + o) ^ ld64u(src + cj + o);
									kk = y ? (u32)__builtin_ctzll(y) >> 3 : 8u;
									stop = kk < 8;
								}
								const u64 sm = wv_ballot(stop);
								if (sm) {
									const int f = wv_ffs(sm) - 1;
									ml = base + 8u * (u32)f + wv_readlane(kk, f);
									break;
								}
							}
							if (ml > bsize - pj)
								ml = bsize - pj;
						}
						u32 bk = wv_readlane(back, j);
						bk = bk < pj - anchor ? bk : pj - anchor;
						{
							const bool me = (u32)lane == (ns & 63);
							r_lp = me ? anchor : r_lp;
							r_ll = me ? pj - bk - anchor : r_ll;
							r_ml = me ? ml + bk : r_ml;
							r_of = me ? pj - cj : r_of;
						}
						ns++;
						if ((ns & 63) == 0)
							op += sn_emit(out + op, src, 64, r_lp, r_ll, r_ml, r_of, lane);
						anchor = cursor = pj + ml;
						mask = cursor - p0 >= 64 ? 0 : mask & ~((1ull << (cursor - p0)) - 1);
					}
				}
			}
#undef SN_LOADV
#undef SN_LOOKUP
			/* the sequences still in registers, then the literals behind the last match */
			{
				const u32 left = ns & 63;
				const bool tail = (u32)lane == left && anchor < bsize;
				r_lp = tail ? anchor : r_lp;
				r_ll = tail ? bsize - anchor : r_ll;
				r_ml = tail ? 0u : r_ml;
				op += sn_emit(out + op, src, left + (anchor < bsize ? 1u : 0u), r_lp, r_ll, r_ml, r_of, lane);
			}
			wv_sync();
		}
		/* record header (lib/snappy-mt_compress.c:280-300): the hint counts 64 KiB units as the
		 * reference does (its decoder ignores it and sizes the output from the preamble) */
		if (lane == 0) {
			const u32 payload = op - SN_HDR;
			const u32 hint = clen < chunk ? (clen >> 16) + 1 : chunk >> 16;
			st32u(out, SN_MAGIC_SKIPPABLE);
			st32u(out + 4, 8u);
			st32u(out + 8, payload);
			out[12] = (u8)SN_MAGICNUMBER;
			out[13] = (u8)(SN_MAGICNUMBER >> 8);
			out[14] = (u8)hint;
			out[15] = (u8)(hint >> 8);
			rec_len[rec] = op;
		}
		wave_mem_fence();
	}
}

/* ---------------------------------------------------------------------------------------- decoder
 * Stream i = stream + rec_off[i], rec_len[i] bytes (the payload behind the 16-byte header, which the
 * host parses while reading); its output goes to out + out_off[i], out_cap[i] bytes at most.
 * out_len[i] = decoded size, status[i] = OK / BAD_BLOCK (malformed or truncated) / SIZE_MISMATCH (the
 * preamble asks for more than the capacity). */
extern "C" __global__ void __launch_bounds__(64)
zmt_snappy_dec_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off, const u32 *__restrict__ rec_len,
		      u32 nrec, u8 *__restrict__ out, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap,
		      u32 *__restrict__ out_len, u32 *__restrict__ status)
{
	const int lane = wv_lane();
	for (u32 r = blockIdx.x; r < nrec; r += gridDim.x) {
		const u8 *src = stream + rec_off[r];
		const u32 n = rec_len[r], cap = out_cap[r];
		u8 *dst = out + out_off[r];
		u32 st = SN_ST_OK, ip = 0, op = 0, want = 0;
		{ /* varint preamble, at most 5 bytes and 32 bits */
			u32 sh = 0;
			for (;;) {
				if (ip >= n || ip >= 5) {
					st = SN_ST_BAD_BLOCK;
					break;
				}
				const u32 b = uld8(src + ip);
				if (ip == 4 && b > 15) {
					st = SN_ST_BAD_BLOCK;
					break;
				}
				want |= (b & 127u) << sh;
				sh += 7;
				ip++;
				if (b < 128)
					break;
			}
			if (st == SN_ST_OK && want > cap)
				st = SN_ST_SIZE_MISMATCH;
		}
		bool dirty = false; /* stores since the last fence: a copy must see them */
		while (st == SN_ST_OK && ip < n) {
			/* tag + up to 4 field bytes in one wave-uniform load (the stream has 256 readable bytes behind it) */
			const u64 w = (u64)uld32(src + ip) | (u64)uld8(src + ip + 4) << 32;
			const u32 tag = (u32)w & 255u, kind = tag & 3u;
			u32 len, off = 0, used;
			if (kind == 0) {
				len = tag >> 2;
				used = 1;
				if (len >= 60) {
					const u32 nb = len - 59;
					len = (u32)(w >> 8) & (nb == 4 ? 0xFFFFFFFFu : (1u << (8 * nb)) - 1u);
					used = 1 + nb;
					if (len == 0xFFFFFFFFu) {
						st = SN_ST_BAD_BLOCK;
						break;
					}
				}
				len += 1;
				if (used > n - ip || len > n - ip - used || len > want - op) {
					st = SN_ST_BAD_BLOCK;
					break;
				}
				wave_copy(dst + op, src + ip + used, len, lane);
				dirty = true;
				ip += used + len;
				op += len;
				continue;
			}
			if (kind == 1) {
				len = 4 + ((tag >> 2) & 7u);
				off = (tag >> 5) << 8 | ((u32)(w >> 8) & 255u);
				used = 2;
			} else if (kind == 2) {
				len = 1 + (tag >> 2);
				off = (u32)(w >> 8) & 0xFFFFu;
				used = 3;
			} else {
				len = 1 + (tag >> 2);
				off = (u32)(w >> 8);
				used = 5;
			}
			if (used > n - ip || off == 0 || off > op || len > want - op) {
				st = SN_ST_BAD_BLOCK;
				break;
			}
			if (dirty) {
				wave_mem_fence();
				dirty = false;
			}
			if (off >= len) {
				if ((u32)lane < len)
					dst[op + (u32)lane] = dst[op - off + (u32)lane];
			} else {
				/* overlapping: the bytes repeat with period off, all of it already written */
				if ((u32)lane < len)
					dst[op + (u32)lane] = dst[op - off + (u32)lane % off];
			}
			dirty = true;
			ip += used;
			op += len;
		}
		if (st == SN_ST_OK && op != want)
			st = SN_ST_BAD_BLOCK;
		if (lane == 0) {
			out_len[r] = st == SN_ST_OK ? op : 0u;
			status[r] = st;
		}
		wave_mem_fence();
	}
}

/* ------------------------------------------------------------------ decoder, batched variant
 * (gpumt_set_variant("snappy_dec", 1); written on the emulator after the round's GPU budget was spent, so
 * it is not the default until it has been timed against the kernel above.)
 * zmt_snappy_dec_kernel pays a dependent load -> store round trip per element.  Here 64 elements are parsed
 * at a time -- one lane walks the tags in a 1 KiB LDS window of the stream and leaves kind, length, offset
 * and positions in LDS -- and then executed by 64 lanes, one element each: literals at once; copies in
 * watermark rounds (a copy runs when everything it reads is in front of the first element still pending,
 * so the first pending one always runs), as the zstd decoder executes its sequences.  Same checks, same
 * verdicts. */
#define SN_WIN 1024u

struct SnDecLds {
	u8 win[SN_WIN + 16];
	u32 e_len[64], e_off[64], e_src[64], e_out[64];
	u32 cnt, next_ip, next_op, err;
};

extern "C" __global__ void __launch_bounds__(64)
zmt_snappy_dec2_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off, const u32 *__restrict__ rec_len,
		       u32 nrec, u8 *__restrict__ out, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap,
		       u32 *__restrict__ out_len, u32 *__restrict__ status)
{
	__shared__ SnDecLds L;
	const int lane = wv_lane();
	for (u32 r = blockIdx.x; r < nrec; r += gridDim.x) {
		const u8 *src = stream + rec_off[r];
		const u32 n = rec_len[r], cap = out_cap[r];
		u8 *dst = out + out_off[r];
		u32 st = SN_ST_OK, ip = 0, op = 0, want = 0;
		{ /* varint preamble, at most 5 bytes and 32 bits */
			u32 sh = 0;
			for (;;) {
				if (ip >= n || ip >= 5) {
					st = SN_ST_BAD_BLOCK;
					break;
				}
				const u32 b = uld8(src + ip);
				if (ip == 4 && b > 15) {
					st = SN_ST_BAD_BLOCK;
					break;
				}
				want |= (b & 127u) << sh;
				sh += 7;
				ip++;
				if (b < 128)
					break;
			}
			if (st == SN_ST_OK && want > cap)
				st = SN_ST_SIZE_MISMATCH;
		}
		while (st == SN_ST_OK && ip < n) {
			/* window: the next 1 KiB of the stream (16 bytes per lane; nothing is read from beyond the
			 * 16 bytes behind the record) */
			const u32 wlen = n - ip < SN_WIN ? n - ip : SN_WIN;
			wv_sync();
			{
				const u32 o = 16u * (u32)lane;
				u64 a = 0, b = 0;
				if (o < wlen) {
					a = ld64u(src + ip + o);
					b = ld64u(src + ip + o + 8);
				}
				__builtin_memcpy(L.win + o, &a, 8);
				__builtin_memcpy(L.win + o + 8, &b, 8);
			}
			wv_sync();
			if (lane == 0) {
				u32 cnt = 0, p = 0, o = op, err = 0;
				while (cnt < 64 && p < wlen) {
					const u32 tag = L.win[p], kind = tag & 3u;
					u32 used = kind == 0 ? ((tag >> 2) >= 60 ? 1 + ((tag >> 2) - 59) : 1) : kind == 1 ? 2 : kind == 2 ? 3 : 5;
					if (p + used > wlen) {
						if (wlen == n - ip)
							err = 1; /* the stream ends inside the element */
						break;       /* else: the window does; it restarts at this element */
					}
					u32 f = 0; /* the field bytes behind the tag */
					for (u32 k = 1; k < used; k++)
						f |= (u32)L.win[p + k] << (8 * (k - 1));
					u32 len, off;
					if (kind == 0) {
						len = (tag >> 2) >= 60 ? f : tag >> 2;
						if (len == 0xFFFFFFFFu || len + 1 > n - (ip + p + used) || len + 1 > want - o) {
							err = 1;
							break;
						}
						len += 1;
						off = 0;
						L.e_src[cnt] = ip + p + used;
						p += used + len;
					} else {
						if (kind == 1) {
							len = 4 + ((tag >> 2) & 7u);
							off = (tag >> 5) << 8 | (f & 255u);
						} else {
							len = 1 + (tag >> 2);
							off = kind == 2 ? f & 0xFFFFu : f;
						}
						if (off == 0 || off > o || len > want - o) {
							err = 1;
							break;
						}
						p += used;
					}
					L.e_len[cnt] = len;
					L.e_off[cnt] = off;
					L.e_out[cnt] = o;
					o += len;
					cnt++;
				}
				L.cnt = cnt;
				L.next_ip = ip + p;
				L.next_op = o;
				L.err = err;
			}
			wv_sync();
			const u32 cnt = L.cnt;
			if (L.err) {
				st = SN_ST_BAD_BLOCK;
				break;
			}
			const bool mine = (u32)lane < cnt;
			const u32 len = mine ? L.e_len[lane] : 0u, off = mine ? L.e_off[lane] : 0u;
			const u32 eo = mine ? L.e_out[lane] : 0u, es = mine ? L.e_src[lane] : 0u;
			/* literals: straight from the stream */
			if (mine && off == 0 && len <= SN_CAP)
				g_copy(dst + eo, src + es, len);
			u64 lm = wv_ballot(mine && off == 0 && len > SN_CAP);
			while (lm) {
				const int j = wv_ffs(lm) - 1;
				lm &= lm - 1;
				wave_copy(dst + wv_readlane(eo, j), src + wv_readlane(es, j), wv_readlane(len, j), lane);
			}
			/* copies: everything in front of the first pending element is written */
			bool pending = mine && off != 0;
			while (wv_any(pending)) {
				wave_mem_fence();
				const int first = wv_ffs(wv_ballot(pending)) - 1;
				const u32 w = wv_readlane(eo, first);
				const u32 src_end = off >= len ? eo - off + len : eo;
				if (pending && src_end <= w) {
					if (off >= len) {
						g_copy(dst + eo, dst + eo - off, len);
					} else {
						u32 j = 0;
						for (u32 k = 0; k < len; k++) { /* the bytes repeat with period off */
							dst[eo + k] = dst[eo - off + j];
							if (++j == off)
								j = 0;
						}
					}
					pending = false;
				}
			}
			wave_mem_fence();
			ip = L.next_ip;
			op = L.next_op;
		}
		if (st == SN_ST_OK && op != want)
			st = SN_ST_BAD_BLOCK;
		if (lane == 0) {
			out_len[r] = st == SN_ST_OK ? op : 0u;
			status[r] = st;
		}
		wave_mem_fence();
	}
}
