/*
 * lz4_dec_parse.hip -- token walk of the LZ4 frame decoder as a data-parallel kernel ("parse3").
 *
 * Same outputs as zmt_dec_parse_kernel of lz4_dec_split.hip (u16 token positions per block at
 * tok_base(), the output offset of every 64th sequence, blk_ntok / blk_olen with the same
 * accept / reject verdicts); part of what replaces LZ4F_decompress at
 * /root/reference/lib/lz4-mt_decompress.c:349-362.
 *
 * Where the tokens of an LZ4 block are is a pointer chase: a token's position follows from the
 * lengths of the one before.  The first design let every LANE chase through its own block (one
 * dependent LDS round trip per token, 34 KiB of LDS per wave, and only 8 waves per CU exist for
 * 8 GiB of 128 KiB chunks: a pure latency chain, 8.4 ms).  A lane-per-block walk straight from
 * global memory is worse still (a 64-line scattered load costs thousands of cycles).  Here one WAVE
 * takes one block and reads it in coalesced 1 KiB stages; the chase is broken up speculatively:
 *
 *   1. every lane owns 16 consecutive bytes of the stage, in registers.  For each of its 16
 *      positions p it computes where a chain of tokens that started at p leaves the lane's bytes:
 *      a token needs only its own byte for that (next = p + 3 + literal nibble) unless a nibble
 *      is 15, so the 16 exits are a backward recurrence E[p] = E[next] over registers (an
 *      escape value marks chains that hit a length byte);
 *   2. the true chain enters lane 0 at a known position; every lane guesses its entry, looks its
 *      exit up in E and hands it to the next lane (DPP wave_shr); repeating this until nothing
 *      changes is exact (after k rounds lanes 0..k are right) and converges in a few rounds
 *      because chains that start at different bytes merge within a token or two;
 *   3. with the true entry known each lane lists its (at most 6) tokens, length bytes are read
 *      from the staged copy in LDS only for tokens of the true chain, a prefix sum packs the
 *      positions into an LDS ring that leaves as full 128-byte lines.
 *
 * All memory traffic is coalesced, LDS use is 5 KiB per wave and the kernel is bound by
 * instruction issue (about 400 per KiB of compressed data) instead of by latency.
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define P3_CB 4096u              /* ring of staged compressed bytes per wave */
#define P3_CBM (P3_CB - 1u)
#define P3_STAGE 1024u
#define P3_TRING 512u            /* token positions buffered per wave */
#define P3_WAVES 4
#define P3_LDS_WAVE (P3_CB + P3_TRING * 2u)
#define P3_BLK_STORED 0x80000000u
#define P3_BLK_EMPTY 0xFFFFFFFFu
#define P3_ESC 0xFFu             /* exit table: chain meets a length byte inside the lane's bytes */
#define P3_XEND 0x40000000u      /* chain ended (last sequence seen) */
#define P3_XBAD 0x80000000u      /* malformed */

typedef u32 p4u __attribute__((vector_size(16)));

static __device__ __forceinline__ u64 p3_tok_base(u64 coff, u32 gb) { return ((coff / 3) & ~63ull) + 128ull * gb; }

/* byte i (0..15) of a 16-byte register window */
static __device__ __forceinline__ u32 p3_byte(u32 d0, u32 d1, u32 d2, u32 d3, u32 i)
{
	const u32 lo = (i & 4u) ? d1 : d0;
	const u32 hi = (i & 4u) ? d3 : d2;
	const u32 d = (i & 8u) ? hi : lo;
	return (d >> ((i & 3u) * 8u)) & 255u;
}

/* One token at stage coordinate x, byte by byte from the staged ring (global memory for bytes not
 * staged): the arithmetic and the verdicts of the serial decoder.  Returns the next token's
 * coordinate, or P3_XEND (last sequence: *len = its literals) / P3_XBAD. */
static __device__ __forceinline__ u32 p3_walk(const u8 *cb, const u8 *gx, u32 c_hi, u32 x, u32 cs_x, u32 *len)
{
#define P3_CBYTE(X) (((X) < c_hi && (X) + P3_CB >= c_hi) ? (u32)cb[(X)&P3_CBM] : (u32)gx[(X)])
	if (x >= cs_x)
		return P3_XBAD;
	const u32 t = P3_CBYTE(x);
	u32 lit = t >> 4, h = x + 1u;
	if (lit == 15u) {
		u32 b;
		do {
			if (h >= cs_x)
				return P3_XBAD;
			b = P3_CBYTE(h);
			h++;
			lit += b;
		} while (b == 255u);
	}
	const u32 lend = h + lit;
	if (lend > cs_x || lend < h)
		return P3_XBAD;
	if (lend == cs_x) {
		*len = lit;
		return P3_XEND;
	}
	if (lend + 2u > cs_x)
		return P3_XBAD;
	u32 ml = t & 15u, m = lend + 2u;
	if (ml == 15u) {
		u32 b;
		do {
			if (m >= cs_x)
				return P3_XBAD;
			b = P3_CBYTE(m);
			m++;
			ml += b;
		} while (b == 255u);
	}
	if (m >= cs_x)
		return P3_XBAD;
	*len = lit + ml + 4u;
	return m;
}

extern "C" __global__ void __launch_bounds__(64 * P3_WAVES)
zmt_dec_parse3_kernel(const u8 *__restrict__ stream, u64 stream_bytes,
		      const u64 *__restrict__ blk_coff, const u32 *__restrict__ blk_csize,
		      const u64 *__restrict__ nblk_ptr, u16 *__restrict__ tok, u32 *__restrict__ bidx,
		      u32 *__restrict__ blk_ntok, u32 *__restrict__ blk_olen)
{
	__shared__ __attribute__((aligned(16))) u8 lds_all[P3_WAVES * P3_LDS_WAVE];
	const int lane = wv_lane();
	const u32 wave = threadIdx.x >> 6;
	const u32 gb = blockIdx.x * P3_WAVES + wave;
	const u64 nblk = *nblk_ptr;
	if ((u64)gb >= nblk)
		return;
	const u32 cs_raw = wv_readfirst(blk_csize[gb]);
	if (cs_raw == P3_BLK_EMPTY || (cs_raw & P3_BLK_STORED)) {
		if (lane == 0) {
			blk_ntok[gb] = 0;
			blk_olen[gb] = (cs_raw == P3_BLK_EMPTY) ? 0 : (cs_raw & 0x7FFFFFFFu);
		}
		return;
	}
	u8 *const cb = lds_all + wave * P3_LDS_WAVE;
	u16 *const tring = (u16 *)(cb + P3_CB);
	const u64 coff = blk_coff[gb];
	const u32 bias = (u32)coff & 15u;              /* stage coordinate x = block position + bias */
	const u8 *const gx = stream + (coff - bias);    /* 16-byte aligned: byte x is gx[x] */
	const u32 cs_x = cs_raw + bias;
	u16 *const tk = tok + p3_tok_base(coff, gb);
	u32 *const bx = bidx + (p3_tok_base(coff, gb) >> 6);

	u32 entry = bias;     /* coordinate of the next token of the true chain */
	u32 ntok = 0, nflushed = 0, opos = 0;
	bool bad = false, ended = false;
	/* stage pipeline: stage s+1 is staged before stage s is parsed, stage s+2 is in flight */
	u32 c_hi = 0;
	p4u cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0}, pend = {0, 0, 0, 0};
#define P3_LOAD(V, S)                                                                              \
	do {                                                                                       \
		(V) = (p4u){0, 0, 0, 0};                                                           \
		if ((S)*P3_STAGE + 16u * (u32)lane < cs_x) /* <= 15 bytes past the block: stream slack */ \
			(V) = *(const p4u *)(gx + (S)*P3_STAGE + 16u * (u32)lane);                 \
	} while (0)
	P3_LOAD(cur, 0u);
	P3_LOAD(nxt, 1u);
	P3_LOAD(pend, 2u);
	*(p4u *)(cb + 16u * (u32)lane) = cur;
	*(p4u *)(cb + ((P3_STAGE + 16u * (u32)lane) & P3_CBM)) = nxt;
	c_hi = 2u * P3_STAGE;
	wv_sync();

	for (u32 s = 0; !ended && !bad; s++) {
		const u32 sbase = s * P3_STAGE;
		if (sbase >= cs_x) {
			bad = true; /* the chain ran past the block without a last sequence */
			break;
		}
		if (entry >= sbase + P3_STAGE) {
			/* nothing of the chain in this stage (a long literal run): skip the work */
		} else {
			const u32 d0 = cur[0], d1 = cur[1], d2 = cur[2], d3 = cur[3];
			const u32 segx = sbase + 16u * (u32)lane; /* coordinate of my byte 0 */
			/* ---- 1. exit table: E[p] = where the chain from p leaves my 16 bytes (minus 16) ---- */
			u32 e0 = 0, e1 = 0, e2 = 0, e3 = 0;
			ZMT_UNROLL
			for (int p = 15; p >= 0; p--) {
				const u32 dw = p < 4 ? d0 : p < 8 ? d1 : p < 12 ? d2 : d3;
				const u32 t = (dw >> ((p & 3) * 8)) & 255u;
				const u32 ll = t >> 4;
				const bool esc = ll == 15u || (t & 15u) == 15u;
				const u32 nx = (u32)p + 3u + ll;
				u32 e;
				if (p >= 13) /* compile time: p + 3 is already outside */
					e = nx - 16u;
				else
					e = nx >= 16u ? nx - 16u : p3_byte(e0, e1, e2, e3, nx);
				if (esc)
					e = P3_ESC;
				const u32 sh = (u32)(p & 3) * 8u;
				if (p < 4)
					e0 |= e << sh;
				else if (p < 8)
					e1 |= e << sh;
				else if (p < 12)
					e2 |= e << sh;
				else
					e3 |= e << sh;
			}
			/* ---- 2. the true chain through the 64 lanes: speculate, compare, repeat ---- */
			u32 g = lane == 0 ? entry - sbase : 0u; /* entry offset into my bytes; >= 16: passes over me */
			u32 xc_g = 0xFFFFFFFFu, xc_x = 0;      /* last escape resolved by walking */
			u32 x;
			for (;;) {
				if (g >= P3_XEND) {
					x = g;
				} else if (g >= 16u) {
					x = g - 16u;
				} else {
					x = p3_byte(e0, e1, e2, e3, g);
					if (x == P3_ESC) {
						if (g == xc_g) {
							x = xc_x;
						} else {
							/* walk the chain from g until it leaves my bytes */
							u32 y = segx + g, l_;
							while (y < segx + 16u)
								y = p3_walk(cb, gx, c_hi, y, cs_x, &l_);
							x = y >= P3_XEND ? y : y - (segx + 16u);
							xc_g = g;
							xc_x = x;
						}
					}
				}
				const u32 xs = wv_shr1(x, 0);
				const u32 gn = lane == 0 ? g : xs;
				const bool ch = gn != g;
				g = gn;
				if (!wv_any(ch))
					break;
			}
			/* ---- 3. my tokens ---- */
			u32 tp[6], cnt = 0, sum = 0, pre[6];
			u32 fin = 0; /* P3_XEND / P3_XBAD seen by this lane */
			{
				u32 y = g < 16u ? segx + g : 0xFFFFFFFFu;
				ZMT_UNROLL
				for (int k = 0; k < 6; k++) {
					tp[k] = 0;
					pre[k] = sum;
					if (y < segx + 16u) {
						const u32 i = y - segx;
						const u32 t = p3_byte(d0, d1, d2, d3, i);
						const u32 ll = t >> 4, mlc = t & 15u;
						u32 ny, len;
						if (ll != 15u && mlc != 15u && y + 3u + ll < cs_x) {
							ny = y + 3u + ll;
							len = ll + mlc + 4u;
						} else {
							len = 0;
							ny = p3_walk(cb, gx, c_hi, y, cs_x, &len);
						}
						if (ny != P3_XBAD) {
							tp[k] = y - bias;
							cnt = (u32)k + 1u;
							sum += len;
						}
						if (ny >= P3_XEND) {
							fin = ny;
							y = 0xFFFFFFFFu;
						} else {
							y = ny;
						}
					}
				}
			}
			const u64 finm = wv_ballot(fin != 0);
			if (finm) {
				const int f = wv_ffs(finm) - 1;
				if (lane > f) {
					cnt = 0;
					sum = 0;
				}
				if (wv_readlane(fin, f) == P3_XBAD)
					bad = true;
				else
					ended = true;
			}
			/* next stage's entry: exit of the last lane */
			{
				const u32 xl = wv_readlane(x, 63);
				if (!finm) {
					if (xl >= P3_XEND)
						bad = true; /* cannot happen: a chain end always shows up in step 3 */
					entry = sbase + P3_STAGE + xl;
				}
			}
			/* ---- pack: positions into the token ring, output offsets of every 64th token ---- */
			const u32 ci = wv_scan_incl(cnt), si = wv_scan_incl(sum);
			const u32 tbase_i = ntok + ci - cnt, obase = opos + si - sum;
			ZMT_UNROLL
			for (int k = 0; k < 6; k++) {
				if ((u32)k < cnt) {
					const u32 idx = tbase_i + (u32)k;
					tring[idx & (P3_TRING - 1u)] = (u16)tp[k];
					if ((idx & 63u) == 0)
						bx[idx >> 6] = obase + pre[k];
				}
			}
			ntok += wv_readlane(ci, 63);
			opos += wv_readlane(si, 63);
			wv_sync();
			/* full lines of 64 positions out */
			while (ntok - nflushed >= 64u) {
				tk[nflushed + (u32)lane] = tring[(nflushed + (u32)lane) & (P3_TRING - 1u)];
				nflushed += 64u;
			}
		}
		/* ---- advance the stage pipeline ---- */
		if (!ended && !bad) {
			wv_sync();
			cur = nxt;
			nxt = pend;
			*(p4u *)(cb + (((s + 2u) * P3_STAGE + 16u * (u32)lane) & P3_CBM)) = nxt;
			c_hi = (s + 3u) * P3_STAGE;
			P3_LOAD(pend, s + 3u);
			wv_sync();
		}
	}
	if (opos > ZMT_BLOCK)
		bad = true;
	if (!bad && nflushed < ntok) {
		/* the last partial line goes out whole: the list has room (tok_base) */
		tk[nflushed + (u32)lane] = tring[(nflushed + (u32)lane) & (P3_TRING - 1u)];
	}
	if (lane == 0) {
		blk_ntok[gb] = bad ? 0 : ntok;
		blk_olen[gb] = bad ? 0xFFFFFFFFu : opos;
	}
}
