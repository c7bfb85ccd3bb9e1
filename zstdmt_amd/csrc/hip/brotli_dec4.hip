/*
 * brotli_dec4.hip -- brotli stream decoder for gfx950, FOUR records per wave ("dec4").
 *
 * Replaces BrotliDecoderDecompress as called per record by the reference
 * (/root/reference/lib/brotli-mt_decompress.c:344-346) for the streams that make up BASELINE configs[4]: what
 * the reference's compressor writes at levels 0..4 -- meta-blocks with ONE block type per category, ONE literal
 * tree (no context modelling), ONE distance tree.  Every other stream (block switching, context maps, static
 * dictionary references) is handed to the general kernel (brotli_dec.hip) by status B4_HANDOFF, so the pair
 * decodes everything RFC 7932 allows.
 *
 * Why: a brotli stream is one serial bitstream, and brotli_dec.hip walks it in wave-uniform control flow -- one
 * record per wave, 70 % of its instructions on the scalar pipe, which the 16 waves of a CU saturate
 * (profiles/r03_sq_counters.json: 31 wave-instructions per output byte).  Here a wave runs FOUR streams side by
 * side, one per 16-lane group, in lockstep at the command level: one pass of the loop decodes one insert&copy
 * command of each of the four records, so every instruction of the command and distance decode serves four
 * commands, and the literal loop runs max(insert lengths) times instead of their sum.  All per-stream state is
 * "group-uniform" vector state (every lane of a group holds the same value); the canonical prefix decode of
 * brotli_dec.hip -- lane l of a tree's vector holds the left-aligned end of the code range of length l, one compare
 * finds the length -- works unchanged on 16 lanes: the ballot's 16 bits of the group, ds_bpermute inside the
 * group.  Headers and prefix-code tables are rare (a few per record) and use all 64 lanes through the shared
 * wave-cooperative code (brotli_dec_common.h), one group at a time.
 *
 * A wave with four streams is bound by the LATENCY of its dependent chain, not by issue (first version, measured:
 * one pass of the command loop ~6 400 cycles, most of them LDS round trips -- 26.2 GB/s), so the chain carries as few
 * LDS round trips as possible:
 *   - the bit buffer of a group is a 64-bit register pair; the next dword of the stream is read from the group's
 *     256-byte LDS window a whole refill ahead of its use, so no LDS access sits between two symbols;
 *   - every tree has a 256-entry direct table (payload | code length + 1 << 12) built with the tree: a symbol whose
 *     code has at most 8 bits costs ONE LDS read; longer codes (rare) take the canonical path -- the ballot's 16 bits
 *     of the group, two ds_bpermute, the sorted-symbol array;
 *   - up to 16 copies wait in a group's 16 lanes and are executed side by side with watermark rounds, as in
 *     brotli_dec.hip; literals are stored straight to their final positions.
 */
#include "brotli_dec_common.h"

#define B4_HANDOFF 102u /* internal status: zmt_brotli_dec_kernel decodes the record afterwards (gpumt.hip) */
#define B4_WIN 256u
#define B4_WSTRIDE (B4_WIN + 16u)
#ifndef B4_NB
#define B4_NB 16u /* copies a group holds back before it executes them (one per lane) */
#endif
#ifndef B4_NG
#define B4_NG 4u /* streams per wave (16-lane groups in use); gpumt.hip sizes the grid with the same number */
#endif

enum { B4_S_HDR = 0, B4_S_DEC = 1, B4_S_FIN = 2, B4_S_DONE = 3 };
#define B4_DIST_MAX 128u /* distance alphabet this kernel takes (NPOSTFIX = NDIRECT = 0 make 64); larger ones are handed over */
#define B4_DIST_STRIDE (128u + 2u * B4_DIST_MAX)
#define B4_T_LIT 0u
#define B4_T_CMD 1u
#define B4_T_DIST 2u

struct B4Lds {
	/* scratch of the shared prefix-code reader (br_read_code: one group at a time) */
	__attribute__((aligned(8))) u8 clrec[128 + 32];
	u8 lens[704];
	u8 tmp[64];
	u32 kins[24], kcopy[24]; /* insert / copy length codes: base | extra bits << 24 */
	/* per group: tree records (16 x u64 vector + sorted symbols), direct tables, stream window */
	__attribute__((aligned(8))) u8 lit[B4_NG][BR_LIT_STRIDE];
	__attribute__((aligned(8))) u8 cmd[B4_NG][BR_CMD_STRIDE];
	__attribute__((aligned(8))) u8 dist[B4_NG][B4_DIST_STRIDE];
	u16 tab[B4_NG][3][256]; /* payload | (code length + 1) << 12; 0 = the code is longer than 8 bits */
	__attribute__((aligned(16))) u8 win[B4_NG][B4_WSTRIDE];
};

/* direct table of a tree record: entry e = the symbol whose code opens the 8 stream bits e (first bit = bit 0), if
 * that code has at most 8 bits.  With the unknown bits behind the 8 taken as zeros the canonical compare finds the
 * same length as with the real ones whenever that length is <= 8 (the bounds of such lengths are multiples of 2^7) */
static __device__ __forceinline__ void b4_build_tab(const u8 *rec, u16 *tab, bool sym16, int lane)
{
	for (u32 e = (u32)lane; e < 256u; e += 64u) {
		const u32 c = br_rev15(e);
		u32 l = 9;
		for (int k = 8; k >= 0; k--)
			if (c < (*(const u32 *)(rec + 8 * k) & 0xFFFFu))
				l = (u32)k;
		u32 ent = 0;
		if (l <= 8u) {
			const u32 a = *(const u32 *)(rec + 8 * l), i0 = *(const u32 *)(rec + 8 * l + 4);
			const u32 idx = i0 + ((c - (a >> 16)) >> (15 - l));
			const u32 sym = sym16 ? (u32) * (const u16 *)(rec + 128 + 2 * idx) : (u32)rec[128 + idx];
			ent = (sym & 0xFFFu) | (l + 1u) << 12;
		}
		tab[e] = (u16)ent;
	}
}

/* order a group's earlier global stores before its later global loads, inside group-divergent control flow (the hardware
 * fence is the wave's; the fiber harness must not wait for lanes of other groups) */
static __device__ __forceinline__ void b4_grp_fence()
{
#ifdef ZMT_EMU
	grp_sync();
#else
	wave_mem_fence();
#endif
}

/* index of the next symbol in the sorted array of the tree whose vector this group's lanes hold (lane l of the group:
 * va / vi of code length l); *len = its code length */
static __device__ __forceinline__ u32 b4_sym_index(u32 bits, u32 va, u32 vi, u32 *len, bool &bad)
{
	const u32 c = br_rev15(bits & 0x7FFFu);
	const u32 m = grp_ballot(c < (va & 0xFFFFu));
	if (!m)
		bad = true;
	const int l = m ? wv_ffs((u64)m) - 1 : 0;
	const u32 a = grp_shfl(va, l), i0 = grp_shfl(vi, l);
	*len = (u32)l;
	return i0 + ((c - (a >> 16)) >> (15 - l));
}

/* copy of `ml` bytes at distance `off` by the 16 lanes of a group (long copies; plain or overlapping) */
static __device__ __forceinline__ void b4_grp_match(u8 *d, u32 off, u32 ml, u32 l16)
{
	const u8 *s = d - off;
	if (off >= ml) {
		const u32 body = ml & ~7u;
		for (u32 i = 8u * l16; i < body; i += 128u)
			st64g(d + i, ld64u(s + i));
		for (u32 i = body + l16; i < ml; i += 16u)
			d[i] = s[i];
	} else if (off >= 16u) {
		for (u32 done = 0; done < ml; done += off) {
			const u32 n = ml - done < off ? ml - done : off;
			for (u32 i = l16; i < n; i += 16u)
				d[done + i] = s[done + i];
			b4_grp_fence();
		}
	} else {
		for (u32 i = l16; i < ml; i += 16u)
			d[i] = s[i % off];
	}
}

#ifdef B4_WPE
#define B4_OCC __attribute__((amdgpu_waves_per_eu(B4_WPE, B4_WPE)))
#else
#define B4_OCC
#endif
extern "C" __global__ void __launch_bounds__(64) B4_OCC
zmt_brotli_dec4_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off, const u32 *__restrict__ rec_len,
		       u32 nrec, u8 *out_base, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap,
		       u32 *__restrict__ out_len, u32 *__restrict__ status)
{
	__shared__ __attribute__((aligned(16))) B4Lds L;
	const int lane = wv_lane();
	const u32 grp = (u32)lane >> 4, l16 = (u32)lane & 15u;
	if (lane < 24) {
		L.kins[lane] = (u32)BR_INS_BASE[lane] | (u32)BR_INS_BITS[lane] << 24;
		L.kcopy[lane] = (u32)BR_COPY_BASE[lane] | (u32)BR_COPY_BITS[lane] << 24;
	}
	wv_sync();
	/* this group's tree records, direct tables and window */
	const u32 gi = grp < B4_NG ? grp : 0u; /* (lanes of groups not in use idle; their pointers stay valid) */
	const u8 *const lit_rec = L.lit[gi];
	const u8 *const cmd_rec = L.cmd[gi];
	const u8 *const dist_rec = L.dist[gi];
	const u16 *const tab_lit = L.tab[gi][B4_T_LIT], *const tab_cmd = L.tab[gi][B4_T_CMD], *const tab_dist = L.tab[gi][B4_T_DIST];
	u8 *const win = L.win[gi];

	const u32 rec = blockIdx.x * B4_NG + grp;
	const bool exists = grp < B4_NG && rec < nrec;
	const u8 *const sp = stream + (exists ? rec_off[rec] : 0);
	const u32 slen = exists ? rec_len[rec] : 0;
	u8 *const out = out_base + (exists ? out_off[rec] : 0);
	const u32 cap = exists ? out_cap[rec] : 0;

	/* ---- per-stream state, the same in the 16 lanes of a group ---- */
	u32 st = exists ? B4_S_HDR : B4_S_DONE;
	u32 stc = ST_OK;
	u32 bitpos = 0; /* valid in B4_S_HDR / B4_S_FIN; while decoding the position is 8 * wptr - navail */
	/* bit buffer: acc holds the next navail (33..64) bits of the stream, nextw the dword behind them (stream byte wptr),
	 * read from the window [wbyte, wbyte + 264) a refill ahead */
	u64 acc = 0;
	u32 navail = 0, wptr = 0, nextw = 0, wbyte = 0;
	bool primed = false; /* the bit buffer is set up for the meta-block being decoded */
	u32 pos = 0, left = 0;
	bool first = true, was_last = false;
	u32 max_backward = 0, npostfix = 0, ndirect = 0;
	u32 rb0 = 16, rb1 = 15, rb2 = 11, rb3 = 4; /* rb3 = last distance */
	u32 lva = 0, lvi = 0, cva = 0, cvi = 0, dva = 0, dvi = 0;
	u32 bm_pos = 0, bm_dist = 0, bm_len = 0, nbatch = 0; /* lane j of the group: pending copy j */
	if (exists && slen >= (1u << 28)) { /* 32-bit bit positions */
		stc = B4_HANDOFF;
		st = B4_S_FIN;
	}

	/* window = stream bytes [W, W + 264), W a multiple of 4 (a damaged stream may run far past its end inside a
	 * meta-block: beyond the record + most of the stream's 256-byte slack the window reads as zeros, never from memory) */
#define B4_LOAD_WIN(W)                                                                                             \
	do {                                                                                                       \
		wbyte = (W);                                                                                       \
		const u32 o_ = wbyte + 16u * l16;                                                                  \
		u64 a_ = 0, b_ = 0, c_ = 0;                                                                        \
		if (o_ + 16u <= slen + 240u) {                                                                     \
			a_ = ld64u(sp + o_);                                                                       \
			b_ = ld64u(sp + o_ + 8);                                                                   \
		}                                                                                                  \
		*(u64 *)(win + 16u * l16) = a_;                                                                    \
		*(u64 *)(win + 16u * l16 + 8) = b_;                                                                \
		if (l16 == 0) {                                                                                    \
			if (wbyte + B4_WIN + 8u <= slen + 240u)                                                    \
				c_ = ld64u(sp + wbyte + B4_WIN);                                                   \
			*(u64 *)(win + B4_WIN) = c_;                                                               \
		}                                                                                                  \
		grp_sync();                                                                                        \
	} while (0)
	/* the next MARGIN stream bytes behind nextw must be in the window */
#define B4_ENSURE(cond, MARGIN)                                                                                    \
	do {                                                                                                       \
		const bool need_ = (cond) && (wptr - wbyte > B4_WIN - (MARGIN));                                   \
		if (wv_any(need_)) { /* (every group with `cond` reloads: one memory round trip for the wave, not one per group) */ \
			if (cond)                                                                                  \
				B4_LOAD_WIN(wptr);                                                                 \
		}                                                                                                  \
	} while (0)
	/* drop n (<= 24) bits; top the buffer up from nextw when 32 or fewer are left, and read the dword behind it */
#define B4_CONSUME(n)                                                                                              \
	do {                                                                                                       \
		acc >>= (n);                                                                                       \
		navail -= (n);                                                                                     \
		const bool rf_ = navail <= 32u;                                                                    \
		acc |= rf_ ? (u64)nextw << (navail & 63u) : 0ull;                                                  \
		navail += rf_ ? 32u : 0u;                                                                          \
		wptr += rf_ ? 4u : 0u;                                                                             \
		nextw = *(const u32 *)(win + (wptr - wbyte));                                                      \
	} while (0)
	/* next symbol of a tree, not consumed yet: direct table first (codes of <= 8 bits: one LDS read), the canonical walk
	 * otherwise.  PAY = the symbol's payload, LEN = its code length */
#define B4_SYMLEN(cond, TAB, REC, VA, VI, SYM16, PAY, LEN)                                                         \
	do {                                                                                                       \
		u32 e_ = (cond) ? (u32)(TAB)[(u32)acc & 255u] : 0x1000u;                                           \
		if (wv_any((cond) && (e_ >> 12) == 0)) {                                                           \
			if ((cond) && (e_ >> 12) == 0) {                                                           \
				u32 len_;                                                                          \
				const u32 k_ = b4_sym_index((u32)acc, VA, VI, &len_, hbad);                        \
				const u32 s_ = (SYM16) ? (u32) * (const u16 *)((REC) + 128 + 2 * (k_ & 1023u))     \
						       : (u32)(REC)[128 + (k_ & 255u)];                            \
				e_ = (s_ & 0xFFFu) | (len_ + 1u) << 12;                                            \
			}                                                                                          \
		}                                                                                                  \
		PAY = e_ & 0xFFFu;                                                                                 \
		LEN = (e_ >> 12) - 1u;                                                                             \
	} while (0)
#define B4_SYMBOL(cond, TAB, REC, VA, VI, SYM16, PAY)                                                              \
	do {                                                                                                       \
		u32 l_;                                                                                            \
		B4_SYMLEN(cond, TAB, REC, VA, VI, SYM16, PAY, l_);                                                 \
		if (cond)                                                                                          \
			B4_CONSUME(l_);                                                                            \
	} while (0)
	/* pending copies of the groups with `cond`: every lane its own copy once its source lies below the watermark (the
	 * destination of the group's first unfinished copy), long ones by the group together */
#define B4_EXEC(cond)                                                                                              \
	do {                                                                                                       \
		const bool ex_ = (cond) && nbatch != 0;                                                            \
		if (wv_any(ex_)) {                                                                                 \
			wave_mem_fence();                                                                          \
			if (ex_) {                                                                                 \
				const bool act_ = l16 < nbatch;                                                    \
				const u32 ml_ = act_ ? bm_len : 0;                                                 \
				const u32 src_ = bm_pos - bm_dist, eff_ = ml_ < bm_dist ? ml_ : bm_dist;           \
				bool fin_ = !act_;                                                                 \
				for (;;) {                                                                         \
					const u32 unf_ = grp_ballot(!fin_);                                        \
					if (!unf_)                                                                 \
						break;                                                             \
					const int fst_ = wv_ffs((u64)unf_) - 1;                                    \
					const u32 W_ = grp_shfl(bm_pos, fst_), hl_ = grp_shfl(ml_, fst_);          \
					if (hl_ > BR_CAP) {                                                        \
						b4_grp_match(out + W_, grp_shfl(bm_dist, fst_), hl_, l16);         \
						if ((int)l16 == fst_)                                              \
							fin_ = true;                                               \
					} else {                                                                   \
						const bool rdy_ = !fin_ && ml_ <= BR_CAP && src_ + eff_ <= W_;     \
						if (rdy_) {                                                        \
							g_match(out + bm_pos, bm_dist, ml_);                       \
							fin_ = true;                                               \
						}                                                                  \
					}                                                                          \
					b4_grp_fence();                                                            \
				}                                                                                  \
				nbatch = 0;                                                                        \
			}                                                                                          \
		}                                                                                                  \
	} while (0)

	for (;;) {
		/* ================= headers: wave-cooperative, one group at a time ================= */
		for (u32 g = 0; g < B4_NG; g++) {
			if (wv_readlane(st, (int)(16u * g)) != B4_S_HDR)
				continue;
			const int gl = (int)(16u * g);
			const bool mine = grp == g;
			BrBits b;
			{
				const u32 r_ = blockIdx.x * B4_NG + g;
				b.p = stream + rec_off[r_];
				b.n = wv_readfirst(rec_len[r_]);
			}
			const u32 g_cap = wv_readlane(cap, gl);
			const u32 bp = wv_readlane(bitpos, gl);
			u32 g_pos = wv_readlane(pos, gl);
			u8 *const g_out = out_base + out_off[blockIdx.x * B4_NG + g];
			br_seek(b, bp >> 3, lane);
			if (bp & 7u)
				(void)br_get(b, bp & 7u, lane);
			u32 g_stc = ST_OK, g_st = B4_S_HDR;
			u32 g_left = 0, g_npostfix = 0, g_ndirect = 0, g_maxb = wv_readlane(max_backward, gl);
			bool g_last = wv_readlane((u32)was_last, gl) != 0;
			if (wv_readlane((u32)first, gl)) {
				/* 9.1 window bits */
				u32 wbits = 16;
				if (br_get(b, 1, lane)) {
					u32 v = br_get(b, 3, lane);
					if (v) {
						wbits = 17 + v;
					} else {
						v = br_get(b, 3, lane);
						if (v == 1)
							g_stc = BRBAD(); /* large-window streams are not brotli-mt's */
						wbits = v ? 8 + v : 17;
					}
				}
				g_maxb = (1u << wbits) - 16u;
			}
			while (g_stc == ST_OK && g_st == B4_S_HDR) {
				if (g_last) { /* the last meta-block is done: zero padding up to the byte boundary, nothing after counts */
					g_st = B4_S_FIN;
					break;
				}
				if (br_over(b)) {
					g_stc = BRBAD();
					break;
				}
				const u32 is_last = br_get(b, 1, lane);
				if (is_last && br_get(b, 1, lane)) { /* ISLASTEMPTY */
					g_last = true;
					continue;
				}
				const u32 nib = br_get(b, 2, lane);
				if (nib == 3) {
					/* metadata meta-block: skipped */
					if (br_get(b, 1, lane)) {
						g_stc = BRBAD();
						break;
					}
					const u32 nb = br_get(b, 2, lane);
					u32 skip = 0;
					bool e = false;
					for (u32 i = 0; i < nb; i++) {
						const u32 v = br_get(b, 8, lane);
						if (i + 1 == nb && nb > 1 && v == 0)
							e = true;
						skip |= v << (8 * i);
					}
					if (nb)
						skip++;
					if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane))
						e = true;
					const u64 at = br_used(b) >> 3;
					if (e || br_over(b) || at + skip > b.n) {
						g_stc = BRBAD();
						break;
					}
					br_seek(b, (u32)at + skip, lane);
					if (is_last)
						g_last = true;
					continue;
				}
				u32 mlen = 0;
				{
					bool e = false;
					for (u32 i = 0; i < nib + 4; i++) {
						const u32 v = br_get(b, 4, lane);
						if (i + 1 == nib + 4 && nib && v == 0)
							e = true;
						mlen |= v << (4 * i);
					}
					if (e) {
						g_stc = BRBAD();
						break;
					}
				}
				mlen++;
				if (!is_last && br_get(b, 1, lane)) {
					/* uncompressed meta-block (pending copies read nothing it writes and it reads nothing at all) */
					if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane)) {
						g_stc = BRBAD();
						break;
					}
					const u64 at = br_used(b) >> 3;
					if (br_over(b) || at + mlen > b.n) {
						g_stc = BRBAD();
						break;
					}
					if (mlen > g_cap - g_pos) {
						g_stc = ST_SIZE_MISMATCH;
						break;
					}
					wave_copy(g_out + g_pos, b.p + at, mlen, lane);
					wave_mem_fence();
					g_pos += mlen;
					br_seek(b, (u32)at + mlen, lane);
					continue;
				}
				/* ---------------- compressed meta-block header (9.2): the simple shape only ---------------- */
				bool simple = true;
				for (u32 k = 0; k < 3 && simple; k++)
					simple = br_varlen8(b, lane) == 0; /* NBLTYPES == 1: no block-switch codes follow */
				if (!simple) {
					g_stc = B4_HANDOFF;
					break;
				}
				g_npostfix = br_get(b, 2, lane);
				g_ndirect = br_get(b, 4, lane) << g_npostfix;
				(void)br_get(b, 2, lane);                          /* context mode of the one literal block type */
				if (br_varlen8(b, lane) != 0 || br_varlen8(b, lane) != 0) { /* NTREESL, NTREESD: context maps */
					g_stc = B4_HANDOFF;
					break;
				}
				if (br_over(b)) {
					g_stc = BRBAD();
					break;
				}
				const u32 dist_alphabet = 16 + g_ndirect + (48u << g_npostfix);
				if (dist_alphabet > B4_DIST_MAX) {
					g_stc = B4_HANDOFF;
					break;
				}
				u8 *const g_lit = L.lit[g], *const g_cmd = L.cmd[g], *const g_dist = L.dist[g];
				bool bad = !br_read_code(b, L, g_lit, 256, false, lane) || br_over(b);
				if (!bad) {
					bad = !br_read_code(b, L, g_cmd, 704, true, lane) || br_over(b);
					/* insert&copy symbol -> insert code | copy code << 5 | "distance is the last one" << 10
					 * (RFC 7932 section 5; as in brotli_dec.hip) */
					u16 *sy = (u16 *)(g_cmd + 128);
					for (u32 k = (u32)lane; k < 704; k += 64) {
						const u32 v = sy[k];
						if (v < 704) {
							const u32 cell = v >> 6;
							const u32 icode = (((0x298500u >> (2 * cell)) & 3u) << 3) + ((v >> 3) & 7);
							const u32 ccode = (((0x262444u >> (2 * cell)) & 3u) << 3) + (v & 7);
							sy[k] = (u16)(icode | ccode << 5 | (v < 128 ? 1u << 10 : 0u));
						}
					}
					wv_sync();
				}
				if (!bad)
					bad = !br_read_code(b, L, g_dist, dist_alphabet, true, lane) || br_over(b);
				if (bad) {
					g_stc = BRBAD();
					break;
				}
				if (mlen > g_cap - g_pos) {
					g_stc = ST_SIZE_MISMATCH;
					break;
				}
				wv_sync();
				b4_build_tab(g_lit, L.tab[g][B4_T_LIT], false, lane);
				b4_build_tab(g_cmd, L.tab[g][B4_T_CMD], true, lane);
				b4_build_tab(g_dist, L.tab[g][B4_T_DIST], true, lane);
				g_left = mlen;
				g_last = is_last != 0;
				g_st = B4_S_DEC;
			}
			if (g_stc == ST_OK && g_st == B4_S_FIN) {
				if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane))
					g_stc = BRBAD();
				if (br_over(b))
					g_stc = BRBAD();
			}
			if (g_stc != ST_OK)
				g_st = B4_S_FIN;
			const u64 used = br_used(b);
			wv_sync();
			if (mine) {
				st = g_st;
				stc = g_stc;
				bitpos = (u32)used;
				primed = false;
				pos = g_pos;
				left = g_left;
				first = false;
				was_last = g_last;
				max_backward = g_maxb;
				npostfix = g_npostfix;
				ndirect = g_ndirect;
				if (g_st == B4_S_DEC) {
					const u64 el = *(const u64 *)(lit_rec + 8u * l16);
					const u64 ec = *(const u64 *)(cmd_rec + 8u * l16);
					const u64 ed = *(const u64 *)(dist_rec + 8u * l16);
					lva = (u32)el;
					lvi = (u32)(el >> 32);
					cva = (u32)ec;
					cvi = (u32)(ec >> 32);
					dva = (u32)ed;
					dvi = (u32)(ed >> 32);
				}
			}
		}
		/* ================= finished streams: pending copies, status ================= */
		if (wv_any(st == B4_S_FIN)) {
			B4_EXEC(st == B4_S_FIN && stc == ST_OK);
			wave_mem_fence();
			if (st == B4_S_FIN) {
				if (l16 == 0) {
					status[rec] = stc;
					out_len[rec] = stc == ST_OK ? pos : 0;
				}
				st = B4_S_DONE;
			}
		}
		if (!wv_any(st != B4_S_DONE))
			break;
		/* ================= bit buffers of the groups that start a meta-block ================= */
		if (wv_any(st == B4_S_DEC && !primed)) {
			if (st == B4_S_DEC && !primed) {
				B4_LOAD_WIN((bitpos >> 3) & ~3u);
				const u32 sh = bitpos - 8u * wbyte; /* 0..31 */
				const u32 w0 = *(const u32 *)win, w1 = *(const u32 *)(win + 4);
				acc = (((u64)w1 << 32) | w0) >> sh;
				navail = 64u - sh;
				wptr = wbyte + 8u;
				nextw = *(const u32 *)(win + 8);
				primed = true;
			}
		}
		/* ================= commands (section 10): the groups in lockstep, one command each per pass ================= */
		bool moved = !wv_any(st == B4_S_DEC) || wv_any(st == B4_S_HDR || st == B4_S_FIN); /* a group left B4_S_DEC: headers / status first */
		while (!moved) {
			const bool act = st == B4_S_DEC;
			bool hbad = false;
			u32 ins = 0, copy = 0;
			bool last_dist = false;
			B4_ENSURE(act, 40u); /* a command's own fields refill at most five times */
			{
				/* ---- insert&copy symbol, the extra bits of both lengths ---- */
				u32 cs, clen;
				B4_SYMLEN(act, tab_cmd, cmd_rec, cva, cvi, true, cs, clen);
				u32 ki = 0, kc = 0;
				bool wide = false;
				if (act) {
					const u32 icode = cs & 31u, ccode = (cs >> 5) & 31u;
					last_dist = (cs >> 10) & 1u;
					ki = L.kins[icode < 24u ? icode : 0u];
					kc = L.kcopy[ccode < 24u ? ccode : 0u];
					const u32 ib = ki >> 24, cb = kc >> 24; /* at most 24 bits each */
					const u32 tot = clen + ib + cb;
					wide = tot > 32u;
					if (!wide) {
						/* symbol and both extra fields inside the low word (nearly always): one step of the bit buffer */
						const u32 x = (u32)acc >> clen;
						ins = (ki & 0xFFFFFFu) + (x & ((1u << ib) - 1u));
						copy = (kc & 0xFFFFFFu) + ((x >> ib) & ((1u << cb) - 1u));
						B4_CONSUME(tot);
					}
				}
				if (wv_any(wide)) {
					if (wide) {
						const u32 ib = ki >> 24, cb = kc >> 24;
						B4_CONSUME(clen);
						ins = (ki & 0xFFFFFFu) + ((u32)acc & ((1u << ib) - 1u));
						B4_CONSUME(ib);
						copy = (kc & 0xFFFFFFu) + ((u32)acc & ((1u << cb) - 1u));
						B4_CONSUME(cb);
					}
				}
				if (act && ins > left)
					hbad = true;
			}
			/* ---- literals: max(insert lengths) passes, a group without literals left idles ---- */
			u32 todo = (act && !hbad) ? ins : 0;
			if (act && !hbad)
				left -= ins;
			while (wv_any(todo != 0)) {
				/* two literals per pass: the second one's table entry is read behind the first one's code length; a
				 * code of more than 8 bits (rare) sends the pass through the general symbol path, one literal */
				const bool la = todo != 0;
				B4_ENSURE(la, 12u);
				const bool two = todo >= 2u;
				/* (both table reads unconditional -- a group without literals left reads an entry it does not use -- and the
				 * pair stored by the group's first two lanes in one instruction: no exec-mask region on the pass's chain) */
				const u32 e0 = (u32)tab_lit[(u32)acc & 255u];
				const u32 l0 = ((e0 >> 12) - 1u) & 15u;
				const u32 e1 = (u32)tab_lit[(u32)(acc >> l0) & 255u];
				if (wv_any(la && ((e0 >> 12) == 0 || (two && (e1 >> 12) == 0)))) {
					u32 sym;
					B4_SYMBOL(la, tab_lit, lit_rec, lva, lvi, false, sym);
					if (la) {
						if (l16 == 0)
							out[pos] = (u8)sym;
						pos++;
						todo--;
						if (hbad)
							todo = 0;
					}
				} else if (la) {
					const u32 adv = two ? 2u : 1u;
					const u32 l1 = two ? (e1 >> 12) - 1u : 0u;
					if (l16 < adv)
						out[pos + l16] = (u8)(l16 ? e1 : e0);
					pos += adv;
					todo -= adv;
					B4_CONSUME(l0 + l1);
				}
			}
			/* ---- distance (section 4), the copy goes to the group's batch ---- */
			const bool dact = act && !hbad && left != 0;
			const bool dsym = dact && !last_dist;
			/* (the window margin left by the command's and the literal loop's checks covers the distance's two refills) */
			bool handoff = false;
			u32 dc, dlen;
			B4_SYMLEN(dsym, tab_dist, dist_rec, dva, dvi, true, dc, dlen);
			if (dact) {
				u32 dist = rb3;
				bool push = false;
				if (!last_dist) {
					push = true;
					if (dc < 16u) {
						B4_CONSUME(dlen);
						const u32 which = dc < 4u ? dc : dc < 10u ? 0u : 1u;
						const u32 r = which == 0 ? rb3 : which == 1 ? rb2 : which == 2 ? rb1 : rb0;
						int del = 0;
						if (dc >= 4u) {
							const u32 q = (dc - 4u) % 6u; /* -1 +1 -2 +2 -3 +3 */
							del = (int)(q / 2u + 1u);
							if (!(q & 1u))
								del = -del;
						}
						const int dd = (int)r + del;
						if (dd <= 0)
							hbad = true;
						dist = (u32)dd;
						push = dc != 0;
					} else if (dc < 16u + ndirect) {
						B4_CONSUME(dlen);
						dist = dc - 15u;
					} else {
						/* (at most B4_DIST_MAX symbols: hcode < 48, nbits <= 24, the distance below 2^28 -- 32-bit arithmetic) */
						const u32 d = dc - ndirect - 16u;
						const u32 hcode = d >> npostfix, lcode = d & ((1u << npostfix) - 1u);
						u32 nbits = (1u + (hcode >> 1)) & 31u;
						if (nbits > 24u) {
							hbad = true;
							nbits = 0;
						}
						const u32 offset = ((2u + (hcode & 1u)) << nbits) - 4u;
						u32 xb;
						if (dlen + nbits <= 32u) { /* the symbol and its extra bits in one step of the bit buffer (nearly always) */
							xb = (u32)(acc >> dlen) & ((1u << nbits) - 1u);
							B4_CONSUME(dlen + nbits);
						} else {
							B4_CONSUME(dlen);
							xb = (u32)acc & ((1u << nbits) - 1u);
							B4_CONSUME(nbits);
						}
						dist = ((offset + xb) << npostfix) + lcode + ndirect + 1u;
					}
				}
				const u32 max_dist = pos < max_backward ? pos : max_backward;
				if (!hbad) {
					if (dist > max_dist) {
						handoff = true; /* static dictionary reference: the general kernel's */
					} else if (copy > left) {
						hbad = true;
					} else {
						if (push) {
							rb0 = rb1;
							rb1 = rb2;
							rb2 = rb3;
							rb3 = dist;
						}
						if (l16 == nbatch) {
							bm_pos = pos;
							bm_dist = dist;
							bm_len = copy;
						}
						nbatch++;
						pos += copy;
						left -= copy;
					}
				}
			}
			/* (as soon as one group's 16 slots are full every group runs what it has pending: one fence and one memory
			 * round trip for the wave instead of one per group -- the groups fill their slots a pass or two apart) */
			if (wv_any(act && nbatch == B4_NB))
				B4_EXEC(act);
			if (act) {
				const u64 used = 8ull * wptr - navail;
				if (handoff) {
					stc = B4_HANDOFF;
					st = B4_S_FIN;
				} else if (hbad || (left == 0 && used > 8ull * slen)) {
					/* (a truncated stream shows at the end of the meta-block: the loop is bounded by MLEN) */
					stc = BRBAD();
					st = B4_S_FIN;
				} else if (left == 0) {
					bitpos = (u32)used;
					st = B4_S_HDR;
				}
			}
			moved = wv_any(st != B4_S_DEC && act);
		}
	}
#undef B4_LOAD_WIN
#undef B4_ENSURE
#undef B4_CONSUME
#undef B4_SYMBOL
#undef B4_EXEC
}
