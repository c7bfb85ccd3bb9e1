/*
 * brotli_dec.hip -- brotli stream decoder for gfx950, one wave per record.
 *
 * Replaces BrotliDecoderDecompress as called per record by the reference
 * (/root/reference/lib/brotli-mt_decompress.c:344-346; one raw brotli stream per 16-byte-header
 * record, output capacity hint << 16, :236-239).  Format: RFC 7932, complete: uncompressed and
 * metadata meta-blocks, simple and complex prefix codes, block switching for the three
 * categories, literal and distance context maps (RLE + inverse move-to-front), the distance ring
 * buffer with its 16 short codes, static dictionary words with the 121 transforms.
 *
 * A brotli stream is ONE serial bitstream per record: every symbol's position depends on the
 * length of the symbol before it, so the wave decodes in wave-uniform control flow and uses its
 * lanes where the format has width:
 *   - the input is held in a register window (lane i = dword i of 256 stream bytes); the bit accumulator is refilled with v_readlane, no memory
 *     access sits on the decode chain;
 *   - prefix codes are kept in canonical form: lane l of a tree's vector holds the left-aligned
 *     end of the code range of length l, so ONE compare + ballot finds a symbol's length; the
 *     symbol itself comes from the tree's sorted-symbol array.  The trees in use (insert&copy
 *     tree of the block type, the four distance trees of the block type on four 16-lane groups,
 *     the literal tree when the meta-block has a single one) stay in registers; further trees
 *     live in LDS (first few) or in the wave's scratch in HBM;
 *   - literals collect in a register (lane = output position mod 64) and leave as coalesced
 *     stores; copies, uncompressed meta-blocks and dictionary words are moved by all 64 lanes;
 *   - tables are built lane-parallel (ballot counting sort of the code lengths).
 *
 * Constant data of the format (dictionary, transforms, context lookup): csrc/data/brotli_static.bin.
 */
#include "brotli_dec_common.h"

/* ------------------------------------------------------------------ the kernel
 * grid = min(nrec, resident waves); wave w decodes records w, w + grid, ...  out_len[r] receives the
 * decoded size, status[r] GPUMT_ST_OK / BAD_BLOCK (malformed or truncated) / SIZE_MISMATCH (does
 * not fit the record's capacity). */
#ifndef ZMT_EMU
#define BRT() (PROF ? (u64)clock64() : 0ull)
#else
#define BRT() 0ull
#endif
#define BRP(i)                                                                                     \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = BRT();                                                      \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

template <bool PROF>
static __device__ __forceinline__ void
brotli_dec_body(BrLds &L, const u8 *__restrict__ stream, const u64 *__restrict__ rec_off, const u32 *__restrict__ rec_len,
		u32 nrec, u8 *out_base, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap,
		u32 *__restrict__ out_len, u32 *__restrict__ status, u8 *__restrict__ scratch,
		const u8 *__restrict__ blob, u32 want, unsigned long long *prof)
{
	const int lane = wv_lane();
	u64 pc[PROF ? 8 : 1] = {0}, tq = BRT();
	const u64 t_begin = tq;
	(void)t_begin;
	u8 *const G = scratch + (u64)blockIdx.x * BR_WSCRATCH;
	{
		const u8 *ctx = blob + uld32(blob + 12);
		for (u32 i = (u32)lane * 4; i < 2048; i += 256)
			*(u32 *)(L.lut + i) = ld32u(ctx + i);
		if (lane < 24) {
			L.kins[lane] = (u32)BR_INS_BASE[lane] | (u32)BR_INS_BITS[lane] << 24;
			L.kcopy[lane] = (u32)BR_COPY_BASE[lane] | (u32)BR_COPY_BITS[lane] << 24;
		}
	}
	wv_sync();

	for (u32 rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
		/* second pass behind zmt_brotli_dec4_kernel: only the records it handed over (status == want) */
		if (want != 0xFFFFFFFFu && wv_readfirst(status[rec]) != want)
			continue;
		u8 *const out = out_base + out_off[rec];
		const u32 cap = wv_readfirst(out_cap[rec]);
		u32 stc = ST_OK;
		BrBits b;
		b.p = stream + rec_off[rec];
		b.n = wv_readfirst(rec_len[rec]);
		br_seek(b, 0, lane);
		u32 pos = 0;

		/* 9.1 window bits */
		u32 wbits = 16;
		if (br_get(b, 1, lane)) {
			u32 v = br_get(b, 3, lane);
			if (v) {
				wbits = 17 + v;
			} else {
				v = br_get(b, 3, lane);
				if (v == 1)
					stc = BRBAD(); /* large-window streams are not brotli-mt's */
				wbits = v ? 8 + v : 17;
			}
		}
		const u32 max_backward = (1u << wbits) - 16u;
		u32 rb0 = 16, rb1 = 15, rb2 = 11, rb3 = 4; /* rb3 = last distance */
		u32 litv = 0, lit_lo = 0;                  /* literals not stored yet: [lit_lo, pos) */
#define BR_FLUSH()                                                                                 \
	do {                                                                                       \
		if (lit_lo < pos) {                                                                \
			const u32 a_ = ((pos - 1) & ~63u) + (u32)lane;                             \
			if (a_ >= lit_lo && a_ < pos)                                              \
				out[a_] = (u8)litv;                                                \
			lit_lo = pos;                                                              \
		}                                                                                  \
	} while (0)

		/* Copies are not executed one by one: a copy costs a load -> store round trip to HBM, which
		 * would stall the serial decode for thousands of cycles per command.  Meta-blocks without
		 * literal context modelling never look at the output while decoding (positions follow
		 * from the lengths alone), so up to 64 copies wait in lane slots -- literals already go to
		 * their final positions -- and are then executed side by side: every lane performs its
		 * own copy once its source lies below the watermark W (the destination of the first
		 * unfinished copy; everything below W is complete). */
		u32 bm_pos = 0, bm_dist = 0, bm_len = 0, nbatch = 0;
#define BR_EXEC()                                                                                  \
	do {                                                                                       \
		if (nbatch) {                                                                      \
			wave_mem_fence();                                                          \
			const bool act_ = (u32)lane < nbatch;                                      \
			const u32 ml_ = act_ ? bm_len : 0;                                         \
			const u32 sp_ = bm_pos - bm_dist, eff_ = ml_ < bm_dist ? ml_ : bm_dist;    \
			bool fin_ = !act_;                                                         \
			for (;;) {                                                                 \
				const u64 unf_ = wv_ballot(!fin_);                                 \
				if (!unf_)                                                         \
					break;                                                     \
				const int fst_ = wv_ffs(unf_) - 1;                                 \
				const u32 W_ = wv_readlane(bm_pos, fst_);                          \
				const u32 hl_ = wv_readlane(ml_, fst_);                            \
				if (hl_ > BR_CAP) {                                                \
					br_wave_match(out + W_, wv_readlane(bm_dist, fst_), hl_, lane); \
					if (lane == fst_)                                          \
						fin_ = true;                                       \
				} else {                                                           \
					const bool rdy_ = !fin_ && ml_ <= BR_CAP && sp_ + eff_ <= W_; \
					if (rdy_) {                                                \
						g_match(out + bm_pos, bm_dist, ml_);               \
						fin_ = true;                                       \
					}                                                          \
				}                                                                  \
				wave_mem_fence();                                                  \
			}                                                                          \
			nbatch = 0;                                                                \
		}                                                                                  \
	} while (0)

		while (stc == ST_OK) {
			if (br_over(b)) {
				stc = BRBAD();
				break;
			}
			const u32 is_last = br_get(b, 1, lane);
			if (is_last && br_get(b, 1, lane))
				break;
			const u32 nib = br_get(b, 2, lane);
			if (nib == 3) {
				if (br_get(b, 1, lane)) {
					stc = BRBAD();
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
					stc = BRBAD();
					break;
				}
				br_seek(b, (u32)at + skip, lane);
				if (is_last)
					break;
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
					stc = BRBAD();
					break;
				}
			}
			mlen++;
			if (!is_last && br_get(b, 1, lane)) {
				/* uncompressed meta-block */
				if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane)) {
					stc = BRBAD();
					break;
				}
				const u64 at = br_used(b) >> 3;
				if (br_over(b) || at + mlen > b.n) {
					stc = BRBAD();
					break;
				}
				if (mlen > cap - pos) {
					stc = ST_SIZE_MISMATCH;
					break;
				}
				BR_FLUSH();
				BR_EXEC();
				wave_copy(out + pos, b.p + at, mlen, lane);
				wave_mem_fence();
				pos += mlen;
				lit_lo = pos;
				br_seek(b, (u32)at + mlen, lane);
				continue;
			}
			/* ---------------- compressed meta-block header (9.2) ---------------- */
			BrCat c0, c1, c2;
			bool bad = false;
#define BR_CAT_HDR(c, k)                                                                           \
	do {                                                                                       \
		c.ntypes = br_varlen8(b, lane) + 1;                                                \
		c.type = 0;                                                                        \
		c.prev = 1;                                                                        \
		c.left = 1u << 24;                                                                 \
		if (!bad && c.ntypes >= 2) {                                                       \
			if (!br_read_code(b, L, G + BR_G_BT + (2 * k) * BR_BT_STRIDE, c.ntypes + 2, true, lane) || \
			    !br_read_code(b, L, G + BR_G_BT + (2 * k + 1) * BR_BT_STRIDE, 26, true, lane))         \
				bad = true;                                                        \
			else                                                                       \
				c.left = br_block_len(b, G + BR_G_BT + (2 * k + 1) * BR_BT_STRIDE, bad, lane);  \
		}                                                                                  \
	} while (0)
			BR_CAT_HDR(c0, 0);
			BR_CAT_HDR(c1, 1);
			BR_CAT_HDR(c2, 2);
			if (bad || br_over(b)) {
				stc = BRBAD();
				break;
			}
			const u32 npostfix = br_get(b, 2, lane);
			const u32 ndirect = br_get(b, 4, lane) << npostfix;
			for (u32 i = 0; i < c0.ntypes; i++) {
				const u32 m = br_get(b, 2, lane);
				if (lane == 0)
					L.cmode[i] = (u8)m;
			}
			u32 ntl = 1, ntd = 1;
			if (!br_context_map(b, L, G + BR_G_LCMAP, 64 * c0.ntypes, G + BR_G_BT + 6 * BR_BT_STRIDE, ntl, lane) ||
			    !br_context_map(b, L, G + BR_G_DCMAP, 4 * c2.ntypes, G + BR_G_BT + 6 * BR_BT_STRIDE, ntd, lane)) {
				stc = BRBAD();
				break;
			}
			{
				/* map entries must name existing trees */
				bool e = false;
				for (u32 i = (u32)lane; i < 64 * c0.ntypes; i += 64)
					e |= G[BR_G_LCMAP + i] >= ntl;
				for (u32 i = (u32)lane; i < 4 * c2.ntypes; i += 64)
					e |= G[BR_G_DCMAP + i] >= ntd;
				if (wv_any(e)) {
					stc = BRBAD();
					break;
				}
			}
			const u32 dist_alphabet = 16 + ndirect + (48u << npostfix);
#define BR_LIT_REC(t) ((t) < BR_NLIT_LDS ? L.lit + (t) * BR_LIT_STRIDE : G + BR_G_LIT + (t) * BR_LIT_STRIDE)
#define BR_CMD_REC(t) ((t) < BR_NCMD_LDS ? L.cmd + (t) * BR_CMD_STRIDE : G + BR_G_CMD + (t) * BR_CMD_STRIDE)
#define BR_DIST_REC(t) ((t) < BR_NDIST_LDS ? L.dist + (t) * BR_DIST_STRIDE : G + BR_G_DIST + (t) * BR_DIST_STRIDE)
			for (u32 i = 0; i < ntl && !bad; i++)
				bad = !br_read_code(b, L, BR_LIT_REC(i), 256, false, lane) || br_over(b);
			for (u32 i = 0; i < c1.ntypes && !bad; i++) {
				bad = !br_read_code(b, L, BR_CMD_REC(i), 704, true, lane) || br_over(b);
				/* the decode loop wants the insert / copy length codes of an insert&copy symbol,
				 * not its number: rewrite the tree's symbol array once as
				 * insert code | copy code << 5 | "distance is the last one" << 10 (RFC 7932 section 5:
				 * 11 cells of 8 x 8 codes, code bases in units of 8 packed two bits per cell) */
				u16 *sy = (u16 *)(BR_CMD_REC(i) + 128);
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
				wave_mem_fence();
			}
			for (u32 i = 0; i < ntd && !bad; i++)
				bad = !br_read_code(b, L, BR_DIST_REC(i), dist_alphabet, true, lane) || br_over(b);
			if (bad) {
				stc = BRBAD();
				break;
			}
			/* ---------------- commands (section 10) ----------------
			 * The loop keeps its whole state in registers, wave-uniform values in SGPRs.
			 * Nothing inside the loop waits on HBM: symbols come from
			 * LDS (explicit ds reads -- a flat access would wait for every literal store still in
			 * flight, loads and stores share one in-order counter on gfx9), trees beyond the LDS
			 * budget and block switches are the (rare) exceptions. */
			const bool ctx_free = ntl == 1;
			bool hbad = false;
			const u8 *const gbt = G + BR_G_BT;
			/* vector of a tree record: explicit LDS / HBM branches, never a flat access */
#define BR_VEC(va_, vi_, ldsarr, nlds, stride, goff, t)                                            \
	do {                                                                                       \
		u64 e_ = 0;                                                                        \
		if (lane < 16) {                                                                   \
			if ((t) < (nlds))                                                          \
				e_ = *(const u64 *)((ldsarr) + (t) * (stride) + 8 * lane);         \
			else {                                                                     \
				e_ = *(const u64 *)(G + (goff) + (t) * (stride) + 8 * lane);       \
				BR_SETTLE(e_);                                                     \
			}                                                                          \
		}                                                                                  \
		va_ = (u32)e_;                                                                     \
		vi_ = (u32)(e_ >> 32);                                                             \
	} while (0)
			u32 lva, lvi; /* literal tree 0 (the only one when ctx_free) */
			BR_VEC(lva, lvi, L.lit, BR_NLIT_LDS, BR_LIT_STRIDE, BR_G_LIT, 0u);
			u32 cva, cvi; /* insert&copy tree of the block type */
			BR_VEC(cva, cvi, L.cmd, BR_NCMD_LDS, BR_CMD_STRIDE, BR_G_CMD, 0u);
			u32 dva = 0, dvi = 0; /* lane 16 k + l: vector of the tree of distance context k */
			u32 dtree_k = 0;      /* lanes 16 k ..: tree index of context k */
#define BR_LOAD_DIST()                                                                             \
	do {                                                                                       \
		const u32 k_ = (u32)lane >> 4;                                                     \
		const u32 t_ = G[BR_G_DCMAP + 4 * c2.type + k_];                                   \
		u64 e_;                                                                            \
		if (t_ < BR_NDIST_LDS)                                                             \
			e_ = *(const u64 *)(L.dist + t_ * BR_DIST_STRIDE + 8 * ((u32)lane & 15)); \
		else                                                                               \
			e_ = *(const u64 *)(G + BR_G_DIST + t_ * BR_DIST_STRIDE + 8 * ((u32)lane & 15)); \
		BR_SETTLE(e_);                                                                     \
		dva = (u32)e_;                                                                     \
		dvi = (u32)(e_ >> 32);                                                             \
		dtree_k = t_;                                                                      \
	} while (0)
#define BR_LOAD_LCMAP()                                                                            \
	do {                                                                                       \
		L.cmap_cur[lane] = G[BR_G_LCMAP + 64 * c0.type + (u32)lane];                       \
		wv_sync();                                                                         \
	} while (0)
			BR_LOAD_DIST();
			u32 p1 = 0, p2 = 0; /* the two bytes before pos (context modelling only) */
			if (!ctx_free) {
				BR_LOAD_LCMAP();
				BR_FLUSH();
				BR_EXEC();
				wave_mem_fence();
				p1 = pos > 0 ? uld8(out + pos - 1) : 0;
				p2 = pos > 1 ? uld8(out + pos - 2) : 0;
			}
			/* MLEN bytes must fit the record's capacity; the loop then only checks against MLEN */
			if (mlen > cap - pos) {
				stc = ST_SIZE_MISMATCH;
				break;
			}
			u32 left = mlen;
			BRP(0);
			while (left) {
				if (BR_RARE(c1.left == 0)) {
					br_switch(b, c1, gbt, 1, hbad, lane);
					BR_VEC(cva, cvi, L.cmd, BR_NCMD_LDS, BR_CMD_STRIDE, BR_G_CMD, c1.type);
				}
				c1.left--;
				u32 cs;
				{
					const u32 k = br_sym_index(b, cva, cvi, 0, hbad, lane);
					if (c1.type < BR_NCMD_LDS)
						cs = wv_readfirst(*(const u16 *)(L.cmd + c1.type * BR_CMD_STRIDE + 128 + 2 * k));
					else
						cs = wv_readfirst(*(const u16 *)(G + BR_G_CMD + c1.type * BR_CMD_STRIDE + 128 + 2 * k));
				}
				if (BR_RARE(hbad))
					break;
				BRP(1);
				const u32 icode = cs & 31, ccode = (cs >> 5) & 31; /* packed when the tree was read */
				const bool last_dist = (cs >> 10) & 1;
				const u32 ki = wv_readfirst(L.kins[icode]), kc = wv_readfirst(L.kcopy[ccode]);
				u32 ins = ki & 0xFFFFFFu, copy = kc & 0xFFFFFFu;
				{
					/* extra bits of both lengths, in one read when they fit (nearly always) */
					const u32 ib = ki >> 24, cb = kc >> 24;
					if (BR_OFTEN(ib + cb <= 24)) {
						const u32 x = br_get(b, ib + cb, lane);
						ins += x & ((1u << ib) - 1u);
						copy += x >> ib;
					} else {
						ins += br_get(b, ib, lane);
						copy += br_get(b, cb, lane);
					}
				}
				const u32 ins0 = ins;
				if (BR_RARE(ins > left)) {
					hbad = true;
					break;
				}
				left -= ins;
				if (ctx_free && c0.ntypes == 1) {
					/* one literal tree, one block type (what levels 0..4 write): no block switch
					 * can fall inside the run, and the run is cut where the pending-literal
					 * register fills up, so the inner loop carries no checks.  The symbol of
					 * literal i is fetched from LDS while literal i + 1 is being located. */
					u32 todo = ins;
					while (todo) {
						const u32 room = 64u - (pos & 63u);
						const u32 seg = todo < room ? todo : room;
						todo -= seg;
						u32 kprev = br_sym_index(b, lva, lvi, 0, hbad, lane);
						for (u32 i = 1; i < seg; i++) {
							const u32 syp = L.lit[128 + kprev];
							const u32 k = br_sym_index(b, lva, lvi, 0, hbad, lane);
							litv = (u32)lane == (pos & 63) ? syp : litv;
							pos++;
							kprev = k;
						}
						const u32 syl = L.lit[128 + kprev];
						litv = (u32)lane == (pos & 63) ? syl : litv;
						pos++;
						if (BR_RARE((pos & 63) == 0))
							BR_FLUSH();
					}
				} else if (ctx_free) {
					/* the symbol of literal i is fetched from LDS while literal i + 1 is being
					 * located in the bitstream */
					u32 kprev = 0xFFFFFFFFu;
					for (; ins; ins--) {
						if (c0.left == 0)
							br_switch(b, c0, gbt, 0, hbad, lane);
						c0.left--;
						u32 syp = 0;
						if (kprev != 0xFFFFFFFFu)
							syp = L.lit[128 + kprev];
						const u32 k = br_sym_index(b, lva, lvi, 0, hbad, lane);
						if (kprev != 0xFFFFFFFFu) {
							litv = (u32)lane == (pos & 63) ? syp : litv;
							pos++;
							if ((pos & 63) == 0)
								BR_FLUSH();
						}
						kprev = k;
					}
					if (kprev != 0xFFFFFFFFu) {
						const u32 syp = L.lit[128 + kprev];
						litv = (u32)lane == (pos & 63) ? syp : litv;
						pos++;
						if ((pos & 63) == 0)
							BR_FLUSH();
					}
				} else {
					for (; ins; ins--) {
						if (c0.left == 0) {
							br_switch(b, c0, gbt, 0, hbad, lane);
							if (hbad)
								break;
							BR_LOAD_LCMAP();
						}
						c0.left--;
						const u8 *lut = L.lut + ((u32)L.cmode[c0.type] << 9);
						const u32 cid = wv_readfirst((u32)lut[p1] | (u32)lut[256 + p2]);
						const u32 tr = wv_readfirst((u32)L.cmap_cur[cid]);
						u32 va, vi, sy;
						BR_VEC(va, vi, L.lit, BR_NLIT_LDS, BR_LIT_STRIDE, BR_G_LIT, tr);
						const u32 k = br_sym_index(b, va, vi, 0, hbad, lane);
						if (tr < BR_NLIT_LDS)
							sy = wv_readfirst(L.lit[tr * BR_LIT_STRIDE + 128 + k]);
						else
							sy = wv_readfirst(G[BR_G_LIT + tr * BR_LIT_STRIDE + 128 + k]);
						p2 = p1;
						p1 = sy;
						litv = (u32)lane == (pos & 63) ? sy : litv;
						pos++;
						if ((pos & 63) == 0)
							BR_FLUSH();
					}
				}
				if (hbad)
					break; /* a truncated stream shows at the end of the meta-block: windows past
						* the end read as zeros, the loop is bounded by MLEN */
				BRP(2);
				if (!left)
					break;
				/* ---- distance (section 4) ---- */
				u32 dist;
				bool push = true;
				if (last_dist) {
					dist = rb3;
					push = false;
				} else {
					if (BR_RARE(c2.left == 0)) {
						br_switch(b, c2, gbt, 2, hbad, lane);
						if (hbad)
							break;
						BR_LOAD_DIST();
					}
					c2.left--;
					const u32 dctx = copy > 4 ? 3 : copy - 2;
					const u32 k = br_sym_index(b, dva, dvi, dctx, hbad, lane);
					if (hbad)
						break;
					const u32 tr = wv_readlane(dtree_k, (int)(16 * dctx));
					u32 dc;
					if (tr < BR_NDIST_LDS)
						dc = wv_readfirst(*(const u16 *)(L.dist + tr * BR_DIST_STRIDE + 128 + 2 * k));
					else
						dc = wv_readfirst(*(const u16 *)(G + BR_G_DIST + tr * BR_DIST_STRIDE + 128 + 2 * k));
					if (dc < 16) {
						const u32 which = dc < 4 ? dc : dc < 10 ? 0 : 1;
						const u32 r = which == 0 ? rb3 : which == 1 ? rb2 : which == 2 ? rb1 : rb0;
						int del = 0;
						if (dc >= 4) {
							const u32 q = (dc - 4) % 6; /* -1 +1 -2 +2 -3 +3 */
							del = (int)(q / 2 + 1);
							if (!(q & 1))
								del = -del;
						}
						const long dd = (long)r + del;
						if (dd <= 0) {
							hbad = true;
							break;
						}
						dist = (u32)dd;
						push = dc != 0;
					} else if (dc < 16 + ndirect) {
						dist = dc - 15;
					} else {
						const u32 d = dc - ndirect - 16;
						const u32 hcode = d >> npostfix, lcode = d & ((1u << npostfix) - 1);
						const u32 nbits = 1 + (hcode >> 1);
						const u64 offset = ((2ull + (hcode & 1)) << nbits) - 4;
						const u64 dd = ((offset + br_get(b, nbits, lane)) << npostfix) + lcode + ndirect + 1;
						if (dd > 0x7FFFFFFCull) {
							hbad = true;
							break;
						}
						dist = (u32)dd;
					}
				}
				const u32 max_dist = pos < max_backward ? pos : max_backward;
				BR_FLUSH();
				BRP(3);
				if (BR_RARE(dist > max_dist)) {
					if (copy < 4 || copy > 24) {
						hbad = true;
						break;
					}
					const u32 id = dist - max_dist - 1;
					const u32 shift = ut8(BR_DICT_BITS, copy);
					const u32 widx = id & ((1u << shift) - 1), tidx = id >> shift;
					if (tidx >= 121) {
						hbad = true;
						break;
					}
					BR_EXEC();
					const u32 wn = wv_readfirst(br_dict_word(L, blob, copy, widx, tidx, out + pos, left, lane));
					if (wn == 0xFFFFFFFFu) {
						hbad = true;
						break;
					}
					if (wn == 0 && ins0 == 0) {
						hbad = true; /* a command without output: never written by an encoder */
						break;
					}
					if (!ctx_free && wn) {
						const u32 np1 = L.tmp[wn - 1], np2 = wn >= 2 ? (u32)L.tmp[wn - 2] : p1;
						p2 = wv_readfirst(np2);
						p1 = wv_readfirst(np1);
					}
					wave_mem_fence();
					pos += wn;
					lit_lo = pos;
					left -= wn;
					BRP(5);
				} else {
					if (BR_RARE(copy > left)) {
						hbad = true;
						break;
					}
					if (push) {
						rb0 = rb1;
						rb1 = rb2;
						rb2 = rb3;
						rb3 = dist;
					}
					if (ctx_free) {
						if ((u32)lane == nbatch) {
							bm_pos = pos;
							bm_dist = dist;
							bm_len = copy;
						}
						nbatch++;
						pos += copy;
						lit_lo = pos;
						left -= copy;
						if (BR_RARE(nbatch == 64)) {
							BR_EXEC();
							BRP(4);
						}
					} else {
						wave_mem_fence();
						const u8 *s = out + pos - dist;
						p1 = uld8(s + (copy - 1) % dist);
						p2 = uld8(s + (copy - 2) % dist);
						br_wave_match(out + pos, dist, copy, lane);
						wave_mem_fence();
						pos += copy;
						lit_lo = pos;
						left -= copy;
					}
				}
			}
			bad = hbad || br_over(b);
			if (bad && stc == ST_OK)
				stc = BRBAD();
			if (stc != ST_OK || is_last)
				break;
		}
		if (stc == ST_OK) {
			/* zero padding up to the byte boundary; bytes after it are ignored */
			if ((br_used(b) & 7) && br_get(b, 8 - (u32)(br_used(b) & 7), lane))
				stc = BRBAD();
			if (br_over(b))
				stc = BRBAD();
		}
		BR_FLUSH();
		BR_EXEC();
		wave_mem_fence();
		if (lane == 0) {
			status[rec] = stc;
			out_len[rec] = stc == ST_OK ? pos : 0;
		}
		BRP(6);
	}
#ifndef ZMT_EMU
	if (PROF && prof && lane == 0) {
		for (int i = 0; i < (PROF ? 7 : 1); i++)
			atomicAdd(prof + i, (unsigned long long)pc[i]);
		atomicAdd(prof + 8, (unsigned long long)(BRT() - t_begin));
		atomicAdd(prof + 9, 1ull);
	}
#endif
}

/* 4 waves per SIMD (128 VGPRs) and < 10 KiB of LDS: 16 records per CU.  One record keeps a wave busy
 * with a chain of dependent scalar and cross-lane operations, so the number of records in flight is
 * what sets the throughput. */
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
zmt_brotli_dec_kernel(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off, const u32 *__restrict__ rec_len,
		      u32 nrec, u8 *out_base, const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap,
		      u32 *__restrict__ out_len, u32 *__restrict__ status, u8 *__restrict__ scratch,
		      const u8 *__restrict__ blob, u32 want)
{
	__shared__ __attribute__((aligned(16))) BrLds L;
	brotli_dec_body<false>(L, stream, rec_off, rec_len, nrec, out_base, out_off, out_cap, out_len, status, scratch,
			       blob, want, nullptr);
}

#ifndef ZMT_EMU
/* same kernel with per-phase cycle counters (developer tool): 0 headers + tables, 1 insert&copy
 * symbols, 2 literals, 3 distances, 4 copy batches, 5 dictionary words / uncompressed, 6 rest */
extern "C" __global__ void __launch_bounds__(64)
zmt_brotli_dec_kernel_prof(const u8 *__restrict__ stream, const u64 *__restrict__ rec_off,
			   const u32 *__restrict__ rec_len, u32 nrec, u8 *out_base, const u64 *__restrict__ out_off,
			   const u32 *__restrict__ out_cap, u32 *__restrict__ out_len, u32 *__restrict__ status,
			   u8 *__restrict__ scratch, const u8 *__restrict__ blob, u32 want, unsigned long long *prof)
{
	__shared__ __attribute__((aligned(16))) BrLds L;
	brotli_dec_body<true>(L, stream, rec_off, rec_len, nrec, out_base, out_off, out_cap, out_len, status, scratch,
			      blob, want, prof);
}
#endif
