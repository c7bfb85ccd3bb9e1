/*
 * lz4_dec_gather.hip -- LZ4 frame decoder, copy stage as an output-side GATHER ("gather" variant).
 *
 * Same contract as the other decoders (replaces LZ4F_decompress at
 * /root/reference/lib/lz4-mt_decompress.c:349-362 for every record of a batch); it consumes what
 * zmt_dec_frames_kernel (block table) and zmt_dec_parse_kernel (token positions) of
 * lz4_dec_split.hip produce and replaces zmt_dec_copy_kernel.
 *
 * Why a gather.  Measured on MI355X (tools/ubench/lds_ops.hip): an LDS access that is not
 * naturally aligned is replayed lane by lane -- 64 LDS cycles per wave instruction for a
 * misaligned ds_read_b32/b64, 64-127 for a write -- while an aligned dword or a single byte costs
 * 2.5-4.7.  A lane-per-sequence copy (sequence k copies its literals and its match) is made of
 * exactly such accesses, and its loops run to the longest sequence of the batch.  Here the lanes
 * own OUTPUT dwords instead: a row is 256 consecutive output bytes, lane l produces bytes
 * [R + 4l, R + 4l + 4), finds the one or two sequences covering them, reads each byte from where
 * it comes from (ds_read_u8: staged compressed bytes for literals, the LDS output window for
 * matches) and writes one aligned dword.  No divergence, no partial stores, long sequences are
 * not special, and every byte of the window is written exactly once.
 *
 *   sequence step   lane per sequence (64 at a time): fields from the staged compressed bytes
 *                   (two aligned dword pairs + v_alignbyte each), DPP prefix sum -> output
 *                   position; a 16-byte descriptor per sequence goes to an LDS table and the
 *                   sequence number to a start map (one byte per output dword: sequences are >= 4
 *                   bytes long, so at most one starts per dword).  Matches whose source is older
 *                   than the LDS window (25 % on text with an 8 KiB window) are fetched from the
 *                   already written output with one or two 8-byte global loads per sequence and
 *                   parked in an LDS staging slot, one step ahead of their use.
 *   row step        lane per output dword: start map -> descriptor of a sequence starting in the
 *                   dword (validated against the table, so the map is never cleared), DPP prefix
 *                   maximum -> the sequence covering the dword's first byte; 4 x (source address,
 *                   ds_read_u8); bytes whose source lies in the row itself are re-read in
 *                   watermark passes (the first unfinished byte always completes); overlapping
 *                   matches read the most recent finished period instead of the byte `offset`
 *                   back, so a run-length match costs no passes.
 *   drain           every four rows 1 KiB of the window leaves as 16-byte stores.
 *
 * LDS per wave: 8 KiB output window (ring), 4 KiB ring of compressed bytes, 256 descriptors,
 * 1 KiB start map, 64 staging slots = 18 KiB -> 8 waves per CU.  Whatever is not resident when a
 * byte is gathered (literals of a stored block, a source beyond window and staging) is read from
 * global memory by that lane; residency is decided per byte at gather time, so correctness never
 * depends on how far the stages run ahead.
 */
#include "lz4_common.h"
#include "lz4_frame.h"

#define GW 8192u               /* LDS output window (ring) per wave */
#define GWM (GW - 1u)
#define GCB 4096u              /* ring of staged compressed bytes */
#define GCBM (GCB - 1u)
#define GSTAGE 1024u           /* one stage = 64 lanes x 16 bytes */
#define GTAB 256u              /* sequence descriptors (16 bytes each) */
#define GMAP 1024u             /* start map entries: one byte per output dword = 4 KiB of output */
#define GSTG 64u               /* staging slots (16 bytes) for matches sourced before the window */
#define GSPAN 1536u            /* output bytes one sequence step may add (map: 256 + 2 x GSPAN < 4 KiB) */
#define GROW 256u
#define G_OFF_WIN 0u
#define G_OFF_CB (GW)
#define G_OFF_TAB (GW + GCB)
#define G_OFF_MAP (G_OFF_TAB + GTAB * 16u)
#define G_OFF_STG (G_OFF_MAP + GMAP)
#define G_LDS (G_OFF_STG + GSTG * 16u)

#define DM_STG 0x10000u        /* match bytes come from staging slot (dm >> 20) */
#define DM_OVL 0x20000u        /* offset < length and offset < GROW: periodic source */
#define DM_FARG 0x40000u       /* source before the window, not staged: global memory */

#define G_BLK_STORED 0x80000000u
#define G_BLK_EMPTY 0xFFFFFFFFu

typedef u32 g4u __attribute__((vector_size(16)));


static __device__ __forceinline__ u64 g_tok_base(u64 coff, u32 gb) { return ((coff / 3) & ~63ull) + 128ull * gb; }

static __device__ __forceinline__ void g_st64(u8 *p, u64 v) { __builtin_memcpy(p, &v, 8); }

/* four bytes of the compressed ring at record coordinate x: two aligned dword reads + funnel shift */
static __device__ __forceinline__ u32 g_cb32(const u8 *lds, u32 x)
{
	const u32 a0 = *(const u32 *)(lds + G_OFF_CB + (x & GCBM & ~3u));
	const u32 a1 = *(const u32 *)(lds + G_OFF_CB + ((x + 4u) & GCBM & ~3u));
	return wv_alignbyte(a1, a0, x & 3u);
}

#ifndef ZMT_EMU
#define GKT() (PROF ? (u64)clock64() : 0ull)
#else
#define GKT() 0ull
#endif
/* per-phase cycle counters of the profiling build (developer tool, tools/dec_prof.py) */
#define GPC(i)                                                                                     \
	do {                                                                                       \
		if (PROF) {                                                                        \
			const u64 t_ = GKT();                                                      \
			pc[PROF ? (i) : 0] += t_ - tq;                                             \
			tq = t_;                                                                   \
		}                                                                                  \
	} while (0)

template <bool PROF>
static __device__ __forceinline__ void
gather_body(const u8 *__restrict__ stream, u64 stream_bytes, u32 nrec, u8 *out_base,
	    const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
	    const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
	    const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
	    const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
	    const u32 *__restrict__ blk_ntok, const u32 *__restrict__ blk_olen,
	    u32 *__restrict__ status, unsigned long long *prof, u8 *lds)
{
	const int lane = wv_lane();
	u64 pc[PROF ? 14 : 1] = {0}, tq = GKT();
	const u64 t_begin = tq;
	const u32 rec = blockIdx.x;
	if (rec >= nrec)
		return;
	if (wv_readfirst(status[rec]) != ST_OK)
		return;
	u8 *const out = out_base + out_off[rec];
	const u32 cap = out_len[rec];
	const u64 b0 = blk0[rec];
	const u32 nb = wv_readfirst(rec_nblk[rec]);
	const bool indep = wv_readfirst(rec_flags[rec]) & 1;
	u32 stc = ST_OK;
	if (nb == 0) {
		if (cap != 0 && lane == 0)
			status[rec] = ST_SIZE_MISMATCH;
		return;
	}
	/* record coordinates: byte x of the compressed side is stream[origin + x] */
	const u64 origin = blk_coff[b0] & ~15ull;
	const u8 *const gsrc = stream + origin;
	const u32 rec_cend = (u32)(blk_coff[b0 + nb - 1] - origin) +
			     (wv_readfirst(blk_csize[b0 + nb - 1]) & 0x7FFFFFFFu);

	/* ---- compressed-byte ring: stages of 1 KiB, the next one in flight in registers ---- */
	u32 c_lo = 0, c_hi = 0, c_next = 0; /* ring holds [max(c_lo, c_hi - GCB), c_hi) */
	g4u pend = {0, 0, 0, 0};
#define G_STAGE_ISSUE()                                                                            \
	do {                                                                                       \
		pend = (g4u){0, 0, 0, 0};                                                          \
		if (c_next + 16u * (u32)lane < rec_cend) /* <= 15 bytes past the record: stream slack */ \
			pend = *(const g4u *)(gsrc + c_next + 16u * (u32)lane);                    \
	} while (0)
#define G_STAGE_COMMIT()                                                                           \
	do {                                                                                       \
		*(g4u *)(lds + G_OFF_CB + ((c_next + 16u * (u32)lane) & GCBM)) = pend;             \
		c_hi = c_next + GSTAGE;                                                            \
		c_next += GSTAGE;                                                                  \
		G_STAGE_ISSUE();                                                                   \
	} while (0)
	G_STAGE_ISSUE();

	/* ---- decode state ---- */
	u32 opos_dec = 0;              /* output bytes covered by the descriptors written so far */
	u32 tab_lo = 0, tab_tail = 0;  /* live sequence numbers [tab_lo, tab_tail) */
	u32 carry_rel = 0;             /* 1: sequence tab_lo covers the byte before the next row */
	u32 R = 0;                     /* next row */
	u32 drained = 0, fenced = 0;   /* output bytes stored / stored and visible to this wave's loads */
	u32 stg_tail = 0, stg_f1 = 0, stg_f2 = 0, stg_f3 = 0; /* slots allocated before the step three back are free */
	/* block cursor */
	u32 bi = 0, t0 = 0, ntok = 0, bbase = 0, bend = 0, bstart = 0, bolen = 0, low = 0;
	bool bstored = false, bopen = false;
	const u16 *tk = tok;
	u32 qrel_next = 0;
	/* staging data of the step whose rows have not run yet */
	u64 pf0 = 0, pf1 = 0;
	u32 pslot = 0;
	bool pstaged = false;

	bool more = true;
	for (;;) {
		u64 nf0 = 0, nf1 = 0;
		u32 nslot = 0;
		bool nstaged = false;
		const u32 before = opos_dec; /* sequences below `before` have their staging data in pf* or in LDS */
		bool did = false;
		/* ================= sequence step: up to 64 tokens of the current block ================= */
		while (more && !did && stc == ST_OK) {
			/* start map: one slot per output dword modulo 4 KiB, so a sequence may only be added
			 * while its start lies within 4 KiB of the oldest row not gathered yet (a long
			 * sequence ahead of it stalls the step until the rows have caught up) */
			if (opos_dec >= R + 4092u)
				break;
			if (!bopen) {
				if (bi >= nb) {
					more = false;
					break;
				}
				const u32 gb = (u32)(b0 + bi);
				const u32 csr = wv_readfirst(blk_csize[gb]);
				const u64 coff = blk_coff[gb];
				bolen = wv_readfirst(blk_olen[gb]);
				if (csr == G_BLK_EMPTY || bolen == 0xFFFFFFFFu || cap - opos_dec < bolen) {
					stc = ST_BAD_BLOCK;
					break;
				}
				bstored = (csr & G_BLK_STORED) != 0;
				bbase = (u32)(coff - origin);
				bend = bbase + (csr & 0x7FFFFFFFu);
				bstart = opos_dec;
				low = indep ? bstart : 0;
				ntok = bstored ? 0 : wv_readfirst(blk_ntok[gb]);
				tk = tok + g_tok_base(coff, gb);
				t0 = 0;
				bopen = true;
				qrel_next = ((u32)lane < ntok) ? (u32)tk[lane] : 0u;
				if (bstored) {
					/* a stored block is one literal run read from global memory at gather time */
					const u32 bsz = bend - bbase;
					if (lane == 0) {
						*(g4u *)(lds + G_OFF_TAB + (tab_tail & (GTAB - 1u)) * 16u) =
							(g4u){opos_dec, opos_dec + bsz, bbase - opos_dec, 0u};
						lds[G_OFF_MAP + ((opos_dec >> 2) & (GMAP - 1u))] = (u8)tab_tail;
					}
					wv_sync();
					tab_tail++;
					opos_dec += bsz;
					bopen = false;
					bi++;
					did = true;
					/* the ring skips the stored bytes */
					if (bend > c_hi + GCB) {
						c_next = bend & ~(GSTAGE - 1u);
						c_lo = c_hi = c_next;
						G_STAGE_ISSUE();
					}
					break;
				}
				if (ntok == 0) { /* cannot happen for a block the parse kernel accepted */
					stc = ST_BAD_BLOCK;
					break;
				}
			}
			const u32 k = ntok - t0 < 64u ? ntok - t0 : 64u;
			const bool act = (u32)lane < k;
			const u32 q = bbase + qrel_next;
			/* stage the compressed bytes this step reads (tokens ascend: the last one bounds them) */
			{
				u32 need = wv_readlane(q, (int)(k - 1u)) + 40u;
				if (need > rec_cend)
					need = rec_cend;
				if (c_hi < need && c_next + GCB < wv_readlane(q, 0)) {
					/* far behind (after stored blocks): restart the ring at the first token */
					c_next = wv_readlane(q, 0) & ~(GSTAGE - 1u);
					c_lo = c_hi = c_next;
					G_STAGE_ISSUE();
				}
				bool wrote = false;
				while (c_hi < need) {
					wv_sync();
					G_STAGE_COMMIT();
					wrote = true;
				}
				if (wrote)
					wv_sync();
			}
			const u32 clo = (c_hi - c_lo > GCB) ? c_hi - GCB : c_lo;
			const u32 cspan = c_hi - clo;
			/* ---- fields ---- */
			u32 lit, ml, off, lsrc;
			bool is_last;
			{
				const bool res1 = (q - clo) < cspan && (q + 4u - clo) <= cspan;
				const u32 w = g_cb32(lds, q);
				const u32 tokb = w & 255u;
				const bool lx = (tokb >> 4) == 15u;
				const u32 b1 = (w >> 8) & 255u;
				lit = (tokb >> 4) + (lx ? b1 : 0u);
				lsrc = q + 1u + (lx ? 1u : 0u);
				const u32 lend = lsrc + lit;
				is_last = lend == bend;
				const bool res2 = (lend - clo) < cspan && (lend + 4u - clo) <= cspan;
				const u32 w2 = g_cb32(lds, lend);
				const bool mx = (tokb & 15u) == 15u;
				const u32 b2 = (w2 >> 16) & 255u;
				off = w2 & 0xFFFFu;
				ml = (tokb & 15u) + 4u + (mx ? b2 : 0u);
				const bool fast = res1 && !(lx && b1 == 255u) && lend <= bend &&
						  (is_last || (res2 && !(mx && b2 == 255u)));
				if (is_last) {
					ml = 0;
					off = 0;
				}
				if (act && !fast) {
					/* generic: byte by byte, ring where resident, else global memory (the
					 * parse kernel validated the chain against the block end) */
#define G_CBYTE(X) (((X)-clo) < cspan ? (u32)lds[G_OFF_CB + ((X)&GCBM)] : (u32)gsrc[(X)])
					u32 h = q;
					const u32 t = G_CBYTE(h);
					h++;
					u32 l2 = t >> 4;
					if (l2 == 15u) {
						u32 b;
						do {
							b = G_CBYTE(h);
							h++;
							l2 += b;
						} while (b == 255u && h < bend);
					}
					lit = l2;
					lsrc = h;
					const u32 le = h + l2;
					if (le >= bend) {
						is_last = true;
						ml = 0;
						off = 0;
					} else {
						is_last = false;
						off = G_CBYTE(le) | (G_CBYTE(le + 1u) << 8);
						u32 m = le + 2u;
						ml = t & 15u;
						if (ml == 15u) {
							u32 b;
							do {
								b = G_CBYTE(m);
								m++;
								ml += b;
							} while (b == 255u && m < bend);
						}
						ml += 4u;
					}
				}
			}
			const u32 len = act ? lit + ml : 0u;
			const u32 incl = wv_scan_incl(len);
			u32 take = k;
			{
				const u64 om = wv_ballot(act && (incl > GSPAN || opos_dec + incl - len >= R + 4092u));
				if (om) {
					take = (u32)wv_ffs(om) - 1u;
					if (take == 0)
						take = 1;
				}
			}
			const bool a2 = (u32)lane < take;
			const u32 total = wv_readlane(incl, (int)(take - 1u));
			const u32 op = opos_dec + incl - len;
			const u32 mpos = op + lit;
			if (wv_any(a2 && !is_last && (off == 0 || off > mpos - low)) ||
			    total > cap - opos_dec || opos_dec + total - bstart > ZMT_BLOCK) {
				stc = ST_BAD_BLOCK;
				break;
			}
			const bool nz = a2 && len != 0;
			const u64 nzm = wv_ballot(nz);
			const u32 sn = tab_tail + wv_mbcnt(nzm);
			u32 dm = 0;
			const bool hasm = nz && ml != 0;
			const bool farm = hasm && off > GW - GROW;
			if (hasm && !farm)
				dm = off | ((off < ml && off < GROW) ? DM_OVL : 0u);
			/* ---- matches sourced before the window: park the bytes in a staging slot ---- */
			{
				const u32 src_end = mpos - off + ml;
				const bool cand = farm && ml <= 16u;
				if (wv_any(cand && src_end > fenced)) {
					wave_mem_fence();
					fenced = drained;
					if (PROF)
						pc[PROF ? 13 : 0]++;
				}
				const u64 cm = wv_ballot(cand && src_end <= fenced);
				const u32 avail = GSTG - (stg_tail - stg_f3);
				const u32 fr = wv_mbcnt(cm);
				const bool stg = cand && src_end <= fenced && fr < avail;
				if (stg) {
					const u8 *g = out + (mpos - off);
					nslot = (stg_tail + fr) & (GSTG - 1u);
					nf0 = ld64u(g);
					nf1 = ml > 8u ? ld64u(g + 8) : 0ull;
					nstaged = true;
					dm = off | DM_STG | (nslot << 20);
				} else if (farm) {
					dm = off | DM_FARG;
				}
				const u32 nc = (u32)wv_popc(cm);
				stg_tail += nc < avail ? nc : avail;
			}
			if (nz) {
				*(g4u *)(lds + G_OFF_TAB + (sn & (GTAB - 1u)) * 16u) = (g4u){op, mpos, lsrc - op, dm};
				lds[G_OFF_MAP + ((op >> 2) & (GMAP - 1u))] = (u8)sn;
			}
			wv_sync();
			tab_tail += (u32)wv_popc(nzm);
			opos_dec += total;
			t0 += take;
			did = true;
			if (t0 >= ntok) {
				if (opos_dec - bstart != bolen) {
					stc = ST_BAD_BLOCK;
					break;
				}
				bopen = false;
				bi++;
			} else {
				qrel_next = (t0 + (u32)lane < ntok) ? (u32)tk[t0 + lane] : 0u;
			}
		}
		if (stc != ST_OK)
			break;
		GPC(0);
		if (PROF && did)
			pc[PROF ? 11 : 0]++;
		/* ================= commit the staging data of the previous step ================= */
		if (wv_any(pstaged)) {
			if (pstaged) {
				*(u64 *)(lds + G_OFF_STG + pslot * 16u) = pf0;
				*(u64 *)(lds + G_OFF_STG + pslot * 16u + 8u) = pf1;
			}
			wv_sync();
		}
		GPC(1);
		const u32 lim = did ? before : opos_dec; /* rows may use sequences below lim */
		const bool final = !did && !more;
		/* ================= rows ================= */
		while (R < lim && (R + GROW <= lim || final)) {
			const u32 P = R + 4u * (u32)lane;
			const u32 clo = (c_hi - c_lo > GCB) ? c_hi - GCB : c_lo;
			const u32 cspan = c_hi - clo;
			const u32 mm = lds[G_OFF_MAP + ((P >> 2) & (GMAP - 1u))];
			const u32 sc = tab_lo + ((mm - tab_lo) & 255u);
			const g4u eB = *(const g4u *)(lds + G_OFF_TAB + (sc & (GTAB - 1u)) * 16u);
			const bool valid = (sc - tab_lo) < (tab_tail - tab_lo) && (eB[0] >> 2) == (P >> 2) && eB[0] < lim;
			const u32 v = valid ? sc - tab_lo + 1u : 0u;
			const u32 inc = wv_scan_max_incl(v);
			const u32 exc = wv_shr1(inc, 0);
			const u32 arel = exc > carry_rel ? exc : carry_rel;
			const g4u eA = *(const g4u *)(lds + G_OFF_TAB + ((tab_lo + arel - 1u) & (GTAB - 1u)) * 16u);
			const u32 r = valid ? (eB[0] & 3u) : 4u;
			GPC(2);
			if (PROF)
				pc[PROF ? 10 : 0]++;
			u32 wd = 0, dmask = 0;
			u32 ad[4], qs[4];
			bool whole = false;
			/* a row inside one non-resident literal run (stored block): dword loads */
			if (!wv_any(valid)) {
				const u32 x0 = P + eA[2];
				const bool lres = (x0 - clo) < cspan;
				if (wv_all(P + 4u <= eA[1] && P + 4u <= lim && !lres)) {
					wd = ld32u(gsrc + x0);
					whole = true;
				}
			}
			if (!whole) {
				u32 gl = 0, gx[4];
				ZMT_UNROLL
				for (u32 j = 0; j < 4; j++) {
					const u32 p = P + j;
					const bool useB = j >= r;
					const u32 e_mpos = useB ? eB[1] : eA[1], e_dl = useB ? eB[2] : eA[2];
					const u32 e_dm = useB ? eB[3] : eA[3];
					const bool pv = p < lim;
					const bool isl = p < e_mpos;
					const u32 x = p + e_dl;
					const bool lres = (x - clo) < cspan;
					const u32 off = e_dm & 0xFFFFu;
					u32 qq = p - off;
					if (wv_any(pv && !isl && (e_dm & DM_OVL))) {
						if (!isl && (e_dm & DM_OVL)) {
							if (R > e_mpos)
								qq = p - off * ((p - R + off) / off); /* newest finished period */
							else
								qq = e_mpos - off + (p - e_mpos) % off; /* first period */
						}
					}
					u32 a;
					if (isl)
						a = G_OFF_CB + (x & GCBM);
					else if (e_dm & DM_STG)
						a = G_OFF_STG + (e_dm >> 20) * 16u + (p - e_mpos);
					else
						a = G_OFF_WIN + (qq & GWM);
					const bool glob = pv && (isl ? !lres : (e_dm & DM_FARG) != 0);
					const bool dirty = pv && !isl && !(e_dm & (DM_STG | DM_FARG)) && qq >= R;
					ad[j] = a;
					qs[j] = qq;
					gx[j] = isl ? x : qq;
					gl |= glob ? (1u << j) | (isl ? 16u << j : 0u) : 0u;
					dmask |= dirty ? 1u << j : 0u;
					const u32 b = (pv && !glob) ? (u32)lds[a] : 0u;
					wd |= b << (8u * j);
				}
				GPC(3);
				if (wv_any(gl != 0)) {
					bool needf = false;
					ZMT_UNROLL
					for (u32 j = 0; j < 4; j++)
						needf |= ((gl >> j) & 1u) && !((gl >> (4u + j)) & 1u) && gx[j] >= fenced;
					if (wv_any(needf)) {
						wave_mem_fence();
						fenced = drained;
					}
					ZMT_UNROLL
					for (u32 j = 0; j < 4; j++) {
						if ((gl >> j) & 1u) {
							const u32 b = ((gl >> (4u + j)) & 1u) ? (u32)gsrc[gx[j]] : (u32)out[gx[j]];
							wd |= b << (8u * j);
						}
					}
				}
			}
			GPC(4);
			if (P < lim)
				*(u32 *)(lds + G_OFF_WIN + (P & GWM)) = wd;
			wv_sync();
			GPC(5);
			/* bytes sourced inside this row: watermark passes */
			for (;;) {
				const u64 dl = wv_ballot(dmask != 0);
				if (!dl)
					break;
				if (PROF)
					pc[PROF ? 12 : 0]++;
				const int fl = wv_ffs(dl) - 1;
				const u32 fpos = P + (u32)__builtin_ctz(dmask | 16u);
				const u32 F = wv_readlane(fpos, fl);
				const bool touch = dmask != 0;
				ZMT_UNROLL
				for (u32 j = 0; j < 4; j++) {
					if ((dmask >> j) & 1u) {
						const u32 b = lds[ad[j]];
						wd = (wd & ~(0xFFu << (8u * j))) | (b << (8u * j));
						if (qs[j] < F)
							dmask &= ~(1u << j);
					}
				}
				if (touch)
					*(u32 *)(lds + G_OFF_WIN + (P & GWM)) = wd;
				wv_sync();
			}
			GPC(6);
			/* carry: the last sequence that starts in or before this row */
			{
				const u32 last = wv_readlane(inc, 63);
				const u32 mx = last > carry_rel ? last : carry_rel;
				if (mx) {
					tab_lo += mx - 1u;
					carry_rel = 1;
				}
			}
			R += GROW;
			if ((R & 1023u) == 0 && R <= lim) {
				const u32 D = R - 1024u;
				const g4u v4 = *(const g4u *)(lds + G_OFF_WIN + ((D + 16u * (u32)lane) & GWM));
				*(g4u *)(out + D + 16u * (u32)lane) = v4;
				drained = R;
			}
			GPC(7);
		}
		if (final)
			break;
		/* rotate the pipeline */
		pf0 = nf0;
		pf1 = nf1;
		pslot = nslot;
		pstaged = nstaged;
		if (did) {
			stg_f3 = stg_f2;
			stg_f2 = stg_f1;
			stg_f1 = stg_tail;
		}
	}
	if (stc == ST_OK) {
		/* rest of the window (the last row may be partial) */
		const u32 total = opos_dec;
		for (u32 D = drained; D < total; D += 1024u) {
			const u32 o = D + 16u * (u32)lane;
			if (o + 16u <= total) {
				const g4u v4 = *(const g4u *)(lds + G_OFF_WIN + (o & GWM));
				*(g4u *)(out + o) = v4;
			} else if (o < total) {
				for (u32 i = o; i < total; i++)
					out[i] = lds[G_OFF_WIN + (i & GWM)];
			}
		}
		if (total != cap)
			stc = ST_SIZE_MISMATCH;
	}
	if (lane == 0 && stc != ST_OK)
		status[rec] = stc;
#ifndef ZMT_EMU
	if (PROF && prof && lane == 0) {
		for (int i = 0; i < 8; i++)
			atomicAdd(prof + i, (unsigned long long)pc[PROF ? i : 0]);
		atomicAdd(prof + 8, (unsigned long long)(GKT() - t_begin));
		atomicAdd(prof + 9, 1ull);
		for (int i = 10; i < 14; i++)
			atomicAdd(prof + i, (unsigned long long)pc[PROF ? i : 0]);
	}
#endif
}

extern "C" __global__ void __launch_bounds__(64)
zmt_dec_gather_kernel(const u8 *__restrict__ stream, u64 stream_bytes, u32 nrec, u8 *out_base,
		      const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
		      const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
		      const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
		      const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
		      const u32 *__restrict__ blk_ntok, const u32 *__restrict__ blk_olen,
		      u32 *__restrict__ status)
{
	__shared__ __attribute__((aligned(16))) u8 lds[G_LDS];
	gather_body<false>(stream, stream_bytes, nrec, out_base, out_off, out_len, blk0, blk_coff, blk_csize,
			   rec_nblk, rec_flags, tok, blk_ntok, blk_olen, status, nullptr, lds);
}

#ifndef ZMT_EMU
extern "C" __global__ void __launch_bounds__(64)
zmt_dec_gather_kernel_prof(const u8 *__restrict__ stream, u64 stream_bytes, u32 nrec, u8 *out_base,
			   const u64 *__restrict__ out_off, const u32 *__restrict__ out_len,
			   const u64 *__restrict__ blk0, const u64 *__restrict__ blk_coff,
			   const u32 *__restrict__ blk_csize, const u32 *__restrict__ rec_nblk,
			   const u32 *__restrict__ rec_flags, const u16 *__restrict__ tok,
			   const u32 *__restrict__ blk_ntok, const u32 *__restrict__ blk_olen,
			   u32 *__restrict__ status, unsigned long long *prof)
{
	__shared__ __attribute__((aligned(16))) u8 lds[G_LDS];
	gather_body<true>(stream, stream_bytes, nrec, out_base, out_off, out_len, blk0, blk_coff, blk_csize,
			  rec_nblk, rec_flags, tok, blk_ntok, blk_olen, status, prof, lds);
}
#endif
