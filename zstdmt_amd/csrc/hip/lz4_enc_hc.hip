/*
 * lz4_enc_hc.hip -- LZ4 frame encoder for the "HC" levels 3..12 of lz4-mt, bit-exact.
 *
 * The reference hands levels >= 3 to LZ4F_compressFrame's HC path (prefs.compressionLevel,
 * /root/reference/lib/lz4-mt_compress.c:141-146, call :281; the CLI default is level 3,
 * /root/reference/programs/lz4-mt.c:19).  For levels 3..9 that is liblz4's hash-chain parser: every
 * position enters a 32 K-entry hash table + 64 K-entry chain of 16-bit deltas, a search walks at
 * most 4 / 8 / 16 / 32 / 64 / 128 / 256 chain links (level 9 also jumps over runs of one byte value:
 * the "pattern analysis"), and a lazy evaluation over up to three overlapping matches decides what
 * is emitted (oracle/lz4hc_oracle.c restates it and is pinned against the reference build).  Levels
 * 10..12 are liblz4's optimal parser on the same chains (96 / 512 / 16 384 links, chain swap and
 * pattern analysis on): a table of the cheapest way to reach each of the next <= 4 096 positions, filled
 * with a search wherever one can pay and read backwards.
 *
 * The parse is inherently serial per chunk (what is found depends on everything inserted before),
 * so the unit of parallelism is the chunk: one wave per chunk on a persistent grid, tables in a
 * 256 KiB global scratch area per wave.  The wave's lanes are used where the algorithm has width:
 *   - insertion of up to 64 consecutive positions at once: hashes in parallel, positions that share
 *     a hash inside the batch are chained to each other in lane order exactly as the serial loop
 *     would (ballot per distinct hash), the last one of each hash updates the table;
 *   - match length and backward extension compare 64 bytes per step (ballot + ffs);
 *   - literals and stored blocks are copied by the whole wave.
 * Control flow (chain walk, lazy evaluation) is wave-uniform.  Throughput is far below the level-1
 * encoder and below the host library (dependent global-memory round trips per chain link); it
 * exists so that the default level of the reference CLI is served on the device, bit for bit.
 */
#include "lz4_common.h"

#define HC_HASH_LOG 15
#define HC_MAXD 65536u
#define HC_DIST_MAX 65535u
#define HC_BASE 65536u
#define HC_MINMATCH 4u
#define HC_MFLIMIT 12u
#define HC_LASTLITERALS 5u
#define HC_OPTIMAL_ML 18
#define HC_OPT_NUM 4096
#define HC_TRAILING 3
#define HC_OPT_ROWS (HC_OPT_NUM + HC_TRAILING + 5) /* entries of one column of the optimal parser's table */
#define HC_SCRATCH (32768u * 4u + HC_MAXD * 2u + 4u * HC_OPT_ROWS * 4u + 128u) /* == GPUMT_LZ4HC_SCRATCH */

static_assert(HC_SCRATCH == 327936u, "keep GPUMT_LZ4HC_SCRATCH (include/gpumt.h) in step");

struct HcState {
	u32 *hash;
	u16 *chain;
	const u8 *src;
	u32 next_to_update;
	u32 *filter; /* 64 words of LDS: two folded bitmaps of the window scan (hc_win_build) */
	/* levels 10..12: the price table of the optimal parser, one column per field */
	int *o_price, *o_off, *o_mlen, *o_litlen;
};

static __device__ __forceinline__ u32 hc_hash(u32 v) { return (v * 2654435761u) >> (32 - HC_HASH_LOG); }

/* wave-uniform loads of table entries (every lane issues the same address) */
static __device__ __forceinline__ u32 hc_uld_hash(const HcState &H, u32 hv) { return wv_readfirst(H.hash[hv]); }
static __device__ __forceinline__ u32 hc_uld_chain(const HcState &H, u32 idx)
{
	return wv_readfirst((u32)H.chain[idx & (HC_MAXD - 1)]);
}

/*
 * LZ4HC_Insert + the lookup that follows it in LZ4HC_InsertAndGetWiderMatch: positions
 * [next_to_update, target) enter the chains, 64 at a time, and the search at `target` gets its four
 * bytes (*pattern) and the head of its chain (*head = hashTable[hash(pattern)] after the insertions).
 * The lookup rides on the last batch: a spare lane loads the bytes at `target` with the batch's source
 * loads and its table entry with the batch's gather; an inserted position with the same hash overrides
 * the entry exactly as the store the serial code would have read back.  Two memory round trips instead
 * of four in front of every search.
 */
static __device__ void hc_insert_lookup(HcState &H, u32 target, u32 *pattern, u32 *head, int lane)
{
	for (;;) {
		const u32 base = H.next_to_update;
		const u32 left = target > base ? target - base : 0u;
		const u32 cnt = left < 64u ? left : 64u;
		const bool fin = left < 64u;                 /* the last batch: lane `cnt` is free for the lookup */
		const bool act = (u32)lane < cnt;
		const bool probe = fin && (u32)lane == cnt;
		const u32 idx = probe ? target : base + (u32)lane;
		const u32 w = (act || probe) ? ld32u(H.src + (idx - HC_BASE)) : 0u;
		const u32 hv = (act || probe) ? hc_hash(w) : 0x10000u + (u32)lane;
		const u32 prev = (act || probe) ? H.hash[hv] : 0u;
		/* lanes of the batch that share a hash: chained in lane order, the last inserting one owns the table.  Two folded LDS
		 * bitmaps find the lanes that may share one (a bit set twice), an exact pass runs over those -- two or three of a
		 * batch, where a pass over every distinct hash took 64 rounds (round 6) */
		u32 pred = 64;
		bool last = true;
		{
			/* (one 2 048-bit filter on the hash's low 11 bits: the returning atomic names the lanes that found their bit taken --
			 * about one of a batch, most of them chance meetings --, one ballot per such lane gives the lanes with its hash and
			 * every member finds its neighbours among the inserting ones by itself; lz4_enc5.hip.  Before: two folded 1 024-bit
			 * filters and a pass with two ballots over both lanes of every pair the fold threw together) */
			const bool in = act || probe;
			const u32 fw = (hv & 2047u) >> 5, fb = in ? 1u << (hv & 31u) : 0u;
			u32 o_ = 0;
			if (in)
				o_ = atomicOr(&H.filter[fw], fb);
			u64 late = wv_ballot((o_ & fb) != 0); /* (the emulator's lanes meet here: every atomic is made before a word is cleared) */
			if (in)
				H.filter[fw] = 0;
			const u64 actm = wv_ballot(act);
			while (late) {
				const int f = wv_ffs(late) - 1;
				const u32 hvf = wv_readlane(hv, f);
				const u64 E = wv_ballot(hv == hvf); /* (a lane that takes no part has a hash value of its own) */
				late &= ~E;
				const u64 EA = E & actm; /* (the lookup's lane chains nothing behind it) */
				if ((E >> (u32)lane) & 1ull) {
					const u64 below = EA & ((1ull << (u32)lane) - 1ull);
					if (below != 0)
						pred = 63u - (u32)__builtin_clzll(below); /* the nearest earlier inserting lane */
					if (act && (EA >> (u32)lane) > 1ull)
						last = false; /* a later inserting lane owns the table */
				}
			}
		}
		const u32 from = pred < 64u ? base + pred : prev;
		if (act) {
			u32 delta = idx - from;
			if (delta > HC_DIST_MAX)
				delta = HC_DIST_MAX;
			H.chain[idx & (HC_MAXD - 1)] = (u16)delta;
			if (last)
				H.hash[hv] = idx;
		}
		if (cnt)
			H.next_to_update = base + cnt;
		wv_sync(); /* the next batch and the search read what other lanes stored */
		if (fin) {
			*pattern = wv_readlane(w, (int)cnt);
			*head = wv_readlane(from, (int)cnt);
			return;
		}
	}
}

/* number of equal bytes of a[0..) and b[0..), at most `limit` (wave-uniform arguments and result) */
static __device__ u32 hc_count(const u8 *a, const u8 *b, u32 limit, int lane)
{
	u32 done = 0;
	for (;;) {
		const u32 i = done + (u32)lane;
		const bool stop = i >= limit || a[i] != b[i];
		const u64 sm = wv_ballot(stop);
		if (sm) {
			const u32 r = done + (u32)wv_ffs(sm) - 1;
			return r < limit ? r : limit;
		}
		done += 64;
	}
}

/* LZ4HC_countBack: how far (as a negative number, >= min) the match extends backwards */
static __device__ int hc_count_back(const u8 *s, u32 ip, u32 m, int min, int lane)
{
	const u32 room = (u32)(-min);
	u32 done = 0;
	for (;;) {
		const u32 i = done + (u32)lane;
		const bool stop = i >= room || s[ip - 1 - i] != s[m - 1 - i];
		const u64 sm = wv_ballot(stop);
		if (sm) {
			u32 r = done + (u32)wv_ffs(sm) - 1;
			if (r > room)
				r = room;
			return -(int)r;
		}
		done += 64;
	}
}

/* LZ4HC_countPattern: bytes from a[0] on that continue the repetition of the 4-byte pattern (phase 0
 * at a[0]), at most `limit`; LZ4HC_reverseCountPattern: the same backwards from a[-1], at most `room` */
static __device__ u32 hc_count_pattern(const u8 *a, u32 limit, u32 pattern, int lane)
{
	u32 done = 0;
	for (;;) {
		const u32 i = done + (u32)lane;
		const bool stop = i >= limit || a[i] != (u8)(pattern >> (8 * (i & 3)));
		const u64 sm = wv_ballot(stop);
		if (sm) {
			const u32 r = done + (u32)wv_ffs(sm) - 1;
			return r < limit ? r : limit;
		}
		done += 64;
	}
}
static __device__ u32 hc_reverse_count_pattern(const u8 *a, u32 room, u32 pattern, int lane)
{
	u32 done = 0;
	for (;;) {
		const u32 i = done + (u32)lane; /* byte a[-1 - i] against pattern byte 3 - (i & 3) */
		const bool stop = i >= room || a[-1 - (int)i] != (u8)(pattern >> (8 * (3 - (i & 3))));
		const u64 sm = wv_ballot(stop);
		if (sm) {
			const u32 r = done + (u32)wv_ffs(sm) - 1;
			return r < room ? r : room;
		}
		done += 64;
	}
}

/* LZ4HC_InsertAndGetWiderMatch.  Levels 3..9 never swap chains and analyse patterns beyond 128
 * attempts (level 9), as liblz4 does; the optimal parser (levels 10..12) turns both on. */
static __device__ int hc_wider_core(HcState &H, u32 ip, u32 low_limit, u32 high_limit, int longest, u32 *mpos,
				    u32 *spos, int max_attempts, bool pattern_analysis, bool chain_swap, int lane)
{
	const u8 *const s = H.src;
	const u32 ip_index = ip + HC_BASE;
	const u32 lowest = (HC_BASE + HC_DIST_MAX + 1 > ip_index) ? HC_BASE : ip_index - HC_DIST_MAX;
	const int look_back = (int)(ip - low_limit);
	int attempts = max_attempts;
	int repeat = 0; /* 0 untested, 1 not a repetition, 2 confirmed */
	u32 src_pattern_len = 0, match_chain_pos = 0;
	u32 pattern, match_index;

	hc_insert_lookup(H, ip_index, &pattern, &match_index, lane);
	while (match_index >= lowest && attempts > 0) {
		const u32 m = match_index - HC_BASE;
		int match_len = 0;
		attempts--;
		/* everything this candidate can need in ONE memory round trip: lanes 0..4 each fetch a
		 * different dword -- the two tails of the quick test, the candidate's first four bytes, the
		 * chain link that is followed and the link at the candidate itself (pattern analysis) */
		u32 d_follow, d_here, tail_a, tail_b, cand4;
		{
			const u8 *addr = s + low_limit + (u32)longest - 1;
			if (lane == 1)
				addr = s + m - (u32)look_back + (u32)longest - 1;
			else if (lane == 2)
				addr = s + m;
			else if (lane == 3)
				addr = (const u8 *)(H.chain + ((match_index + match_chain_pos) & (HC_MAXD - 1)));
			else if (lane == 4)
				addr = (const u8 *)(H.chain + (match_index & (HC_MAXD - 1)));
			const u32 v = ld32u(addr);
			tail_a = wv_readlane(v, 0) & 0xFFFFu;
			tail_b = wv_readlane(v, 1) & 0xFFFFu;
			cand4 = wv_readlane(v, 2);
			d_follow = wv_readlane(v, 3) & 0xFFFFu;
			d_here = wv_readlane(v, 4) & 0xFFFFu;
		}
		if (tail_a == tail_b) {
			if (cand4 == pattern) {
				int back = 0;
				if (look_back) {
					const int min_i = -look_back, min_m = -(int)m;
					back = hc_count_back(s, ip, m, min_i > min_m ? min_i : min_m, lane);
				}
				int ml = (int)HC_MINMATCH +
					 (int)hc_count(s + ip + HC_MINMATCH, s + m + HC_MINMATCH, high_limit - (ip + HC_MINMATCH), lane);
				ml -= back;
				match_len = ml;
				if (ml > longest) {
					longest = ml;
					*mpos = (u32)((int)m + back);
					*spos = (u32)((int)ip + back);
				}
			}
		}
		if (chain_swap && match_len == longest) {
			/* a match as long as the best one: go on along the chain of the position inside it whose
			 * next candidate lies farthest back (forward searches only) */
			if (match_index + (u32)longest <= ip_index) {
				const int k_trigger = 4;
				u32 dist_to_next = 1;
				const int end = longest - (int)HC_MINMATCH + 1;
				int step = 1, accel = 1 << k_trigger;
				for (int pos = 0; pos < end; pos += step) {
					const u32 cand_dist = hc_uld_chain(H, match_index + (u32)pos);
					step = (accel++ >> k_trigger);
					if (cand_dist > dist_to_next) {
						dist_to_next = cand_dist;
						match_chain_pos = (u32)pos;
						accel = 1 << k_trigger;
					}
				}
				if (dist_to_next > 1) {
					if (dist_to_next > match_index)
						break;
					match_index -= dist_to_next;
					continue;
				}
				d_follow = hc_uld_chain(H, match_index + match_chain_pos); /* the position may have moved */
			}
		}
		if (pattern_analysis && match_chain_pos == 0 && d_here == 1) {
			/* the candidate sits inside a run of one byte value: jump to where that run can match
			 * the run at ip in full instead of walking it link by link (oracle/lz4hc_oracle.c) */
			const u32 cand = match_index - 1;
			if (repeat == 0) {
				if (((pattern & 0xFFFF) == (pattern >> 16)) & ((pattern & 0xFF) == (pattern >> 24))) {
					repeat = 2;
					src_pattern_len = hc_count_pattern(s + ip + 4, high_limit - (ip + 4), pattern, lane) + 4;
				} else {
					repeat = 1;
				}
			}
			if (repeat == 2 && cand >= lowest) {
				const u32 mp = cand - HC_BASE;
				if (uld32(s + mp) == pattern) {
					const u32 fwd = hc_count_pattern(s + mp + 4, high_limit - (mp + 4), pattern, lane) + 4;
					u32 back = hc_reverse_count_pattern(s + mp, mp, pattern, lane);
					{
						const u32 far = cand - back;
						back = cand - (far > lowest ? far : lowest);
					}
					const u32 cur = back + fwd;
					if (cur >= src_pattern_len && fwd <= src_pattern_len) {
						match_index = cand + fwd - src_pattern_len;
					} else {
						match_index = cand - back;
						if (look_back == 0) {
							const u32 max_ml = cur < src_pattern_len ? cur : src_pattern_len;
							if ((u32)longest < max_ml) {
								if (ip_index - match_index > HC_DIST_MAX)
									break;
								longest = (int)max_ml;
								*mpos = match_index - HC_BASE;
								*spos = ip;
							}
							{
								const u32 d = hc_uld_chain(H, match_index);
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
		match_index -= d_follow;
	}
	return longest;
}
static __device__ __forceinline__ int hc_wider(HcState &H, u32 ip, u32 low_limit, u32 high_limit, int longest, u32 *mpos,
					       u32 *spos, int max_attempts, int lane)
{
	return hc_wider_core(H, ip, low_limit, high_limit, longest, mpos, spos, max_attempts, max_attempts > 128, false, lane);
}

/* ---- the first search of a sequence, 64 positions at a time (levels 3..8; round 6) ------------------------------------
 * LZ4HC_compress_hashChain looks for the next match with `ml = InsertAndFindBestMatch(ip); if (ml < MINMATCH) ip++;` -- 4.6
 * searches per sequence of the bench text, each a chain walk of dependent memory round trips (insert + lookup, then one per
 * candidate).  Unlike the fast levels' table, HC's tables do not depend on the parse: EVERY position is inserted, in order,
 * before a later one is searched.  So what the search at position p returns -- longest = MINMATCH - 1, no backward extension --
 * is a function of the data alone, and the searches of 64 consecutive positions can be made side by side, one per lane, against
 * the tables as they are once every position in front of the window is inserted: the same hash head and chain links the serial
 * search would follow, unless a position of the same window has the same hash (then the serial head would be that position:
 * the lane is left to the serial search), the same candidates, the same "strictly longer wins".  A lane compares 28 bytes behind
 * the first four by itself; a candidate that is equal throughout leaves the lane to the serial search too.  The results stay
 * valid for the whole window whatever the lazy evaluation between two scans inserts (only positions of the window, never seen
 * by a lane that has no twin), so one build serves the ~5 sequences of a window: 4 round trips per window where the serial scan
 * took ~25 per sequence.  Levels 9..12 (pattern analysis, chain swap, the optimal parser) keep the serial search. */
#define HCW_FWD 28u
#ifdef ZMT_EMU
/* developer statistics of the emulator build (tools/emu_hc_stats.py): lane 0 counts */
extern "C" { unsigned long long zmt_hc_stat[16]; }
#define HC_STAT(i) do { if (lane == 0) zmt_hc_stat[i]++; } while (0)
#define HC_STAT_ALL(i, c) do { if (c) __atomic_fetch_add(&zmt_hc_stat[i], 1ull, __ATOMIC_RELAXED); } while (0)
#else
#define HC_STAT_ALL(i, c) do { } while (0)
#define HC_STAT(i) do { } while (0)
#endif
#define HCW_SLOTS 4 /* candidates of a position the window keeps for the wider searches (level 3 walks at most 4) */
struct HcWin {
	u32 w0;    /* first position (chunk-relative) of the window; HCW_NONE: no window */
	u32 ml, ref; /* this lane's position: the search's result (ml >= MINMATCH: a match at chunk position ref) */
	bool hard; /* the serial search decides (a twin in the window, or a candidate longer than the lane can count) */
	/* the first HCW_SLOTS candidates of the position's chain whose four bytes equal the position's: chunk position (HCW_NONE:
	 * the candidate in that place of the chain has other bytes, or the chain ended) and equal bytes behind the four */
	u32 cm[HCW_SLOTS], cf[HCW_SLOTS];
};
#define HCW_NONE 0xFFFFFFFFu

static __device__ void hc_win_build(HcState &H, HcWin &W, u32 ip, u32 mflimit, u32 matchlimit, int max_attempts, int lane)
{
	const u8 *const s = H.src;
	u32 dummy_p, dummy_h;
	HC_STAT(0);
	hc_insert_lookup(H, ip + HC_BASE, &dummy_p, &dummy_h, lane); /* every position in front of the window is in the chains */
	const u32 p = ip + (u32)lane;
	const bool valid = p <= mflimit;
	/* the position's 32 bytes and its table entry (invalid lanes read the window's first position: no exec-mask regions) */
	const u8 *const own = s + (valid ? p : ip);
	u64 o0, o1, o2, o3;
	{
		struct { u64 a, b; } q;
		__builtin_memcpy(&q, own, 16);
		o0 = q.a;
		o1 = q.b;
		__builtin_memcpy(&q, own + 16, 16);
		o2 = q.a;
		o3 = q.b;
	}
	const u32 pattern = (u32)o0;
	const u32 hv = hc_hash(pattern);
	u32 match_index = H.hash[hv];
	/* twins: an earlier position of the window with the same hash is the head of this position's chain when its turn comes
	 * (every position in front of it is inserted by then), and that position's own head follows it: the link the insertion
	 * would write.  prev = the nearest earlier lane with this lane's hash (two folded LDS bitmaps find the lanes that may
	 * have one, an exact pass over those settles it) */
	u32 prev = 64u;
	{
		/* (the filter of hc_insert_lookup: one bitmap, the lanes its returning atomic names, one ballot each) */
		const u32 fw = (hv & 2047u) >> 5, fb = valid ? 1u << (hv & 31u) : 0u;
		u32 o_ = 0;
		if (valid)
			o_ = atomicOr(&H.filter[fw], fb);
		u64 late = wv_ballot((o_ & fb) != 0);
		if (valid)
			H.filter[fw] = 0;
		const u64 validm = wv_ballot(valid);
		while (late) {
			const int i = wv_ffs(late) - 1;
			const u32 hi_ = wv_readlane(hv, i);
			const u64 E = wv_ballot(hv == hi_) & validm;
			late &= ~E;
			const u64 below = E & ((1ull << (u32)lane) - 1ull);
			if (((E >> (u32)lane) & 1ull) && below != 0)
				prev = 63u - (u32)__builtin_clzll(below);
		}
	}
	const bool twin = prev != 64u;
	if (twin)
		match_index = ip + prev + HC_BASE;
	/* (a chain link is a 16-bit distance: a head farther back than that is out of reach either way) */
	const u32 head0 = match_index;
	const bool any_twin = wv_any(twin);
	const u32 p_index = p + HC_BASE;
	const u32 lowest = (HC_BASE + HC_DIST_MAX + 1 > p_index) ? HC_BASE : p_index - HC_DIST_MAX;
	const u32 limit = matchlimit - (p + HC_MINMATCH); /* (valid: p <= mflimit < matchlimit - 4) */
	u32 longest = HC_MINMATCH - 1, ref = 0;
	bool undecided = false;
	int attempts = max_attempts;
	bool active = valid;
	for (int k = 0; k < HCW_SLOTS; k++) {
		W.cm[k] = HCW_NONE;
		W.cf[k] = 0;
	}
	for (int step = 0;; step++) {
		active = active && match_index >= lowest && attempts > 0;
		if (!wv_any(active))
			break;
		attempts--;
		const u32 m = active ? match_index - HC_BASE : ip;
		u64 c0, c1, c2, c3;
		{
			struct { u64 a, b; } q;
			__builtin_memcpy(&q, s + m, 16);
			c0 = q.a;
			c1 = q.b;
			__builtin_memcpy(&q, s + m + 16, 16);
			c2 = q.a;
			c3 = q.b;
		}
		const u32 delta = H.chain[(active ? match_index : HC_BASE) & (HC_MAXD - 1)];
		HC_STAT_ALL(8, active);
		HC_STAT_ALL(9, active && (u32)c0 == pattern);
		if (active && (u32)c0 == pattern) {
			/* equal bytes behind the four: 28 looked at */
			const u32 da = (u32)(o0 >> 32) ^ (u32)(c0 >> 32);
			const u64 db = o1 ^ c1, dc = o2 ^ c2, dd = o3 ^ c3;
			u32 cnt = da   ? (u32)__builtin_ctz(da) >> 3
				  : db ? 4u + ((u32)__builtin_ctzll(db) >> 3)
				  : dc ? 12u + ((u32)__builtin_ctzll(dc) >> 3)
				  : dd ? 20u + ((u32)__builtin_ctzll(dd) >> 3)
				       : HCW_FWD;
			if (cnt == HCW_FWD && limit > HCW_FWD)
				undecided = true; /* runs on: its length is the serial search's to count */
			if (cnt > limit)
				cnt = limit;
			const u32 ml = HC_MINMATCH + cnt;
			if (ml > longest) {
				longest = ml;
				ref = m;
			}
			ZMT_UNROLL
			for (int k = 0; k < HCW_SLOTS; k++) {
				if (step == k) { /* (step is wave-uniform: one of the four pairs is written) */
					W.cm[k] = m;
					W.cf[k] = cnt;
				}
			}
		}
		/* the link behind a position of the window is not in the table yet: it leads to that position's head */
		u32 next = match_index - delta;
		if (any_twin) {
			const bool inwin = active && match_index >= ip + HC_BASE;
			const u32 hw = wv_shfl(head0, (int)((match_index - ip - HC_BASE) & 63u));
			if (inwin)
				next = hw;
		}
		match_index = next;
	}
	W.w0 = ip;
	W.ml = valid ? longest : 0u;
	W.ref = ref;
	W.hard = valid && undecided;
}

/* a wider search (LZ4HC_InsertAndGetWiderMatch with longest = the match in hand, backward extension down to low_limit) from
 * the window's record of its position q: the candidates are the lane's, their forward counts too; what is new is how far
 * each reaches backwards -- at most q - low_limit bytes -- which 16 lanes per candidate compare in one round trip.  The
 * reference's two-byte test in front of each candidate is implied whenever the candidate beats the length in hand (its bytes
 * lie inside the forward part that is already known equal: q - low_limit is the length in hand minus 2 or 3), so the result
 * is the first candidate with the greatest 4 + forward + backward above `longest`, as the serial walk's.  false: not
 * available (no window over q, a lane the serial search decides, more than 16 bytes to look back, a level that walks more
 * candidates than the window keeps) */
static __device__ bool hc_wider_fast(HcState &H, HcWin &W, u32 q, u32 low_limit, u32 mflimit, u32 matchlimit, int max_attempts,
				     int *longest, u32 *mpos, u32 *spos, int lane)
{
	if (max_attempts > HCW_SLOTS || q > mflimit)
		return false;
	if (W.w0 == HCW_NONE || q < W.w0 || q >= W.w0 + 64u) {
		/* the match in hand runs out of the window: the next window now, from q on (the positions in front of q lie inside
		 * the match: no first search will ask for them, and their insertion does not depend on who asks) */
		if (q < W.w0 || H.next_to_update > q + HC_BASE) {
			HC_STAT(4);
			return false;
		}
		hc_win_build(H, W, q, mflimit, matchlimit, max_attempts, lane);
	}
	const int j = (int)(q - W.w0);
	const u32 look_back = q - low_limit;
	if (look_back > 32u || wv_readlane((u32)W.hard, j)) {
		HC_STAT(look_back > 32u ? 5 : 6);
		return false;
	}
	HC_STAT(3);
	const u8 *const s = H.src;
	/* half a wave looks back from one candidate: candidates 0 and 1, then 2 and 3 -- all four byte pairs are requested before
	 * the first is looked at: one round trip */
	const u32 g = (u32)lane >> 5, i = (u32)lane & 31u;
	u32 m_[HCW_SLOTS], f_[HCW_SLOTS];
	ZMT_UNROLL
	for (int k = 0; k < HCW_SLOTS; k++) {
		m_[k] = wv_readlane(W.cm[k], j);
		f_[k] = wv_readlane(W.cf[k], j);
	}
	const u32 ma = g ? m_[1] : m_[0], mb = g ? m_[3] : m_[2];
	const u32 ra = ma == HCW_NONE ? 0u : (look_back < ma ? look_back : ma);
	const u32 rb = mb == HCW_NONE ? 0u : (look_back < mb ? look_back : mb);
	const u32 own = (i < ra || i < rb) ? (u32)s[q - 1 - i] : 0u;
	const u32 ca = i < ra ? (u32)s[ma - 1 - i] : 0x100u, cb = i < rb ? (u32)s[mb - 1 - i] : 0x100u;
	const u64 nea = ~wv_ballot(own == ca), neb = ~wv_ballot(own == cb);
	int best = *longest;
	ZMT_UNROLL
	for (int k = 0; k < HCW_SLOTS; k++) {
		if (m_[k] == HCW_NONE)
			continue;
		const u32 sl = (u32)((k < 2 ? nea : neb) >> (32 * (k & 1)));
		u32 back = sl ? (u32)__builtin_ctz(sl) : 32u;
		const u32 rk = look_back < m_[k] ? look_back : m_[k];
		if (back > rk)
			back = rk;
		const int ml = (int)(HC_MINMATCH + f_[k] + back);
		if (ml > best) {
			best = ml;
			*mpos = m_[k] - back;
			*spos = q - back;
		}
	}
	*longest = best;
	return true;
}

/* LZ4HC_encodeSequence: 1 when the output limit is hit.  op / oend are offsets into dst. */
static __device__ int hc_encode(const u8 *s, u8 *dst, u32 *ip, u32 *op, u32 *anchor, int ml, u32 mpos, u32 oend,
				int lane)
{
	const u32 tokpos = (*op)++;
	u32 length = *ip - *anchor, token;
	if (*op + (length / 255) + length + (2 + 1 + HC_LASTLITERALS) > oend)
		return 1;
	if (length >= 15) {
		u32 len = length - 15;
		token = 15u << 4;
		const u32 n255 = len / 255;
		for (u32 i = (u32)lane; i < n255; i += 64)
			dst[*op + i] = 255;
		if (lane == 0)
			dst[*op + n255] = (u8)(len - n255 * 255);
		*op += n255 + 1;
	} else {
		token = length << 4;
	}
	wave_copy(dst + *op, s + *anchor, length, lane);
	*op += length;
	if (lane == 0) {
		dst[*op] = (u8)(*ip - mpos);
		dst[*op + 1] = (u8)((*ip - mpos) >> 8);
	}
	*op += 2;
	length = (u32)ml - HC_MINMATCH;
	if (*op + (length / 255) + (1 + HC_LASTLITERALS) > oend)
		return 1;
	if (length >= 15) {
		token += 15;
		length -= 15;
		/* the reference writes pairs of 255 then singles: the bytes are n x 255 followed by the rest */
		const u32 n255 = length / 255;
		for (u32 i = (u32)lane; i < n255; i += 64)
			dst[*op + i] = 255;
		if (lane == 0)
			dst[*op + n255] = (u8)(length - n255 * 255);
		*op += n255 + 1;
	} else {
		token += length;
	}
	if (lane == 0)
		dst[tokpos] = (u8)token;
	*ip += (u32)ml;
	*anchor = *ip;
	return 0;
}

/* LZ4HC_compress_hashChain over chunk positions [start, start + n) into dst[0, cap): size or 0 */
static __device__ u32 hc_block(HcState &H, u32 start, u32 n, u8 *dst, u32 cap, int max_attempts, int lane)
{
	const u8 *const s = H.src;
	u32 ip = start, anchor = start;
	const u32 iend = start + n;
	const u32 mflimit = iend - HC_MFLIMIT, matchlimit = iend - HC_LASTLITERALS;
	u32 op = 0;
	const u32 oend = cap;
	int ml0, ml, ml2, ml3;
	u32 start0, ref0, ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0;

	HcWin W;
	W.w0 = HCW_NONE;
	W.ml = 0;
	W.ref = 0;
	W.hard = false;
	for (int k = 0; k < HCW_SLOTS; k++) {
		W.cm[k] = HCW_NONE;
		W.cf[k] = 0;
	}
	const bool use_win = max_attempts <= 128; /* levels 3..8: no pattern analysis in the search */

	/* the wider searches of the lazy evaluation walk the same chain as the first search of their position, with the same
	 * attempts, and take a candidate only if its four bytes equal the position's: a position of the window whose lane met no
	 * such candidate (and has no twin) has nothing to offer them either -- the search is skipped, its insertions are made by
	 * the next one that runs (they do not depend on who asks) */
#define HCW_NOTHING(Q)                                                                                             \
	(use_win && W.w0 != HCW_NONE && (Q) >= W.w0 && (Q) < W.w0 + 64u && (Q) <= mflimit &&                        \
	 wv_readlane((u32)(!W.hard && W.ml < HC_MINMATCH), (int)((Q) - W.w0)) != 0)

	if (n < HC_MFLIMIT + 1)
		goto last_literals;
	while (ip <= mflimit) {
		if (use_win && H.next_to_update <= ip + HC_BASE) {
			/* the next position whose search returns a match: from the window's lanes; a lane that cannot tell asks the
			 * serial search */
			if (W.w0 == HCW_NONE || ip < W.w0 || ip >= W.w0 + 64u)
				hc_win_build(H, W, ip, mflimit, matchlimit, max_attempts, lane);
			const u64 cand = wv_ballot((W.hard || W.ml >= HC_MINMATCH) && W.w0 + (u32)lane >= ip);
			if (cand == 0) {
				ip = W.w0 + 64u;
				continue;
			}
			const int j = wv_ffs(cand) - 1;
			ip = W.w0 + (u32)j;
			if (wv_readlane((u32)W.hard, j)) {
				HC_STAT(2);
				u32 dummy = ip;
				ml = hc_wider(H, ip, ip, matchlimit, (int)HC_MINMATCH - 1, &ref, &dummy, max_attempts, lane);
				if (ml < (int)HC_MINMATCH) {
					ip++;
					continue;
				}
			} else {
				ml = (int)wv_readlane(W.ml, j);
				ref = wv_readlane(W.ref, j);
			}
		} else {
			u32 dummy = ip;
			ml = hc_wider(H, ip, ip, matchlimit, (int)HC_MINMATCH - 1, &ref, &dummy, max_attempts, lane);
			if (ml < (int)HC_MINMATCH) {
				ip++;
				continue;
			}
		}
		HC_STAT(1);
		start0 = ip;
		ref0 = ref;
		ml0 = ml;
search2:
		ml2 = ml;
		if (ip + (u32)ml <= mflimit && !HCW_NOTHING(ip + (u32)ml - 2)) {
			if (!use_win || !hc_wider_fast(H, W, ip + (u32)ml - 2, ip, mflimit, matchlimit, max_attempts, &ml2, &ref2, &start2, lane))
				ml2 = hc_wider(H, ip + (u32)ml - 2, ip, matchlimit, ml, &ref2, &start2, max_attempts, lane);
		}
		if (ml2 == ml) {
			if (hc_encode(s, dst, &ip, &op, &anchor, ml, ref, oend, lane))
				return 0;
			continue;
		}
		if (start0 < ip) {
			if (start2 < ip + (u32)ml0) {
				ip = start0;
				ref = ref0;
				ml = ml0;
			}
		}
		if (start2 - ip < 3) {
			ml = ml2;
			ip = start2;
			ref = ref2;
			goto search2;
		}
search3:
		if (start2 - ip < (u32)HC_OPTIMAL_ML) {
			int new_ml = ml;
			if (new_ml > HC_OPTIMAL_ML)
				new_ml = HC_OPTIMAL_ML;
			if (ip + (u32)new_ml > start2 + (u32)ml2 - HC_MINMATCH)
				new_ml = (int)(start2 - ip) + ml2 - (int)HC_MINMATCH;
			const int correction = new_ml - (int)(start2 - ip);
			if (correction > 0) {
				start2 += (u32)correction;
				ref2 += (u32)correction;
				ml2 -= correction;
			}
		}
		ml3 = ml2;
		if (start2 + (u32)ml2 <= mflimit && !HCW_NOTHING(start2 + (u32)ml2 - 3)) {
			if (!use_win || !hc_wider_fast(H, W, start2 + (u32)ml2 - 3, start2, mflimit, matchlimit, max_attempts, &ml3, &ref3, &start3, lane))
				ml3 = hc_wider(H, start2 + (u32)ml2 - 3, start2, matchlimit, ml2, &ref3, &start3, max_attempts, lane);
		}
		if (ml3 == ml2) {
			if (start2 < ip + (u32)ml)
				ml = (int)(start2 - ip);
			if (hc_encode(s, dst, &ip, &op, &anchor, ml, ref, oend, lane))
				return 0;
			ip = start2;
			if (hc_encode(s, dst, &ip, &op, &anchor, ml2, ref2, oend, lane))
				return 0;
			continue;
		}
		if (start3 < ip + (u32)ml + 3) {
			if (start3 >= ip + (u32)ml) {
				if (start2 < ip + (u32)ml) {
					const int correction = (int)(ip + (u32)ml - start2);
					start2 += (u32)correction;
					ref2 += (u32)correction;
					ml2 -= correction;
					if (ml2 < (int)HC_MINMATCH) {
						start2 = start3;
						ref2 = ref3;
						ml2 = ml3;
					}
				}
				if (hc_encode(s, dst, &ip, &op, &anchor, ml, ref, oend, lane))
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
		if (start2 < ip + (u32)ml) {
			if (start2 - ip < (u32)HC_OPTIMAL_ML) {
				if (ml > HC_OPTIMAL_ML)
					ml = HC_OPTIMAL_ML;
				if (ip + (u32)ml > start2 + (u32)ml2 - HC_MINMATCH)
					ml = (int)(start2 - ip) + ml2 - (int)HC_MINMATCH;
				const int correction = ml - (int)(start2 - ip);
				if (correction > 0) {
					start2 += (u32)correction;
					ref2 += (u32)correction;
					ml2 -= correction;
				}
			} else {
				ml = (int)(start2 - ip);
			}
		}
		if (hc_encode(s, dst, &ip, &op, &anchor, ml, ref, oend, lane))
			return 0;
		ip = start2;
		ref = ref2;
		ml = ml2;
		start2 = start3;
		ref2 = ref3;
		ml2 = ml3;
		goto search3;
	}
#undef HCW_NOTHING
last_literals:
	{
		const u32 run = iend - anchor;
		const u32 lit_len = (run + 255 - 15) / 255;
		if (op + 1 + lit_len + run > oend)
			return 0;
		if (run >= 15) {
			const u32 acc = run - 15, n255 = acc / 255;
			if (lane == 0)
				dst[op] = 15u << 4;
			op++;
			for (u32 i = (u32)lane; i < n255; i += 64)
				dst[op + i] = 255;
			if (lane == 0)
				dst[op + n255] = (u8)(acc - n255 * 255);
			op += n255 + 1;
		} else {
			if (lane == 0)
				dst[op] = (u8)(run << 4);
			op++;
		}
		wave_copy(dst + op, s + anchor, run, lane);
		op += run;
	}
	return op;
}

/* ---- levels 10..12: LZ4HC_compress_optimal (oracle/lz4hc_oracle.c: hc_opt_block) ----------------
 * The price table lives in the wave's global scratch, one column per field.  The loops over the
 * lengths of a match (up to 4 095 prices each) run 64 lengths per step; the position loop, the searches
 * and the reverse traversal are wave-uniform.  Lane 0 owns single-entry stores; wv_sync() between a
 * phase that stores and one that loads. */
static __device__ __forceinline__ int hc_lit_price(int litlen)
{
	int price = litlen;
	if (litlen >= 15)
		price += 1 + (litlen - 15) / 255;
	return price;
}
static __device__ __forceinline__ int hc_seq_price(int litlen, int mlen)
{
	int price = 1 + 2 + hc_lit_price(litlen);
	if (mlen >= 15 + (int)HC_MINMATCH)
		price += 1 + (mlen - (15 + (int)HC_MINMATCH)) / 255;
	return price;
}
static __device__ __forceinline__ int hc_uld_int(const int *p) { return (int)wv_readfirst((u32)*p); }
static __device__ __forceinline__ void hc_opt_set(HcState &H, int pos, int price, int off, int mlen, int litlen, int lane)
{
	if (lane == 0) {
		H.o_price[pos] = price;
		H.o_off[pos] = off;
		H.o_mlen[pos] = mlen;
		H.o_litlen[pos] = litlen;
	}
}
static __device__ void hc_opt_trailing(HcState &H, int last_match_pos, int lane)
{
	wv_sync();
	const int base = hc_uld_int(H.o_price + last_match_pos);
	if (lane >= 1 && lane <= HC_TRAILING) {
		H.o_mlen[last_match_pos + lane] = 1;
		H.o_off[last_match_pos + lane] = 0;
		H.o_litlen[last_match_pos + lane] = lane;
		H.o_price[last_match_pos + lane] = base + hc_lit_price(lane);
	}
	wv_sync();
}
static __device__ void hc_find_longer(HcState &H, u32 ip, u32 high_limit, int min_len, int nb_searches, int *len, int *off,
				      int lane)
{
	u32 mpos = 0, spos = ip;
	const int ml = hc_wider_core(H, ip, ip, high_limit, min_len, &mpos, &spos, nb_searches, true, true, lane);
	*len = 0;
	*off = 0;
	if (ml <= min_len)
		return;
	*len = ml;
	*off = (int)(ip - mpos);
}

static __device__ u32 hc_opt_block(HcState &H, u32 start, u32 n, u8 *dst, u32 cap, int nb_searches, int sufficient_len,
				   bool full_update, int lane)
{
	const u8 *const s = H.src;
	u32 ip = start, anchor = start, op = 0;
	const u32 iend = start + n;
	const u32 mflimit = iend - HC_MFLIMIT, matchlimit = iend - HC_LASTLITERALS;
	const u32 oend = cap;

	if (sufficient_len >= HC_OPT_NUM)
		sufficient_len = HC_OPT_NUM - 1;
	if (n < HC_MFLIMIT + 1)
		goto last_literals;
	while (ip <= mflimit) {
		const int llen = (int)(ip - anchor);
		int best_mlen, best_off, cur, last_match_pos = 0, first_len, first_off;

		hc_find_longer(H, ip, matchlimit, (int)HC_MINMATCH - 1, nb_searches, &first_len, &first_off, lane);
		if (first_len == 0) {
			ip++;
			continue;
		}
		if (first_len > sufficient_len) {
			if (hc_encode(s, dst, &ip, &op, &anchor, first_len, ip - (u32)first_off, oend, lane))
				return 0;
			continue;
		}
		/* prices of the first positions (literals) and of the first match */
		if (lane < (int)HC_MINMATCH) {
			H.o_mlen[lane] = 1;
			H.o_off[lane] = 0;
			H.o_litlen[lane] = llen + lane;
			H.o_price[lane] = hc_lit_price(llen + lane);
		}
		for (int ml = (int)HC_MINMATCH + lane; ml <= first_len; ml += 64) {
			H.o_mlen[ml] = ml;
			H.o_off[ml] = first_off;
			H.o_litlen[ml] = llen;
			H.o_price[ml] = hc_seq_price(llen, ml);
		}
		last_match_pos = first_len;
		hc_opt_trailing(H, last_match_pos, lane);
		/* further positions */
		for (cur = 1; cur < last_match_pos; cur++) {
			const u32 cur_pos = ip + (u32)cur;
			int new_len, new_off;
			if (cur_pos > mflimit)
				break;
			const int p_cur = hc_uld_int(H.o_price + cur), p_next = hc_uld_int(H.o_price + cur + 1);
			if (full_update) {
				if (p_next <= p_cur && hc_uld_int(H.o_price + cur + (int)HC_MINMATCH) < p_cur + 3)
					continue;
			} else {
				if (p_next <= p_cur)
					continue;
			}
			hc_find_longer(H, cur_pos, matchlimit, full_update ? (int)HC_MINMATCH - 1 : last_match_pos - cur, nb_searches,
				       &new_len, &new_off, lane);
			if (!new_len)
				continue;
			if (new_len > sufficient_len || new_len + cur >= HC_OPT_NUM) {
				best_mlen = new_len;
				best_off = new_off;
				last_match_pos = cur + 1;
				goto encode;
			}
			const int cur_mlen = hc_uld_int(H.o_mlen + cur), cur_litlen = hc_uld_int(H.o_litlen + cur);
			/* before the match: literals (three entries, lanes 1..3) */
			{
				const int litlen = lane;
				if (litlen >= 1 && litlen < (int)HC_MINMATCH) {
					const int price = p_cur - hc_lit_price(cur_litlen) + hc_lit_price(cur_litlen + litlen);
					const int pos = cur + litlen;
					if (price < H.o_price[pos]) {
						H.o_mlen[pos] = 1;
						H.o_off[pos] = 0;
						H.o_litlen[pos] = cur_litlen + litlen;
						H.o_price[pos] = price;
					}
				}
			}
			wv_sync();
			/* the match at cur: one price per length, 64 lengths per step */
			{
				int ll, base;
				if (cur_mlen == 1) {
					ll = cur_litlen;
					base = (cur > ll) ? hc_uld_int(H.o_price + cur - ll) : 0;
				} else {
					ll = 0;
					base = p_cur;
				}
				bool took_last = false;
				for (int ml0 = (int)HC_MINMATCH; ml0 <= new_len; ml0 += 64) {
					const int ml = ml0 + lane;
					if (ml <= new_len) {
						const int pos = cur + ml;
						const int price = base + hc_seq_price(ll, ml);
						if (pos > last_match_pos + HC_TRAILING || price <= H.o_price[pos]) {
							if (ml == new_len)
								took_last = true;
							H.o_mlen[pos] = ml;
							H.o_off[pos] = new_off;
							H.o_litlen[pos] = ll;
							H.o_price[pos] = price;
						}
					}
				}
				if (wv_any(took_last) && last_match_pos < cur + new_len)
					last_match_pos = cur + new_len;
			}
			hc_opt_trailing(H, last_match_pos, lane);
		}
		wv_sync();
		best_mlen = hc_uld_int(H.o_mlen + last_match_pos);
		best_off = hc_uld_int(H.o_off + last_match_pos);
		cur = last_match_pos - best_mlen;
encode:
		/* reverse traversal: the chosen sequences, first to last */
		wv_sync();
		{
			int candidate_pos = cur, selected_ml = best_mlen, selected_off = best_off;
			for (;;) {
				const int next_ml = hc_uld_int(H.o_mlen + candidate_pos), next_off = hc_uld_int(H.o_off + candidate_pos);
				wv_sync();
				if (lane == 0) {
					H.o_mlen[candidate_pos] = selected_ml;
					H.o_off[candidate_pos] = selected_off;
				}
				selected_ml = next_ml;
				selected_off = next_off;
				if (next_ml > candidate_pos)
					break;
				candidate_pos -= next_ml;
			}
		}
		wv_sync();
		{
			int r = 0;
			while (r < last_match_pos) {
				const int ml = hc_uld_int(H.o_mlen + r), offset = hc_uld_int(H.o_off + r);
				if (ml == 1) {
					ip++;
					r++;
					continue;
				}
				r += ml;
				if (hc_encode(s, dst, &ip, &op, &anchor, ml, ip - (u32)offset, oend, lane))
					return 0;
			}
		}
		wv_sync();
	}
last_literals:
	{
		const u32 run = iend - anchor;
		const u32 lit_len = (run + 255 - 15) / 255;
		if (op + 1 + lit_len + run > oend)
			return 0;
		if (run >= 15) {
			const u32 acc = run - 15, n255 = acc / 255;
			if (lane == 0)
				dst[op] = 15u << 4;
			op++;
			for (u32 i = (u32)lane; i < n255; i += 64)
				dst[op + i] = 255;
			if (lane == 0)
				dst[op + n255] = (u8)(acc - n255 * 255);
			op += n255 + 1;
		} else {
			if (lane == 0)
				dst[op] = (u8)(run << 4);
			op++;
		}
		wave_copy(dst + op, s + anchor, run, lane);
		op += run;
	}
	return op;
}

/*
 * Persistent grid: wave w takes records w, w + gridDim.x, ...; scratch + w * HC_SCRATCH holds its
 * hash table and chain.  Record layout as the level-1 encoder writes it (lz4_enc3.hip).
 */
extern "C" __global__ void __launch_bounds__(64)
zmt_lz4hc_enc_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nrec, u8 *__restrict__ slots,
		     u64 slot_stride, u32 *__restrict__ rec_len, const u32 *__restrict__ chk,
		     u8 *__restrict__ scratch, int level)
{
	/* liblz4 clTable: searches per position, and for the optimal parser the "good enough" length */
	const int max_attempts = level <= 9 ? 1 << (level - 1) : level == 10 ? 96 : level == 11 ? 512 : 16384;
	const int sufficient_len = level == 10 ? 64 : level == 11 ? 128 : HC_OPT_NUM;
	const int lane = wv_lane();
	__shared__ u32 hc_filter[64];
	HcState H;
	hc_filter[lane] = 0;
	H.filter = hc_filter;
	H.hash = (u32 *)(scratch + (u64)blockIdx.x * HC_SCRATCH);
	H.chain = (u16 *)(scratch + (u64)blockIdx.x * HC_SCRATCH + 32768u * 4u);
	H.o_price = (int *)(scratch + (u64)blockIdx.x * HC_SCRATCH + 32768u * 4u + HC_MAXD * 2u);
	H.o_off = H.o_price + HC_OPT_ROWS;
	H.o_mlen = H.o_off + HC_OPT_ROWS;
	H.o_litlen = H.o_mlen + HC_OPT_ROWS;
	for (u32 rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
		const u64 start = (u64)rec * chunk;
		const u32 len = (u32)((n - start) < (u64)chunk ? (n - start) : (u64)chunk);
		const u8 *src = in + start;
		u8 *dst = slots + (u64)rec * slot_stride;
		const bool single = len <= ZMT_BLOCK;
		const u32 hdr = len ? 15 : 7;
		u32 op = 12 + hdr;
		if (lane == 0) {
			u8 d[10];
			st32u(dst, ZMT_SKIP_MAGIC);
			st32u(dst + 4, 4);
			st32u(dst + 12, ZMT_LZ4F_MAGIC);
			d[0] = (u8)(0x40 | (single ? 0x20 : 0) | (len ? 0x08 : 0) | 0x04);
			d[1] = 0x40;
			for (int i = 0; i < 8; i++)
				d[2 + i] = (i < 4) ? (u8)(len >> (8 * i)) : 0;
			for (u32 i = 0; i < hdr - 5; i++)
				dst[16 + i] = d[i];
			dst[12 + hdr - 1] = (u8)(xxh32_short(d, hdr - 5) >> 8);
		}
		/* a fresh context per frame: empty hash table (the chain needs no clearing: an entry is
		 * written before it can be reached) */
		for (u32 i = (u32)lane * 4; i < 32768u; i += 256) {
			H.hash[i] = 0;
			H.hash[i + 1] = 0;
			H.hash[i + 2] = 0;
			H.hash[i + 3] = 0;
		}
		wv_sync();
		H.src = src;
		H.next_to_update = HC_BASE;
		for (u32 pos = 0; pos < len; pos += ZMT_BLOCK) {
			const u32 blen = len - pos < ZMT_BLOCK ? len - pos : ZMT_BLOCK;
			u32 c = level >= 10 ? hc_opt_block(H, pos, blen, dst + op + 4, blen - 1, max_attempts, sufficient_len, level == 12, lane)
					    : hc_block(H, pos, blen, dst + op + 4, blen - 1, max_attempts, lane);
			u32 bh = c;
			if (c == 0) { /* did not shrink: stored; the chains keep what the attempt inserted */
				wv_sync(); /* every lane's stores of the attempt lie behind it before the same bytes are rewritten */
				wave_copy(dst + op + 4, src + pos, blen, lane);
				c = blen;
				bh = blen | 0x80000000u;
			}
			if (lane == 0)
				st32u(dst + op, bh);
			op += 4 + c;
		}
		if (lane == 0) {
			st32u(dst + op, 0);
			st32u(dst + op + 4, chk[rec]);
			st32u(dst + 8, op + 8 - 12);
			rec_len[rec] = op + 8;
		}
		wv_sync();
	}
}
