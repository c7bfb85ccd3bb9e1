"""Design-time simulation for the round-3 LZ4 decode pipeline (parse3 + copy3) on the BENCH text:
batch cuts as the parse kernel makes them, far-match fraction by ring size, watermark rounds, offsets CDF."""
import ctypes as C, os, struct, sys
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import helpers as H
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 4 << 20
buf = np.empty(n, np.uint8)
T.zmt_gen_text(buf.ctypes.data, n, 20260926, 0, 8)
s = H.oracle_compress(buf.tobytes(), 131072)
print("ratio %.3f" % (n / len(s)))

def records(s):
    i = 0
    while i < len(s):
        c = struct.unpack_from('<I', s, i + 8)[0]; p = i + 12 + 15; blocks = []
        while True:
            bh = struct.unpack_from('<I', s, p)[0]; p += 4
            if bh == 0: break
            bs = bh & 0x7fffffff
            if bh >> 31: p += bs; blocks.append(None); continue
            e = p + bs; seqs = []; p0 = p
            while p < e:
                q = p - p0; t = s[p]; p += 1; l = t >> 4
                if l == 15:
                    while True:
                        b = s[p]; p += 1; l += b
                        if b != 255: break
                p += l
                if p >= e: seqs.append((l, 0, 0, q, p - p0)); break
                o = s[p] | s[p + 1] << 8; p += 2; m = t & 15
                if m == 15:
                    while True:
                        b = s[p]; p += 1; m += b
                        if b != 255: break
                seqs.append((l, m + 4, o, q, p - p0))
            blocks.append(seqs)
        yield blocks
        i += 12 + c

XOUT = 2048; CAPL = CAPM = 64
offs = []; nb = ns = nsingle = 0; lanes = []; spans = []; rounds = []
fars = {4096: 0, 8192: 0, 16384: 0, 32768: 0}; nm = 0; inbatch = 0
for blocks in records(s):
    opos = 0
    for seqs in blocks:
        if seqs is None: continue
        cur = []  # open batch
        def close():
            global nb, nm, inbatch
            if not cur: return
            nb += 1; lanes.append(len(cur)); o0 = cur[0][5]; oend = cur[-1][5] + cur[-1][0] + cur[-1][1]
            spans.append(oend - o0)
            # far classification + watermark rounds
            fin = []; need = []; mp = []
            for (l, m, o, q, qe, op) in cur:
                mpos = op + l; src = mpos - o; nm += 1
                for W in fars:
                    if src < oend + 8 - W: fars[W] += 1
                e = src + min(m, o)
                if e > o0: inbatch += 1
                fin.append(e <= o0); need.append(e); mp.append(mpos)
            r = 0
            fin = np.array(fin); need = np.array(need); mp = np.array(mp)
            while not fin.all():
                r += 1; first = int(np.argmin(fin)); w = mp[first]
                fin = fin | (need <= w); fin[first] = True
            rounds.append(r)
            cur.clear()
        base = opos
        for (l, m, o, q, qe) in seqs:
            ns += 1
            if m: offs.append(o)
            small = m != 0 and l <= CAPL and m <= CAPM
            oe = opos + l + m
            if small and ((oe - 1) >> 12) != (opos >> 12): small = False   # crosses a 4 KiB lap alone
            if not small:
                close(); nsingle += 1
            else:
                if cur:
                    g0 = cur[0][3] & ~15
                    fits = len(cur) < 64 and qe - g0 <= 1016 and oe - cur[0][5] <= XOUT and ((oe - 1) >> 12) == (cur[0][5] >> 12)
                    if not fits: close()
                cur.append((l, m, o, q, qe, opos))
            opos = oe
        close()
offs = np.array(offs)
print("sequences %d, batches %d (mean %.1f lanes), singles %d (%.2f%% of sequences), descs per block %.0f" % (ns, nb, np.mean(lanes), nsingle, 100 * nsingle / ns, (nb + nsingle) / (n / 65536)))
print("batch out span mean %.0f p99 %d max %d" % (np.mean(spans), np.percentile(spans, 99), max(spans)))
print("offset CDF:", {k: "%.1f%%" % (100 * (offs < k).mean()) for k in (64, 256, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768)})
print("far fraction by ring size (src < batch_end + 8 - WIN):", {k: "%.1f%%" % (100 * v / nm) for k, v in fars.items()})
print("in-batch dependent matches %.1f%%; watermark rounds after the first pass: mean %.2f p90 %d max %d" % (100 * inbatch / nm, np.mean(rounds), np.percentile(rounds, 90), max(rounds)))
