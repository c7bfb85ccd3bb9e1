"""Design-time simulation of the line-oriented LZ4 copy stage ("copy5") on the BENCH text.

A lane owns one 16-byte line of the batch's output and assembles it from the pieces (literal / match segments cut at
line boundaries) that overlap it.  What this prints, per batch as copy3 cuts it (<= 64 small sequences, <= 2 KiB of
output, inside one lap of a 4 KiB ring):
  - pieces per line (mean, and the wave-wide maximum = the piece loop's trip count with no dependencies at all);
  - dependency passes, scheme A1: a pass ends with the completed lines written to the ring; a line advances through
    its pieces until one's source is not in the ring yet;
  - loop steps, scheme A2: lockstep piece loop, a lane whose next piece's source is not complete idles that step,
    completed lines are visible to the next step (one ballot per step);
  - the dependency depth of the sequence DAG (what any exact scheme is bounded by).
Kill criteria written before the run (VERDICT): > 400 wave-instructions per 680 output bytes or > 2.5 average passes.
Instruction model: see COST below.  Its per-step figure (28) is kind: the piece loop written out and compiled for gfx950
(tools/ubench/line_copy_step.hip) is 170 instructions per step (profiles/r05_sweeps/copy5_sim.txt)."""
import ctypes as C, os, struct, sys
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import helpers as H
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 2 << 20
LINE = int(sys.argv[2]) if len(sys.argv) > 2 else 16
buf = np.empty(n, np.uint8)
T.zmt_gen_text(buf.ctypes.data, n, 20260926, 0, 8)
s = H.oracle_compress(buf.tobytes(), 131072)

# instruction model per batch (wave-instructions): fixed part + per-step part
COST = dict(fields=300,   # fields, scan, cut, far loads, flush, loop bookkeeping: copy3's measured "rest" (ablation table)
            owner=30,     # owner lookup per 64 lines: first-of-line scatter, ballot, ffs, table read
            line_io=8,    # read old line, write line (+ mirror)
            step=28)      # one piece: table read, source address, unaligned 16-byte read, threshold merge, cursor, loop


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
st = dict(batches=0, seqs=0, out=0, lines=0, pieces=0, maxpieces=[], passesA1=[], stepsA2=[], depth=[], liters=[],
          stepsA1=[])


def close(cur):
    if not cur: return
    o0 = cur[0][5]; oend = cur[-1][5] + cur[-1][0] + cur[-1][1]
    l0 = o0 // LINE; nl = (oend - 1) // LINE - l0 + 1
    # pieces per line: (dst_lo, dst_hi, src_lo or -1 for literal) in stream order
    lines = [[] for _ in range(nl)]
    # per-sequence DAG depth (byte granular)
    depth_of = np.zeros(oend - o0, np.int32)
    for (l, m, o, q, qe, op) in cur:
        for (a, b, src) in ((op, op + l, -1), (op + l, op + l + m, op + l - o)):
            x = a
            while x < b:
                e = min(b, (x // LINE + 1) * LINE)
                lines[x // LINE - l0].append((x, e, -1 if src < 0 else src + (x - a), o if src >= 0 else 0))
                x = e
        # depth: a match byte's depth = 1 + depth of its source byte if inside the batch
        mp = op + l
        for x in range(mp, mp + m):
            sx = x - o
            depth_of[x - o0] = depth_of[sx - o0] + 1 if sx >= o0 else 0
    st['batches'] += 1; st['seqs'] += len(cur); st['out'] += oend - o0; st['lines'] += nl
    st['pieces'] += sum(len(x) for x in lines); st['maxpieces'].append(max(len(x) for x in lines))
    st['depth'].append(int(depth_of.max()) + 1); st['liters'].append((nl + 63) // 64)
    # scheme A1: passes; A2: lockstep steps.  Availability of a source range [a, b): every byte < o0, or in a done line.
    # (a match piece with offset < its length inside one line is treated as needing its own line's earlier bytes:
    # overlapping pieces are handled by a special path -- counted as "ovl")
    def ready(a, b, done, self_line, upto):
        if b <= o0: return True
        for ln in range(max(a, o0) // LINE - l0, (b - 1) // LINE - l0 + 1):
            if ln == self_line:
                # own line: bytes below the cursor are assembled in registers; allow (register forwarding path)
                continue
            if not done[ln]: return False
        return True
    # A1
    cursor = [0] * nl; done = [False] * nl; passes = 0; steps1 = 0
    while not all(done):
        passes += 1; newdone = []; most = 0
        for j in range(nl):
            if done[j]: continue
            k = 0
            while cursor[j] < len(lines[j]):
                (x, e, src, off) = lines[j][cursor[j]]
                if src >= 0 and not ready(src, src + (e - x), done, j, x): break
                cursor[j] += 1; k += 1
            most = max(most, k + 1)
            if cursor[j] == len(lines[j]): newdone.append(j)
        for j in newdone: done[j] = True
        steps1 += most
        if passes > 200: raise SystemExit("A1 does not converge")
    st['passesA1'].append(passes); st['stepsA1'].append(steps1)
    # A2
    cursor = [0] * nl; done = [False] * nl; steps = 0
    while not all(done):
        steps += 1; newdone = []
        for j in range(nl):
            if done[j]: continue
            (x, e, src, off) = lines[j][cursor[j]]
            if src < 0 or ready(src, src + (e - x), done, j, x):
                cursor[j] += 1
                if cursor[j] == len(lines[j]): newdone.append(j)
        for j in newdone: done[j] = True
        if steps > 2000: raise SystemExit("A2 does not converge")
    st['stepsA2'].append(steps)


for blocks in records(s):
    opos = 0
    for seqs in blocks:
        if seqs is None: continue
        cur = []
        for (l, m, o, q, qe) in seqs:
            small = m != 0 and l <= CAPL and m <= CAPM
            oe = opos + l + m
            if small and ((oe - 1) >> 12) != (opos >> 12): small = False
            if not small:
                close(cur); cur = []
            else:
                if cur:
                    g0 = cur[0][3] & ~15
                    fits = len(cur) < 64 and qe - g0 <= 1016 and oe - cur[0][5] <= XOUT and ((oe - 1) >> 12) == (cur[0][5] >> 12)
                    if not fits: close(cur); cur = []
                cur.append((l, m, o, q, qe, opos))
            opos = oe
        close(cur)

B = st['batches']
print("line = %d bytes; %d batches, %.1f sequences and %.0f output bytes per batch, %.1f lines per batch (%.2f line iterations of 64)" % (
    LINE, B, st['seqs'] / B, st['out'] / B, st['lines'] / B, np.mean(st['liters'])))
print("pieces per line: mean %.2f; wave-wide maximum per batch: mean %.2f p90 %d max %d" % (
    st['pieces'] / st['lines'], np.mean(st['maxpieces']), np.percentile(st['maxpieces'], 90), max(st['maxpieces'])))
print("byte-granular dependency depth per batch: mean %.2f p90 %d max %d" % (np.mean(st['depth']), np.percentile(st['depth'], 90), max(st['depth'])))
print("scheme A1: passes per batch mean %.2f p90 %d max %d; piece-loop steps summed over the passes: mean %.2f" % (
    np.mean(st['passesA1']), np.percentile(st['passesA1'], 90), max(st['passesA1']), np.mean(st['stepsA1'])))
print("scheme A2: lockstep steps per batch mean %.2f p90 %d max %d" % (np.mean(st['stepsA2']), np.percentile(st['stepsA2'], 90), max(st['stepsA2'])))
per = lambda steps: COST['fields'] + np.mean(st['liters']) * (COST['owner'] + COST['line_io']) + steps * COST['step']
outb = st['out'] / B
for name, steps in (("no dependencies (lower bound)", np.mean(st['maxpieces'])), ("A1", np.mean(st['stepsA1'])), ("A2", np.mean(st['stepsA2']))):
    c = per(steps)
    print("model %-30s %.0f wave-instructions per batch = %.0f per 680 output bytes (copy3 measured: 692)" % (name, c, c * 680 / outb))
