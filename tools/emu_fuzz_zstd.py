"""Developer tool: device zstd encoder -> bit flips -> device decoder (both under the CPU emulator)
against the oracle's verdict and content.  Exercises the unit-wide decode of zstd_dec.hip.
  python tools/emu_fuzz_zstd.py SEED0 SEED1"""
import random
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, "tests/golden")
import numpy as np

import emu_driver as E
import helpers as H
from cases import text

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad_total = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 104729 + 5)
    n = rng.choice([rng.randrange(1, 400000), 131072, 262144 + rng.randrange(0, 100), rng.randrange(100000, 200000)])
    kind = rng.randrange(4)
    if kind == 0:
        data = text(n, seed=rng.randrange(1 << 30))
    elif kind == 1:
        data = H.dense_sequences(n, rng.choice([b"aaaabbcd", None, b"ab"]))
    elif kind == 2:
        data = H.soup(rng, n)
    else:
        t = bytearray(text(n, seed=rng.randrange(1 << 30)))
        for _ in range(rng.randrange(1, 12)):
            if n < 100:
                break
            a = rng.randrange(0, n - 50)
            ln = rng.randrange(1, min(5000, n - a))
            b = rng.randrange(0, n - ln)
            t[b:b + ln] = t[a:a + ln]
        data = bytes(t)
    chunk = rng.choice([131072, 1 << 20, 1 << 20])
    st = E.zstd_compress(data, chunk)
    out, status = E.zstd_decompress(st)
    ok = (status == 0).all() and out == data
    ro, rl = E.walk_records(st)
    fails = []
    for _ in range(8):
        r = rng.randrange(len(rl))
        lo, ln = int(ro[r]), int(rl[r])
        rec = bytearray(st[lo:lo + ln])
        pos = rng.randrange(12, ln)
        rec[pos] ^= 1 << rng.randrange(8)
        rec = bytes(rec)
        cap = min(chunk, len(data)) + 64
        want = H.oracle_zstdmt_decompress(rec, cap)
        o2, s2 = E.zstd_decompress(rec, rec=(np.array([0], np.uint64), np.array([ln], np.uint32)))
        if s2[0] == 7 and want is not None:
            continue   # the flip removed the content size: the probe leaves such frames to the caller (capacity)
        if (want is None) != (s2[0] != 0) or (want is not None and o2 != want):
            fails.append((r, pos))
    print(seed, kind, n, chunk, "OK" if ok and not fails else "FAIL", fails, flush=True)
    bad_total += (not ok) + len(fails)
sys.exit(1 if bad_total else 0)
