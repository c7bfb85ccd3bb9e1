"""Developer tool.  Emulator fuzz of the brotli kernels: (a) the device encoder's streams must decode to
the input with the oracle and the emulated decoder; (b) damaged streams of libbrotli (every quality) and
of the device encoder must get the oracle's verdict from the emulated decoder, and the oracle's bytes
where it accepts (python tools/emu_fuzz_brotli.py [first] [last])."""
import random
import struct
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, "tests/golden")
import numpy as np

import emu_driver as E
import helpers as H
from cases import text

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 6151 + 11)
    n = rng.choice([rng.randrange(1, 120000), 65536, rng.randrange(1, 3000), 131072 + rng.randrange(0, 50)])
    kind = rng.random()
    data = text(n, seed=rng.randrange(1 << 30)) if kind < 0.6 else H.soup(rng, n)
    own = rng.random() < 0.4 or not H.have_libbrotli()
    if own:
        chunk = rng.choice([65536, 131072, 1 << 20])
        st = E.brotli_compress(data, chunk, grid=rng.choice([1, 3]))
        ok = H.oracle_brotlimt_decompress(st, len(data) + 64) == data
        recs, status = E.brotli_decompress(st)
        ok = ok and (status == 0).all() and b"".join(recs) == data
        label = "own"
    else:
        q = rng.randrange(0, 12)
        hint = (len(data) >> 16) + 1
        st = H.brotli_record(H.libbrotli_compress(data, quality=q, lgwin=rng.choice([16, 18, 22, 24])), hint)
        recs, status = E.brotli_decompress(st)
        ok = (status == 0).all() and b"".join(recs) == data
        label = "q%d" % q
    ro, rl, cap = E.walk_brotli_records(st)
    for _ in range(8):
        r = rng.randrange(len(ro))
        lo, ln = int(ro[r]), int(rl[r])
        if ln == 0:
            continue
        p = bytearray(st[lo:lo + ln])
        for _ in range(rng.randrange(1, 3)):
            k = rng.randrange(len(p))
            p[k] = p[k] ^ (1 << rng.randrange(8)) if rng.random() < 0.6 else rng.randrange(256)
        if rng.random() < 0.1:
            p = p[:rng.randrange(1, len(p) + 1)]
        rec = H.brotli_record(bytes(p), int(cap[r]) >> 16)
        want = H.oracle_brotlimt_decompress(rec, int(cap[r]) + 64)
        out, s2 = E.brotli_decompress(rec)
        good = (s2[0] != 0) if want is None else (s2[0] == 0 and out[0] == want)
        if not good:
            print(seed, "DAMAGE MISMATCH", label, "oracle", "reject" if want is None else "accept", "kernel status", s2[0], flush=True)
        ok = ok and good
    print(seed, n, label, "OK" if ok else "FAIL", flush=True)
    bad += not ok
print("mismatches:", bad)
