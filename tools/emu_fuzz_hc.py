"""Developer tool.  Emulator fuzz of lz4_enc_hc.hip: every level 3..12 must give the oracle's bytes (which
are the reference's: tests/test_oracle_vs_ref.py) on soups, texts with far repeats and byte runs
(python tools/emu_fuzz_hc.py [first] [last])."""
import random
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, "tests/golden")
import emu_driver as E
import helpers as H
from cases import rnd, text

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 4099 + 5)
    level = rng.choice([3, 4, 5, 6, 7, 8, 9, 9, 10, 11, 12])
    n = rng.choice([rng.randrange(1, 40000), rng.randrange(1, 3000), 65536 + rng.randrange(0, 50), rng.randrange(60000, 140000) if level < 10 else rng.randrange(1, 30000)])
    kind = rng.randrange(5)
    if level >= 10 and n > 30000:
        n = 30000   # the optimal parser on 64 KiB of low-entropy bytes runs for many minutes on the emulator
    if kind == 0:
        data = H.soup(rng, n)
    elif kind == 1:
        data = text(n, seed=rng.randrange(1 << 30))
    elif kind == 2:
        k = rng.choice([1, 2, 3, 5])
        data = bytes((b % k) + 65 for b in rnd(n, rng.randrange(1 << 30)))
    elif kind == 3:
        t = bytearray(text(n, seed=rng.randrange(1 << 30)))
        for _ in range(rng.randrange(1, 15)):
            if n < 100:
                break
            a = rng.randrange(0, n - 50)
            ln = rng.randrange(1, min(3000, n - a))
            b = rng.randrange(0, n - ln)
            t[b:b + ln] = t[a:a + ln] if rng.random() < 0.7 else bytes([rng.randrange(256)]) * ln
        data = bytes(t)
    else:
        p = rnd(rng.randrange(1, 40), rng.randrange(1 << 30))
        data = (p * (n // len(p) + 1))[:n]
    chunk = rng.choice([65536, 131072, 100000])
    got = E.compress(data, chunk, level)[0]
    want = H.oracle_compress_level(data, chunk, level)
    ok = got == want
    print(seed, level, kind, n, chunk, "OK" if ok else "FAIL", flush=True)
    bad += not ok
print("mismatches:", bad)
