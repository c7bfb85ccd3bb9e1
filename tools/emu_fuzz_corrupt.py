"""Emulator fuzz of the LZ4 decode pipeline on damaged streams: the pipeline frames + parse3 + copy3 at three ring
sizes (variant 0 | ring << 4) and the frame-serial kernel (variant 1) must give the verdict of the oracle on every record, and the same bytes
where the oracle accepts (developer tool: python tools/emu_fuzz_corrupt.py [first] [last])."""
import sys, random
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import helpers as H, emu_driver as E, numpy as np
from cases import text, rnd
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 104729 + 7)
    n = rng.choice([rng.randrange(1, 200000), 131072, 65536 + rng.randrange(0, 100), rng.randrange(1, 5000)])
    chunk = rng.choice([65536, 131072, 100000])
    data = text(n, seed=rng.randrange(1 << 30)) if rng.random() < 0.7 else rnd(n, rng.randrange(1 << 30))
    s = bytearray(H.oracle_compress(data, chunk))
    for _ in range(rng.randrange(1, 4)):
        k = rng.randrange(12, len(s))
        if rng.random() < 0.5:
            s[k] ^= 1 << rng.randrange(8)
        else:
            s[k] = rng.randrange(256)
    s = bytes(s)
    walk = E.walk_records(s)
    want = H.oracle_decompress(s, max(n, 65536))
    if walk is None:
        print(seed, "record walk rejects it (host side)", "oracle:", "reject" if want is None else "accept", flush=True)
        continue
    res = []
    for v in (0, 1, 13 << 4, 14 << 4):
        out, st = E.decompress(s, v)
        res.append((bool(st.any()), out if not st.any() else None))
    ok = all((r[0] == (want is None)) and (r[0] or r[1] == want) for r in res)
    print(seed, n, chunk, "OK" if ok else "FAIL", "reject" if want is None else "accept", [r[0] for r in res], flush=True)
    bad += not ok
print("mismatches:", bad)
