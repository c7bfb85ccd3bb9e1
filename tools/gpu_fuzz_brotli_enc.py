"""Developer tool: random sizes / chunk sizes / data kinds through the device brotli encoder, every
record decoded by the oracle (test infrastructure)."""
import sys, random
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, R)
import torch
import helpers as H, emu_driver as E
from golden import cases
import zstdmt_amd as z
import numpy as np
eng = z.Engine(0)
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    n = rng.choice([0, 1, 5, 63, 64, 65, 1000, 65535, 65536, 65537, 131071, 131072, 131073, rng.randrange(1, 700000)])
    chunk = rng.choice([1000, 4096, 65536, 100000, 131072, 262144, 1 << 20])
    kind = rng.randrange(6)
    if kind == 0: d = cases.text(n, rng.randrange(1<<30))
    elif kind == 1: d = cases.rnd(n, rng.randrange(1<<30))
    elif kind == 2: d = bytes(n)
    elif kind == 3: d = H.soup(rng, n) if n else b""
    elif kind == 4: d = cases.english(n, rng.randrange(1<<30))
    else: d = bytes((b % 3) + 65 for b in cases.rnd(n, rng.randrange(1<<30)))
    st, ro, rl = eng.compress_bytes(d, chunk, codec="brotli")
    # decode with the oracle using a capacity large enough regardless of the hint quirk
    out = b""
    ok = True
    ip = 0
    import struct
    while ip < len(st):
        magic, eight, csize, br, hint = struct.unpack_from("<IIIHH", st, ip)
        r = H.oracle_brotli_decompress(st[ip+16:ip+16+csize], chunk + 16)
        if isinstance(r, int): ok = False; break
        out += r; ip += 16 + csize
    if not ok or out != d:
        bad += 1; print("FAIL", it, n, chunk, kind)
print("done, failures:", bad)
