"""Developer tool.  Emulator fuzz of every encoder on small inputs (13..400 bytes with short repeats: chunks
that attempt matches and then may not shrink, ragged tails): LZ4 levels 1, 3, 9, 12 bit-exact against the
oracle; zstd / brotli / snappy decompress-identical (python tools/emu_fuzz_small.py [first] [last])."""
import random
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, "tests/golden")
import emu_driver as E
import helpers as H
from cases import rnd, text

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 2003 + 9)
    n = rng.randrange(1, 400)
    base = bytearray(text(n, seed=rng.randrange(1 << 30)) if rng.random() < 0.6 else rnd(n, rng.randrange(1 << 30)))
    for _ in range(rng.randrange(0, 4)):
        if n < 12:
            break
        a = rng.randrange(0, n - 4)
        ln = rng.randrange(4, min(40, n - a) + 1)
        b = rng.randrange(0, n - ln + 1)
        base[b:b + ln] = base[a:a + ln]
    head = text(rng.choice([0, 0, 65536, 4096]), seed=seed)        # sometimes as the ragged tail of a full chunk
    data = head + bytes(base)
    chunk = 65536 if len(head) in (0, 65536) else 4096
    ok = True
    for lv in (1, 3, 9, 12):
        if lv == 12 and len(data) > 5000:
            continue
        ok = ok and E.compress(data, chunk, lv)[0] == H.oracle_compress_level(data, chunk, lv) if lv >= 3 else \
            ok and E.compress(data, chunk, 1)[0] == H.oracle_compress(data, chunk)
    zc = chunk if chunk == 65536 else 131072
    ok = ok and H.oracle_zstdmt_decompress(E.zstd_compress(data, zc), len(data) + 64) == data
    ok = ok and H.oracle_brotlimt_decompress(E.brotli_compress(data, zc), len(data) + 65536) == data
    ok = ok and H.oracle_snappymt_decompress(E.snappy_compress(data, chunk), len(data) + 64) == data
    if not ok or seed % 50 == 0:
        print(seed, len(data), chunk, "OK" if ok else "FAIL", flush=True)
    bad += not ok
print("mismatches:", bad)
