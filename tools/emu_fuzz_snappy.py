"""Developer tool.  Emulator fuzz of snappy.hip: the encoder's streams must decode to the input with the
oracle, libsnappy (where present) and the emulated decoder; damaged payloads of both writers must get the
oracle's verdict and bytes from the emulated decoder (python tools/emu_fuzz_snappy.py [first] [last])."""
import random
import sys

sys.path.insert(0, "tests")
sys.path.insert(0, "tests/golden")
import emu_driver as E
import helpers as H
from cases import rnd, text

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 3571 + 17)
    import os
    os.environ["EMU_SNAPPY_DEC"] = str(seed & 1)   # both decoders of snappy.hip
    n = rng.choice([rng.randrange(1, 250000), 65536, 65537, 131072, rng.randrange(1, 600), 0])
    kind = rng.random()
    data = b"" if n == 0 else text(n, seed=rng.randrange(1 << 30)) if kind < 0.5 else H.soup(rng, n) if kind < 0.8 else \
        H.dense_sequences(n, rng.choice([b"aaaabbcd", None])) if kind < 0.9 else rnd(n, seed)
    chunk = rng.choice([4096, 65536, 65536, 100000, 131072, 1 << 20])
    own = rng.random() < 0.6 or not H.have_libsnappy()
    st = E.snappy_compress(data, chunk, grid=rng.choice([1, 2, 5])) if own else H.snappymt_stream(data, chunk)
    ok = H.oracle_snappymt_decompress(st, len(data) + 64) == data
    out, status = E.snappy_decompress(st, grid=rng.choice([1, 3]))
    ok = ok and (status == 0).all() and out == data
    ro, rl, _ = E.walk_snappy_records(st)
    if own and H.have_libsnappy():
        ok = ok and b"".join(H.libsnappy_decompress(st[o:o + m], chunk) or b"" for o, m in zip(ro.tolist(), rl.tolist())) == data
    for _ in range(10):
        r = rng.randrange(len(ro))
        p = bytearray(st[int(ro[r]):int(ro[r]) + int(rl[r])])
        for _ in range(rng.randrange(1, 3)):
            k = rng.randrange(len(p))
            p[k] = p[k] ^ (1 << rng.randrange(8)) if rng.random() < 0.6 else rng.randrange(256)
        if rng.random() < 0.1:
            p = p[:rng.randrange(1, len(p) + 1)]
        p = bytes(p)
        cap = chunk + 4096
        want = H.oracle_snappy_decompress(p, cap)
        o2, s2 = E.snappy_decompress(H.snappy_record(p, 1), caps=[cap])
        good = (s2[0] != 0 and o2 == b"") if want is None else (s2[0] == 0 and o2 == want)
        if not good:
            print(seed, "DAMAGE MISMATCH oracle", "reject" if want is None else "accept", "kernel", s2[0], flush=True)
        ok = ok and good
    print(seed, n, chunk, "own" if own else "lib", "OK" if ok else "FAIL", flush=True)
    bad += not ok
print("mismatches:", bad)
