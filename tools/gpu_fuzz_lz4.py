"""GPU fuzz of the LZ4 paths against the oracle (developer tool, run through gpurun):
every level 1..12 on mixed / run-heavy inputs and odd chunk sizes, decode through both variants.
    python tools/gpu_fuzz_lz4.py [seconds] [first seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import helpers as H
import zstdmt_amd as z
from test_oracle_vs_ref import _mix, _runs
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = z.Engine(0)
t0 = time.time()
n_ok = 0
while time.time() - t0 < budget:
    rng = random.Random(500000 + seed)
    level = rng.choice([1, 1, 2, 3, 3, 4, 5, 6, 7, 8, 9, 9, 10, 11, 12])
    big = level < 10
    n = rng.choice([rng.randrange(1, 3_000_000 if big else 400_000), rng.randrange(1, 70_000), 65536, 131072, 131073])
    chunk = rng.choice([65536, 131072, 131072, 100000, 262144, 1 << 20, 70001, 4 << 20]) if big else rng.choice([65536, 131072, 100000, 30000])
    data = _mix(rng, n) if rng.random() < 0.6 else _runs(rng, n)
    want = H.oracle_compress_level(data, chunk, level) if level >= 3 else H.oracle_compress(data, chunk)
    stream, ro, rl = eng.compress_bytes(data, chunk, level=level)
    if stream != want:
        print("ENCODE MISMATCH seed", seed, "level", level, "n", n, "chunk", chunk, flush=True)
        sys.exit(1)
    for v in (0, 1):
        eng.set_variant("lz4_dec", v)
        out, st = eng.decompress_bytes(stream, ro, rl)
        eng.set_variant("lz4_dec", 0)
        if st.any() or out != data:
            print("DECODE MISMATCH seed", seed, "variant", v, "level", level, "n", n, "chunk", chunk, flush=True)
            sys.exit(1)
    n_ok += 1
    seed += 1
print(f"{n_ok} cases ok in {time.time() - t0:.0f} s (next seed {seed})")
