"""Developer tool.  Emulator fuzz of the zstd sequence pre-pass (zstd_dec_seq.hip): frames of several blocks written by the
reference build (or the image's libzstd), decoded by the emulated pre-pass + frame decoder; then damaged copies, whose
verdict -- and content, where the stream stays valid -- must be the oracle's.
python tools/emu_fuzz_zstd_seq.py [first] [last]"""
import sys, random, struct
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import helpers as H, emu_driver as E, numpy as np
from cases import text, rnd, rep
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
assert H.have_zref(), "needs the reference build (oracle/_ref)"
bad = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 104729 + 11)
    kind = rng.random()
    n = rng.choice([rng.randrange(140000, 700000), 262144 + rng.randrange(0, 100), 1 << 20, rng.randrange(1100000, 1400000)])
    if kind < 0.35:
        data = text(n, seed=rng.randrange(1 << 30))
    elif kind < 0.6:
        data = H.soup(rng, n)
    elif kind < 0.75:
        data = H.dense_sequences(n, rng.choice([b"aaaabbcd", None]))      # more sequences than the region holds
    elif kind < 0.9:
        # few sequences per block: repeat / predefined / RLE table modes
        unit = rnd(rng.choice([300, 5000, 40000]), rng.randrange(1 << 30))
        data = rep(unit, n)
    else:
        data = text(n // 2, seed=3) + rnd(n // 4, 5) + bytes(n - n // 2 - n // 4)
    chunk = rng.choice([1 << 20, 2 << 20, 4 << 20])
    if seed >= 100:
        # more than ZS_NB blocks per frame: several groups, tables carried from one group to the next
        n = rng.randrange(1100000, 2600000)
        chunk = 4 << 20
        k2 = rng.random()
        if k2 < 0.4:
            unit = rnd(rng.choice([300, 5000, 40000, 200000]), rng.randrange(1 << 30))
            data = rep(unit, n)
        elif k2 < 0.7:
            parts, left = [], n
            while left > 0:
                m = min(left, rng.randrange(50000, 400000))
                c = rng.random()
                parts.append(text(m, seed=rng.randrange(1 << 30)) if c < 0.4 else rep(rnd(rng.choice([100, 3000]), rng.randrange(1 << 30)), m) if c < 0.8 else bytes(m))
                left -= m
            data = b"".join(parts)
        else:
            data = text(n, seed=rng.randrange(1 << 30))
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, chunk, threads=2, level=rng.choice([1, 1, 2, 3, 5, 7, 12, 19]))
    ok = rv == 0
    out, status = E.zstd_decompress(st)
    ok = ok and (status == 0).all() and out == data
    recs = []
    at = 0
    while at < len(st):
        c = struct.unpack_from("<I", st, at + 8)[0]
        recs.append((at, 12 + c))
        at += 12 + c
    for _ in range(5):
        ro, rl = rng.choice(recs)
        rec = bytearray(st[ro:ro + rl])
        for _ in range(rng.randrange(1, 3)):
            k = rng.randrange(12, len(rec))
            if rng.random() < 0.6:
                rec[k] ^= 1 << rng.randrange(8)
            else:
                rec[k] = rng.randrange(256)
        rec = bytes(rec)
        want = H.oracle_zstdmt_decompress(rec, chunk + 64)
        out, status = E.zstd_decompress(rec, rec=(np.array([0], np.uint64), np.array([len(rec)], np.uint32)))
        if status[0] == 7 and want is not None:
            continue
        good = (status[0] != 0) if want is None else (status[0] == 0 and out == want)
        if not good:
            print(seed, "DAMAGE MISMATCH record", ro, "oracle", "reject" if want is None else "accept", "kernel status", status[0], flush=True)
        ok = ok and good
    print(seed, n, chunk, "OK" if ok else "FAIL", flush=True)
    bad += not ok
print("mismatches:", bad)
