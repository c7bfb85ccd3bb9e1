"""Developer tool.  Emulator fuzz of the zstd kernels: (a) the device encoder's streams must decode to the input with
the oracle, the emulated decoder and -- where built -- the reference library; (b) damaged streams of
both writers must get the oracle's verdict from the emulated decoder, and the oracle's bytes where it
accepts (developer tool: python tools/emu_fuzz_zstd.py [first] [last])."""
import sys, random
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import helpers as H, emu_driver as E, numpy as np
from cases import text, rnd
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(n0, n1):
    rng = random.Random(seed * 7919 + 3)
    kind = rng.random()
    n = rng.choice([rng.randrange(1, 300000), 131072, 262144 + rng.randrange(0, 100), rng.randrange(1, 9000)])
    if kind < 0.5:
        data = text(n, seed=rng.randrange(1 << 30))
    elif kind < 0.8:
        data = H.soup(rng, n)
    else:
        data = H.dense_sequences(n, rng.choice([b"aaaabbcd", None]))
    chunk = rng.choice([65536, 131072, 1 << 20])
    own = rng.random() < 0.7 or not H.have_zref()
    if own:
        st = E.zstd_compress(data, chunk, grid=rng.choice([1, 3, 8]))
        ok = H.oracle_zstdmt_decompress(st, len(data) + 64) == data
        out, status = E.zstd_decompress(st)
        ok = ok and (status == 0).all() and out == data
        if H.have_zref():
            rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st, threads=2)
            ok = ok and rv == 0 and back == data
    else:
        rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, chunk, threads=2, level=rng.choice([1, 3, 7]))
        ok = rv == 0
    # damage one record
    recs = []
    at = 0
    while at < len(st):
        import struct
        c = struct.unpack_from("<I", st, at + 8)[0]
        recs.append((at, 12 + c))
        at += 12 + c
    flips = 0
    for _ in range(6):
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
            continue   # the damage removed the content size: the probe leaves such frames to the caller (capacity)
        good = (status[0] != 0) if want is None else (status[0] == 0 and out == want)
        if not good:
            print(seed, "DAMAGE MISMATCH record", ro, "oracle", "reject" if want is None else "accept", "kernel status", status[0], flush=True)
        ok = ok and good
        flips += 1
    print(seed, n, chunk, "own" if own else "ref", "OK" if ok else "FAIL", flush=True)
    bad += not ok
print("mismatches:", bad)
