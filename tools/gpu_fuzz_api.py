"""GPU fuzz of the four drop-in APIs (developer tool, run through gpurun): random inputs, chunk sizes,
thread counts (1 = the inline decompress path) and levels; every stream must decode back through this
library and, where oracle/_ref is present, through the reference library; and what the reference library writes
(zstd, brotli) must decode through this one.
    python tools/gpu_fuzz_api.py [seconds] [first seed]"""
import ctypes as C, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import helpers as H
from zstdmt_amd._native import lib_path
from test_oracle_vs_ref import _mix, _runs
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = C.CDLL(lib_path())
lz = H.bind_lz4mt(lib)
zs = H.bind_lz4mt(lib, "ZSTDCB_")
br = H.bind_lz4mt(lib, "BROTLIMT_")
sn = H.bind_lz4mt(lib, "SNAPPYMT_")
ref_l = H.ref() if H.have_ref() else None
ref_z = H.zref() if H.have_zref() else None
ref_b = H.bref() if H.have_bref() else None
t0 = time.time(); n_ok = 0
while time.time() - t0 < budget:
    rng = random.Random(770000 + seed)
    n = rng.choice([0, 1, rng.randrange(1, 100_000), rng.randrange(1, 6_000_000), 1 << 20, (1 << 20) + 1])
    data = _mix(rng, n) if rng.random() < 0.7 else _runs(rng, n)
    th = rng.choice([1, 2, 5])
    codec = rng.choice(["lz4", "zstd", "brotli", "snappy"])
    if codec == "lz4":
        chunk = rng.choice([65536, 131072, 100000, 1 << 20, 0]); level = rng.choice([1, 1, 3, 5, 9])
        rv, s, _, _ = H.lz4mt_compress_via(lz, data, chunk, threads=th, level=level)
        assert rv == 0, (seed, "lz4 compress", rv)
        want = H.oracle_compress_level(data, chunk or (4 << 20), level) if level >= 3 else H.oracle_compress(data, chunk or (4 << 20))
        assert s == want, (seed, "lz4 bytes", level, chunk, n)
        rv, out, _, _ = H.lz4mt_decompress_via(lz, s, threads=th)
        assert rv == 0 and out == data, (seed, "lz4 decompress")
    elif codec == "zstd":
        chunk = rng.choice([131072, 1 << 20, 300000, 0]); level = rng.choice([1, 3, 19])
        rv, s, _, _ = H.lz4mt_compress_via(zs, data, chunk, threads=th, level=level, pfx="ZSTDCB_")
        assert rv == 0, (seed, "zstd compress", rv)
        rv, out, _, _ = H.lz4mt_decompress_via(zs, s, threads=th, pfx="ZSTDCB_")
        assert rv == 0 and out == data, (seed, "zstd decompress")
        if ref_z is not None:
            rv, out, _, _ = H.lz4mt_decompress_via(ref_z, s, threads=2, pfx="ZSTDCB_")
            assert rv == 0 and out == data, (seed, "zstd reference decompress")
            # and what the reference writes decodes here (frames of several blocks take the sequence pre-pass)
            rv, rs, _, _ = H.lz4mt_compress_via(ref_z, data, chunk, threads=2, level=rng.choice([1, 2, 3, 7, 19]), pfx="ZSTDCB_")
            assert rv == 0, (seed, "zstd reference compress", rv)
            rv, out, _, _ = H.lz4mt_decompress_via(zs, rs, threads=th, pfx="ZSTDCB_")
            assert rv == 0 and out == data, (seed, "zstd foreign decompress")
    elif codec == "snappy":
        chunk = rng.choice([65536, 4096, 1 << 20, 100000, 0])
        rv, s, _, _ = H.lz4mt_compress_via(sn, data, chunk, threads=th, level=0, pfx="SNAPPYMT_")
        assert rv == 0, (seed, "snappy compress", rv)
        assert H.oracle_snappymt_decompress(s, len(data) + 64) == data, (seed, "snappy oracle decompress")
        rv, out, _, _ = H.lz4mt_decompress_via(sn, s, threads=th, pfx="SNAPPYMT_")
        assert rv == 0 and out == data, (seed, "snappy decompress")
        if H.have_libsnappy() and len(data) < 2_000_000:      # and what libsnappy writes decodes here
            rv, out, _, _ = H.lz4mt_decompress_via(sn, H.snappymt_stream(data, chunk or 65536), threads=th, pfx="SNAPPYMT_")
            assert rv == 0 and out == data, (seed, "snappy foreign decompress")
    else:
        chunk = rng.choice([65536, 1 << 20, 100000, 0]); level = rng.choice([0, 1, 5, 11])
        rv, s, _, _ = H.lz4mt_compress_via(br, data, chunk, threads=th, level=level, pfx="BROTLIMT_")
        assert rv == 0, (seed, "brotli compress", rv)
        rv, out, _, _ = H.lz4mt_decompress_via(br, s, threads=th, pfx="BROTLIMT_")
        assert rv == 0 and out == data, (seed, "brotli decompress")
        if ref_b is not None:
            rv, out, _, _ = H.lz4mt_decompress_via(ref_b, s, threads=2, pfx="BROTLIMT_")
            assert rv == 0 and out == data, (seed, "brotli reference decompress")
            rv, rs, _, _ = H.lz4mt_compress_via(ref_b, data, chunk, threads=2, level=rng.choice([0, 1, 5, 9]), pfx="BROTLIMT_")
            assert rv == 0, (seed, "brotli reference compress", rv)
            # (the reference cannot always read its own streams back -- e.g. level >= 2 at a chunk size that is no multiple of
            # 64 KiB, SURVEY row B2 --: then this library must refuse them too)
            rv0, out0, _, _ = H.lz4mt_decompress_via(ref_b, rs, threads=2, pfx="BROTLIMT_")
            rv, out, _, _ = H.lz4mt_decompress_via(br, rs, threads=th, pfx="BROTLIMT_")
            if rv0 == 0:
                assert rv == 0 and out == data, (seed, "brotli foreign decompress")
            else:
                assert rv != 0, (seed, "brotli foreign stream: the reference refuses it, this library took it")
    n_ok += 1; seed += 1
print(f"{n_ok} cases ok in {time.time() - t0:.0f} s (next seed {seed})")
