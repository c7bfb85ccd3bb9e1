#!/usr/bin/env python3
"""Developer measurement (test infrastructure: uses the reference build under oracle/_ref to WRITE
the input): decode a large brotli-mt stream produced by the reference compressor on the device.
  python tests/perf_brotli_decode.py [MiB] [level] [chunk]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (must precede the native library)
import emu_driver as E
import helpers as H
from golden import cases


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import zstdmt_amd as z
    n = mib << 20
    t = time.time()
    data = cases.text(n, 20260926)
    print(f"gen {time.time()-t:.1f}s", flush=True)
    t = time.time()
    rv, st, _, stats = H.brotlimt_compress_via(H.bref(), data, chunk, threads=64, level=level)
    assert rv == 0
    t_c = time.time() - t
    print(f"reference compress: {t_c:.1f}s ratio {n/len(st):.3f} frames {stats[0]}", flush=True)
    t = time.time()
    rv, back, _, _ = H.brotlimt_decompress_via(H.bref(), st, threads=64)
    t_d = time.time() - t
    print(f"reference decompress (64 threads, python callbacks): {t_d:.2f}s = {n/t_d/1e9:.2f} GB/s", flush=True)
    ro, rl, cap = E.walk_brotli_records(st)
    nrec = len(rl)
    e = z.Engine(0)
    out_off = np.zeros(nrec + 1, np.uint64)
    out_off[1:] = np.cumsum(cap.astype(np.uint64))
    total = int(out_off[nrec])
    d_stream = e.upload(st)
    d_ro, d_rl = e.upload(ro.copy()), e.upload(rl.copy())
    d_oo, d_oc = e.upload(out_off), e.upload(cap.copy())
    d_ol, d_st = e.alloc(nrec * 4), e.alloc(nrec * 4)
    d_out = e.alloc(total + 64)
    for it in range(3):
        e.sync()
        t = time.time()
        e.brotli_decompress(d_stream, d_ro, d_rl, nrec, d_out, d_oo, d_oc, d_ol, d_st)
        e.sync()
        dt = time.time() - t
        print(f"decode {dt*1e3:.2f} ms  {n/dt/1e9:.1f} GB/s out, {(n+len(st))/dt/1e9:.1f} GB/s alg", flush=True)
    if os.environ.get("BPROF"):
        import ctypes as C
        e.set_variant("profile", 7)
        cnt = (C.c_ulonglong * 16)()
        e.L.gpumt_debug_counters(e.h, cnt, 16)
        e.brotli_decompress(d_stream, d_ro, d_rl, nrec, d_out, d_oo, d_oc, d_ol, d_st)
        e.sync()
        e.L.gpumt_debug_counters(e.h, cnt, 16)
        c = list(cnt)
        w = max(c[9], 1)
        nm = ["headers", "cmd", "literals", "distance", "copy-batches", "dict/raw", "rest"]
        print("  Mcycles per wave: " + ", ".join(f"{nm[i]}={c[i]/w/1e6:.2f}" for i in range(7)) +
              f" total={c[8]/w/1e6:.2f} waves={c[9]}")
        e.set_variant("profile", 0)
    status = e.download(d_st, nrec * 4, np.uint32)
    olen = e.download(d_ol, nrec * 4, np.uint32)
    ok = bool((status == 0).all())
    if ok:
        raw = e.download(d_out, total)
        out = b"".join(raw[int(out_off[i]):int(out_off[i]) + int(olen[i])].tobytes() for i in range(nrec))
        ok = out == data
    print("status ok:", bool((status == 0).all()), "content ok:", ok)


if __name__ == "__main__":
    main()
