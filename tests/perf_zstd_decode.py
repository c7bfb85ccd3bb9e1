#!/usr/bin/env python3
"""Developer measurement (test infrastructure: uses the reference build under oracle/_ref to WRITE
the input): decode a large zstd-mt stream produced by the reference compressor on the device.
  python tests/perf_zstd_decode.py [MiB] [level] [chunk]
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
    rv, st, _, stats = H.zstdmt_compress_via(H.zref(), data, chunk, threads=64, level=level)
    assert rv == 0
    print(f"reference compress: {time.time()-t:.1f}s ratio {n/len(st):.3f} frames {stats[0]}", flush=True)
    ro, rl = E.walk_records(st)
    nrec = len(rl)
    e = z.Engine(0)
    if os.environ.get("ZDEC_VARIANT"):
        e.set_variant("zstd_dec", int(os.environ["ZDEC_VARIANT"]))
    if os.environ.get("ZSEQ"):
        e.set_variant("zstd_seq", int(os.environ["ZSEQ"]))     # 1 = no sequence pre-pass
    d_stream = e.upload(st)
    d_ro, d_rl = e.upload(ro.copy()), e.upload(rl.copy())
    d_ol, d_oo, d_st = e.alloc(nrec * 4), e.alloc((nrec + 1) * 8), e.alloc(nrec * 4)
    e.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
    total = int(e.download(d_oo, (nrec + 1) * 8, np.uint64)[nrec])
    assert total == n
    d_out = e.alloc(total + 64)
    for it in range(3):
        e.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
        e.sync()
        t = time.time()
        e.zstd_decompress(d_stream, len(st), d_ro, d_rl, nrec, d_out, total, d_oo, d_ol, d_st)
        e.sync()
        dt = time.time() - t
        print(f"decode {dt*1e3:.2f} ms  {n/dt/1e9:.1f} GB/s out, {(n+len(st))/dt/1e9:.1f} GB/s alg", flush=True)
    if os.environ.get("ZPROF"):
        import ctypes as C
        e.set_variant("profile", 5)
        cnt = (C.c_ulonglong * 16)()
        e.L.gpumt_debug_counters(e.h, cnt, 16)
        e.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
        e.zstd_decompress(d_stream, len(st), d_ro, d_rl, nrec, d_out, total, d_oo, d_ol, d_st)
        e.sync()
        e.L.gpumt_debug_counters(e.h, cnt, 16)
        c = list(cnt)
        w = max(c[9], 1)
        nm = ["hdr+huftab", "huffman", "seqhdr+tables", "stage+fse", "exec-lit", "exec-match", "other"]
        print("  cycles per record (M): " + ", ".join(f"{nm[i]}={c[i]/w/1e6:.2f}" for i in range(7)) + f" total={c[8]/w/1e6:.2f}")
        e.set_variant("profile", 0)
    status = e.download(d_st, nrec * 4, np.uint32)
    out = e.download(d_out, total)
    print("status ok:", bool((status == 0).all()), "content ok:", out.tobytes() == data)


if __name__ == "__main__":
    main()
