#!/usr/bin/env python3
"""Generate tests/golden/brotli/manifest.json (+ *.brmt) by running the REFERENCE's own brotli-mt code.

Run in the build container only (needs oracle/_ref/libbrotlimt_ref.so: `make -C oracle ref` compiles
/root/reference/lib/brotli-mt_*.c in place against the image's brotli 1.0.9).  The outputs are data:
streams written by the reference compressor (decoder inputs) and the SHA-256 of the plaintext each
must decode to.  Inputs are defined by the generators in cases.py.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402
from cases import BCASES  # noqa: E402

FILE_MAX = 64 * 1024


def main():
    assert H.have_bref(), "build oracle/_ref first: make -C oracle ref"
    z = H.bref()
    out = os.path.join(HERE, "brotli")
    os.makedirs(out, exist_ok=True)
    man = {"_generator": "tests/golden/gen_golden_brotli.py",
           "_reference": "mcmilk/zstdmt lib/brotli-mt_{common,compress,decompress}.c + brotli 1.0.9",
           "cases": {}}
    for name, (level, chunk, thunk) in BCASES.items():
        data = thunk()
        rv, st, _, stats = H.brotlimt_compress_via(z, data, chunk, threads=2, level=level)
        assert rv == 0, name
        rvd, back, _, dst = H.brotlimt_decompress_via(z, st, threads=2)
        assert rvd == 0 and back == data, name
        ent = {"level": level, "chunk": chunk, "in_len": len(data), "in_sha256": H.sha256(data),
               "out_len": len(st), "out_sha256": H.sha256(st), "frames": stats[0],
               "d_insize": dst[1], "d_outsize": dst[2]}
        if len(st) <= FILE_MAX:
            fn = name + ".brmt"
            with open(os.path.join(out, fn), "wb") as f:
                f.write(st)
            ent["out_file"] = fn
        man["cases"][name] = ent
        print(f"{name:24s} L{level:<2d} in={len(data):8d} out={len(st):8d} frames={stats[0]}")
    with open(os.path.join(out, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
