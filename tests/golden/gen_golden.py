#!/usr/bin/env python3
"""Generate tests/golden/manifest.json (+ streams/) by running the REFERENCE's own lz4-mt code.

Run in the build container only (needs oracle/_ref/liblz4mt_ref.so, i.e. `make -C oracle ref`,
which compiles /root/reference/lib/lz4-mt_*.c in place against the image's liblz4).  The outputs
are data (expected streams / digests); no reference source is stored.

For every case the reference is run at T=1 and T=4 (outputs must agree: SURVEY.md Appendix C) and
its own decompressor must round-trip the stream.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402
from cases import CASES  # noqa: E402

INLINE_MAX = 2048      # streams up to this size are stored in full
FILE_MAX = 80 * 1024   # up to this size as a file; above: digest only


def main():
    assert H.have_ref(), "build oracle/_ref first: make -C oracle ref"
    ref = H.ref()
    os.makedirs(os.path.join(HERE, "streams"), exist_ok=True)
    man = {"_generator": "tests/golden/gen_golden.py",
           "_reference": "mcmilk/zstdmt lib/lz4-mt_{common,compress,decompress}.c + liblz4 1.9.3",
           "level": 1, "cases": {}}
    for name, (chunk, thunk) in CASES.items():
        data = thunk()
        rv1, s1, _, st1 = H.lz4mt_compress_via(ref, data, chunk, threads=1, level=1)
        rv4, s4, _, st4 = H.lz4mt_compress_via(ref, data, chunk, threads=4, level=1)
        assert rv1 == 0 and rv4 == 0 and s1 == s4, name
        rvd, back, _, dst = H.lz4mt_decompress_via(ref, s1, threads=2)
        assert rvd == 0 and back == data, name
        ent = {"chunk": chunk, "in_len": len(data), "in_sha256": H.sha256(data),
               "out_len": len(s1), "out_sha256": H.sha256(s1),
               "frames": st1[0], "insize": st1[1], "outsize": st1[2],
               "d_insize": dst[1], "d_outsize": dst[2]}
        if len(s1) <= INLINE_MAX:
            ent["out_hex"] = s1.hex()
        elif len(s1) <= FILE_MAX:
            fn = os.path.join("streams", name + ".lz4mt")
            with open(os.path.join(HERE, fn), "wb") as f:
                f.write(s1)
            ent["out_file"] = fn
        man["cases"][name] = ent
        print(f"{name:28s} in={len(data):8d} out={len(s1):8d} frames={st1[0]}")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
