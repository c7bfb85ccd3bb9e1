#!/usr/bin/env python3
"""Generate tests/golden/lz4hc/manifest.json by running the REFERENCE's lz4-mt code at the LZ4 HC
levels (lib/lz4-mt_compress.c:141-146 with level >= 3; liblz4 1.9.3 of the image behind it).
Run in the build container only (needs oracle/_ref/liblz4mt_ref.so: `make -C oracle ref`).
Inputs are the deterministic generators of cases.py; expected outputs are stored as length + SHA-256
(+ hex for tiny streams).  Every stream is decoded back by the reference."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402
from cases import CASES  # noqa: E402

LEVELS = (3, 4, 5, 6, 7, 8, 9, 10, 11, 12)
NAMES = ["empty", "hello_5", "abc_12", "abc_13", "text_100", "text_64k", "text_64k_p1", "text_64k_p20", "text_128k",
         "text_3x128k_p100", "text_200k_chunk100000", "text_300k_chunk64k", "text_1m_chunk1m", "zeros_128k",
         "zeros_70000", "A64k_B64k", "period_300", "period_65535", "period_65536", "lcg_128k", "mixed_text_rnd",
         "mixed_rnd_zero_rnd", "lowentropy_128k"]


def main():
    assert H.have_ref(), "build oracle/_ref first: make -C oracle ref"
    ref = H.ref()
    man = {"_generator": "tests/golden/gen_golden_lz4hc.py",
           "_reference": "mcmilk/zstdmt lib/lz4-mt_{common,compress,decompress}.c + liblz4 1.9.3", "cases": {}}
    for name in NAMES:
        chunk, thunk = CASES[name]
        data = thunk()
        ent = {"chunk": chunk, "in_len": len(data), "in_sha256": H.sha256(data), "levels": {}}
        for lv in LEVELS:
            rv1, s1, _, st1 = H.lz4mt_compress_via(ref, data, chunk, threads=1, level=lv)
            rv4, s4, _, _ = H.lz4mt_compress_via(ref, data, chunk, threads=4, level=lv)
            assert rv1 == 0 and rv4 == 0 and s1 == s4, (name, lv)
            rvd, back, _, _ = H.lz4mt_decompress_via(ref, s1, threads=2)
            assert rvd == 0 and back == data, (name, lv)
            e = {"out_len": len(s1), "out_sha256": H.sha256(s1), "frames": st1[0]}
            if len(s1) <= 256:
                e["out_hex"] = s1.hex()
            ent["levels"][str(lv)] = e
        man["cases"][name] = ent
        print(name, {lv: ent["levels"][str(lv)]["out_len"] for lv in LEVELS})
    with open(os.path.join(HERE, "lz4hc", "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
