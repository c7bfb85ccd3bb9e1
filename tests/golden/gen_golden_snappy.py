#!/usr/bin/env python3
"""Generate tests/golden/snappy/manifest.json (+ *.spmt): snappy-mt streams whose payloads were written
by the image's libsnappy 1.1.8 (snappy-c.h API), framed as lib/snappy-mt_compress.c:280-300 does.

The reference's own snappy library (a C port the zstdmt repository vendors) is not part of
/root/reference, so no reference build exists for this codec; these vectors pin the *format* -- what
any snappy decoder must make of them -- with an independent implementation.  The outputs are data:
streams (decoder inputs) and the SHA-256 of the plaintext each must decode to.  Inputs come from the
generators in cases.py.  Run in the build container (needs /opt/conda/lib/libsnappy.so.1).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402
import cases  # noqa: E402

FILE_MAX = 48 * 1024

SCASES = {
    "s_empty": (65536, lambda: b""),
    "s_one": (65536, lambda: b"x"),
    "s_hello": (65536, lambda: b"hello hello hello hello, hello!"),
    "s_text_60": (65536, lambda: cases.text(60, 1)),              # literal of exactly 60 bytes: one-byte tag
    "s_text_61": (65536, lambda: cases.text(61, 1)),              # 61: the length moves to a byte of its own
    "s_text_20k": (65536, lambda: cases.text(20000, 2)),
    "s_text_chunks_4k": (4096, lambda: cases.text(30000, 3)),     # eight records, ragged last one
    "s_zeros_70k": (65536, lambda: bytes(70000)),                 # overlapping copies (offset 1), two records
    "s_period_300": (65536, lambda: cases.rep(cases.rnd(300, 9), 40000)),
    "s_random_5k": (65536, lambda: cases.rnd(5000, 4)),           # stored as one long literal
    "s_mixed": (16384, lambda: cases.text(9000, 5) + bytes(5000) + cases.rnd(2000, 6) + cases.text(12000, 7)),
    "s_text_150k_default": (65536, lambda: cases.text(150000, 8)),  # digests only (too large to commit)
}


def main():
    assert H.have_libsnappy()
    out = os.path.join(HERE, "snappy")
    os.makedirs(out, exist_ok=True)
    man = {"_generator": "tests/golden/gen_golden_snappy.py",
           "_writer": "libsnappy 1.1.8 (snappy_compress of snappy-c.h) + the record framing of lib/snappy-mt_compress.c:280-300",
           "cases": {}}
    for name, (chunk, thunk) in SCASES.items():
        data = thunk()
        st = H.snappymt_stream(data, chunk)
        parts = []
        at = 0
        while at < len(st):   # every payload decodes with libsnappy itself
            csz = int.from_bytes(st[at + 8:at + 12], "little")
            parts.append(H.libsnappy_decompress(st[at + 16:at + 16 + csz], chunk))
            at += 16 + csz
        assert b"".join(parts) == data, name
        frames = max(1, -(-len(data) // chunk))
        ent = {"chunk": chunk, "in_len": len(data), "in_sha256": H.sha256(data), "out_len": len(st),
               "out_sha256": H.sha256(st), "frames": frames}
        if len(st) <= FILE_MAX:
            fn = name + ".spmt"
            with open(os.path.join(out, fn), "wb") as f:
                f.write(st)
            ent["out_file"] = fn
        man["cases"][name] = ent
        print(f"{name:24s} in={len(data):8d} out={len(st):8d} frames={frames}")
    with open(os.path.join(out, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
