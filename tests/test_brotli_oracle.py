"""The brotli restatement (oracle/brotli_oracle.c) against (a) the committed streams the
reference's brotli-mt build wrote (tests/golden/brotli), (b) the image's libbrotli 1.0.9 -- the
library the reference's brotli-mt sources call into -- on live streams, corrupted streams and all
121 dictionary transforms.  CPU only."""
import json
import os
import random
import struct

import pytest

import helpers as H
from golden import cases

GDIR = os.path.join(H.GOLDEN_DIR, "brotli")
with open(os.path.join(GDIR, "manifest.json")) as f:
    MAN = json.load(f)["cases"]

needs_lib = pytest.mark.skipif(not H.have_libbrotli(), reason="libbrotli 1.0.9 not in this image")
needs_ref = pytest.mark.skipif(not H.have_bref(), reason="oracle/_ref/libbrotlimt_ref.so not built")


def golden_stream(name):
    ent = MAN[name]
    if "out_file" in ent:
        with open(os.path.join(GDIR, ent["out_file"]), "rb") as f:
            return f.read()
    if not H.have_bref():
        pytest.skip("stream too large to commit; needs the reference build")
    level, chunk, thunk = cases.BCASES[name]
    rv, st, _, _ = H.brotlimt_compress_via(H.bref(), thunk(), chunk, threads=2, level=level)
    assert rv == 0 and H.sha256(st) == ent["out_sha256"]
    return st


@pytest.mark.parametrize("name", sorted(MAN))
def test_golden_streams_decode(name):
    ent = MAN[name]
    st = golden_stream(name)
    assert len(st) == ent["out_len"] and H.sha256(st) == ent["out_sha256"]
    out = H.oracle_brotlimt_decompress(st, ent["in_len"] + 16)
    assert out is not None and len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]


def test_case_generators_match_manifest():
    for name, (_, _, thunk) in cases.BCASES.items():
        d = thunk()
        assert len(d) == MAN[name]["in_len"] and H.sha256(d) == MAN[name]["in_sha256"], name


@needs_ref
@pytest.mark.parametrize("name", ["b_empty", "b_text_3x128k", "b_english_chunks", "b_mixed_l7"])
def test_reference_decoder_agrees(name):
    st = golden_stream(name)
    rv, back, _, stats = H.brotlimt_decompress_via(H.bref(), st, threads=2)
    assert rv == 0
    assert H.oracle_brotlimt_decompress(st, len(back) + 16) == back
    assert stats[0] == MAN[name]["frames"] and stats[1] == len(st) and stats[2] == len(back)


LIVE = {
    "empty": lambda: b"",
    "one": lambda: b"a",
    "text": lambda: cases.text(90000, 21),
    "english": lambda: cases.english(90000, 22),
    "zeros": lambda: bytes(70000),
    "random": lambda: cases.rnd(40000, 23),
    "soup": lambda: H.soup(random.Random(24), 120000),
}


@needs_lib
@pytest.mark.parametrize("kind", sorted(LIVE))
@pytest.mark.parametrize("quality", [0, 1, 2, 4, 5, 6, 9, 10, 11])
def test_live_streams(kind, quality):
    d = LIVE[kind]()
    for lgwin in (24, 16, 10):
        st = H.libbrotli_compress(d, quality, lgwin)
        assert H.oracle_brotli_decompress(st, len(d)) == d, (kind, quality, lgwin)


@needs_lib
def test_transforms_match_library():
    import ctypes as C
    com = C.CDLL("/opt/conda/lib/libbrotlicommon.so.1")
    com.BrotliGetTransforms.restype = C.c_void_p
    com.BrotliTransformDictionaryWord.restype = C.c_int
    com.BrotliTransformDictionaryWord.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    tr = com.BrotliGetTransforms()
    blob = H.brotli_blob()
    off_dict = struct.unpack_from("<I", blob, 8)[0]
    bits = [0, 0, 0, 0, 10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5]
    rng = random.Random(1)
    o = 0
    orc = H.oracle()
    for ln in range(4, 25):
        for _ in range(12):
            w = rng.randrange(1 << bits[ln])
            word = blob[off_dict + o + w * ln: off_dict + o + (w + 1) * ln]
            for t in range(121):
                a, b = C.create_string_buffer(64), C.create_string_buffer(64)
                na = com.BrotliTransformDictionaryWord(a, word, ln, tr, t)
                nb = orc.zo_brotli_transform(blob, b, word, ln, t)
                assert na == nb and a.raw[:na] == b.raw[:nb], (ln, w, t)
        o += ln << bits[ln]
    assert o == 122784


@needs_lib
@pytest.mark.parametrize("quality", [1, 5, 9])
def test_corrupt_streams_same_verdict(quality):
    """Bit flips: the restatement and BrotliDecoderDecompress accept the same streams with the
    same content, and reject the same streams."""
    d = cases.english(30000, 31) + cases.text(20000, 32)
    st = H.libbrotli_compress(d, quality, 22)
    rng = random.Random(quality)
    for _ in range(250):
        s = bytearray(st)
        for _ in range(rng.choice([1, 1, 2])):
            s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
        a = H.libbrotli_decompress(bytes(s), len(d) + 100)
        b = H.oracle_brotli_decompress(bytes(s), len(d) + 100)
        if a is None:
            assert isinstance(b, int)
        else:
            assert a == b


@needs_lib
def test_truncation_capacity_trailing():
    d = cases.text(50000, 3)
    st = H.libbrotli_compress(d, 5, 22)
    assert H.libbrotli_decompress(st, len(d)) == d and H.oracle_brotli_decompress(st, len(d)) == d
    assert H.libbrotli_decompress(st, len(d) - 1) is None and H.oracle_brotli_decompress(st, len(d) - 1) == -2
    # bytes after the end of the stream are ignored by the one-shot call
    assert H.libbrotli_decompress(st + b"xyz", len(d)) == d and H.oracle_brotli_decompress(st + b"xyz", len(d)) == d
    for cut in (1, 2, 10, len(st) // 2, len(st) - 1, len(st)):
        assert H.libbrotli_decompress(st[:len(st) - cut], len(d)) is None
        assert H.oracle_brotli_decompress(st[:len(st) - cut], len(d)) == -1


def test_record_walk_errors():
    """lib/brotli-mt_decompress.c:187-284: magic, length field 8, "BR", payload size, hint."""
    payload = bytes([0x06])  # WBITS 16, ISLAST, ISLASTEMPTY: the empty stream
    ok = H.brotli_record(payload, 1)
    assert H.oracle_brotlimt_decompress(ok, 16) == b""
    assert H.oracle_brotlimt_decompress(ok + ok, 16) == b""
    bad = bytearray(ok)
    bad[0] ^= 1
    assert H.oracle_brotlimt_decompress(bytes(bad), 16) is None
    bad = bytearray(ok)
    bad[4] = 4
    assert H.oracle_brotlimt_decompress(bytes(bad), 16) is None
    bad = bytearray(ok)
    bad[12] = 0x43
    assert H.oracle_brotlimt_decompress(bytes(bad), 16) is None
    assert H.oracle_brotlimt_decompress(ok[:-1], 16) is None
    assert H.oracle_brotlimt_decompress(ok + ok[:10], 16) is None
    assert H.oracle_brotlimt_decompress(b"", 16) is None


def test_hint_is_the_capacity():
    """The decoder sizes its buffer from the hint alone (hint << 16, lib/brotli-mt_decompress.c:
    236-239): a chunk that is not a multiple of 64 KiB written with hint = chunk >> 16 cannot be
    decoded.  Raw uncompressed meta-block streams keep this test independent of an encoder."""
    d = bytes(range(256)) * 300  # 76 800 bytes
    # WBITS=16 ("0"), ISLAST=0, MNIBBLES=5 (code 1): MLEN-1 in 20 bits, ISUNCOMPRESSED=1, pad
    mlen1 = len(d) - 1
    bits, nb = 0, 0
    for v, n in ((0, 1), (0, 1), (1, 2), (mlen1, 20), (1, 1)):
        bits |= v << nb
        nb += n
    hdr = bits.to_bytes((nb + 7) // 8, "little")
    st = hdr + d + bytes([0x03])  # then ISLAST=1, ISLASTEMPTY=1
    assert H.oracle_brotli_decompress(st, len(d)) == d
    assert H.oracle_brotlimt_decompress(H.brotli_record(st, 2), len(d)) == d
    assert H.oracle_brotlimt_decompress(H.brotli_record(st, 1), len(d)) is None
