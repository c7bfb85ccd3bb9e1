"""oracle/snappy_oracle.c (the CPU restatement of the raw snappy format behind snappy-mt) against the
committed vectors (tests/golden/snappy: payloads written by libsnappy 1.1.8) and, where the image's
libsnappy is present, against that library both ways: what it writes decodes to the input, and on
damaged streams the verdict and the bytes are snappy_uncompress's."""
import json
import os
import random

import pytest

import helpers as H
from golden import cases

SDIR = os.path.join(H.GOLDEN_DIR, "snappy")
MAN = json.load(open(os.path.join(SDIR, "manifest.json")))["cases"]


def varint(v):
    out = bytearray()
    while v >= 128:
        out.append(v & 127 | 128)
        v >>= 7
    out.append(v)
    return bytes(out)


def lit(b, nbytes=None):
    """A literal element; nbytes forces the 1..4-byte length form."""
    n = len(b) - 1
    if nbytes is None and n < 60:
        return bytes([n << 2]) + b
    nbytes = nbytes or max(1, (n.bit_length() + 7) // 8)
    return bytes([(59 + nbytes) << 2]) + n.to_bytes(nbytes, "little") + b


def copy1(length, off):
    return bytes([1 | (length - 4) << 2 | (off >> 8) << 5, off & 255])


def copy2(length, off):
    return bytes([2 | (length - 1) << 2]) + off.to_bytes(2, "little")


def copy4(length, off):
    return bytes([3 | (length - 1) << 2]) + off.to_bytes(4, "little")


# element forms no encoder here writes (4-byte offsets, long length fields, every overlap period)
HAND = {
    "copy4": (varint(20) + lit(b"abcdefgh") + copy4(12, 8), b"abcdefgh" + b"abcdefghabcd"),
    "copy1_max_offset": (varint(2047 + 11) + lit(bytes(range(256)) * 7 + bytes(255), 2) + copy1(11, 2047),
                         (bytes(range(256)) * 7 + bytes(255)) + (bytes(range(256)) * 7 + bytes(255))[:11]),
    "literal_len_2_3_4_bytes": (varint(3 * 70) + lit(b"q" * 70, 2) + lit(b"r" * 70, 3) + lit(b"s" * 70, 4),
                                b"q" * 70 + b"r" * 70 + b"s" * 70),
    "overlap_periods": (varint(3 + 64 + 5 + 64) + lit(b"abc") + copy2(64, 3) + lit(b"vwxyz") + copy2(64, 5),
                        (b"abc" * 23)[:67] + b"vwxyz" + (b"vwxyz" * 13)[:64]),
    "overlap_period_1": (varint(1 + 64 + 64) + lit(b"z") + copy2(64, 1) + copy1(4, 1) + copy2(60, 60), b"z" * 129),
    "copy_len_1": (varint(5) + lit(b"wxyz") + copy2(1, 4), b"wxyzw"),
}

BAD = {
    "empty_input": b"",
    "preamble_only_but_nonzero": varint(5),
    "preamble_unterminated": b"\xff\xff\xff\xff\xff",
    "preamble_33_bits": b"\xff\xff\xff\xff\x1f",
    "offset_zero": varint(8) + lit(b"abcd") + copy2(4, 0),
    "offset_beyond_output": varint(8) + lit(b"abcd") + copy2(4, 5),
    "literal_truncated": varint(8) + bytes([7 << 2]) + b"abc",
    "copy_truncated": varint(8) + lit(b"abcd") + bytes([2 | 3 << 2, 4]),
    "too_short": varint(9) + lit(b"abcd") + copy2(4, 4),
    "too_long": varint(7) + lit(b"abcd") + copy2(4, 4),
    "literal_past_size": varint(3) + lit(b"abcd"),
    "trailing_element": varint(4) + lit(b"abcd") + lit(b"e"),
}


@pytest.mark.parametrize("name", sorted(MAN))
def test_oracle_decodes_golden(name):
    ent = MAN[name]
    if "out_file" in ent:
        st = open(os.path.join(SDIR, ent["out_file"]), "rb").read()
    elif H.have_libsnappy():
        from gen_golden_snappy import SCASES
        st = H.snappymt_stream(SCASES[name][1](), ent["chunk"])
    else:
        pytest.skip("stream not committed (size) and no libsnappy on this box")
    assert len(st) == ent["out_len"] and H.sha256(st) == ent["out_sha256"]
    out = H.oracle_snappymt_decompress(st, ent["in_len"] + 64)
    assert out is not None and len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]


@pytest.mark.parametrize("name", sorted(HAND))
def test_oracle_hand_built_elements(name):
    st, want = HAND[name]
    assert H.oracle_snappy_decompress(st, len(want)) == want
    assert H.oracle_snappy_decompress(st, len(want) - 1) is None        # capacity below the preamble
    if H.have_libsnappy():
        assert H.libsnappy_decompress(st, len(want)) == want


@pytest.mark.parametrize("name", sorted(BAD))
def test_oracle_rejects(name):
    assert H.oracle_snappy_decompress(BAD[name], 4096) is None
    if H.have_libsnappy() and BAD[name]:
        assert H.libsnappy_decompress(BAD[name], 4096) is None


def test_record_walk_rejects():
    st = H.snappy_record(varint(3) + lit(b"abc"), 1)
    assert H.oracle_snappymt_decompress(st, 64) == b"abc"
    assert H.oracle_snappymt_decompress(st + st, 64) == b"abcabc"
    assert H.oracle_snappymt_decompress(st[:-1], 64) is None
    assert H.oracle_snappymt_decompress(st + b"\0", 64) is None
    for at, v in ((0, 0x51), (4, 9), (12, 0x42), (13, 0x52)):
        bad = bytearray(st)
        bad[at] = v
        assert H.oracle_snappymt_decompress(bytes(bad), 64) is None
    bad = bytearray(st)                     # the hint is not read (lib/snappy-mt_decompress.c:234-238)
    bad[14:16] = b"\0\0"
    assert H.oracle_snappymt_decompress(bytes(bad), 64) == b"abc"


@pytest.mark.skipif(not H.have_libsnappy(), reason="libsnappy not on this box")
@pytest.mark.parametrize("seed", range(40))
def test_oracle_equals_libsnappy_on_damaged_streams(seed):
    rng = random.Random(7000 + seed)
    n = rng.choice([1, 5, 59, 60, 61, 300, 4096, 70000, rng.randrange(1, 150000)])
    data = H.soup(rng, n) if seed % 4 == 0 else rng.choice([cases.text, cases.rnd])(n, seed)
    st = H.libsnappy_compress(data)
    assert H.oracle_snappy_decompress(st, len(data)) == data
    for _ in range(25):
        bad = bytearray(st)
        for _ in range(rng.randrange(1, 3)):
            k = rng.randrange(len(bad))
            bad[k] = bad[k] ^ (1 << rng.randrange(8)) if rng.random() < 0.6 else rng.randrange(256)
        if rng.random() < 0.1:
            bad = bad[:rng.randrange(1, len(bad) + 1)]
        bad = bytes(bad)
        assert H.oracle_snappy_decompress(bad, len(data) + 70000) == H.libsnappy_decompress(bad, len(data) + 70000)
