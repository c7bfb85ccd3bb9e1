"""zstd decode kernels (zstdmt_amd/csrc/hip/zstd_dec.hip) under the CPU fiber emulator, against the
golden streams the reference wrote and the oracle.  No GPU needed; the GPU run of the same kernels
is tests/test_gpu_zstd.py."""
import json
import os

import numpy as np
import pytest

import emu_driver as E
import helpers as H
from golden import cases

ZDIR = os.path.join(H.GOLDEN_DIR, "zstd")
MAN = json.load(open(os.path.join(ZDIR, "manifest.json")))["cases"]
SMALL = sorted(n for n, e in MAN.items() if "out_file" in e)


def _stream(name):
    return open(os.path.join(ZDIR, MAN[name]["out_file"]), "rb").read()


@pytest.mark.parametrize("name", SMALL)
def test_emu_decode_golden(name):
    ent = MAN[name]
    out, status = E.zstd_decompress(_stream(name))
    assert status.tolist() == [0] * ent["frames"]
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]


@pytest.mark.parametrize("name", ["z_text_3000", "z_hello", "z_text_64k_l1"])
def test_emu_corrupt_streams(name):
    """Bit flips and truncations: status must flag every record whose bytes differ from the
    oracle's verdict; nothing may be written outside the output (checked by the driver)."""
    st = _stream(name)
    n = MAN[name]["in_len"]
    good = H.oracle_zstdmt_decompress(st, n + 64)
    rng = np.random.default_rng(7)
    for pos in sorted(set(rng.integers(12, len(st), 24 if len(st) < 20000 else 12).tolist())):
        bad = bytearray(st)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = H.oracle_zstdmt_decompress(bad, n + 64)
        out, status = E.zstd_decompress(bad, rec=(np.array([0], np.uint64), np.array([len(bad)], np.uint32)))
        if want is None:
            assert status[0] != 0, f"flip at {pos}: oracle rejects, kernel accepted"
        else:
            assert status[0] == 0 and out == want, f"flip at {pos}"
    assert good is not None


def test_emu_probe_rejects():
    st = bytearray(_stream("z_hello"))
    ro, rl = np.array([0], np.uint64), np.array([len(st)], np.uint32)
    for mut, code in (((0, 0x51), 1), ((4, 5), 1), ((8, 99), 1), ((12, 0x29), 2)):
        bad = bytearray(st)
        bad[mut[0]] = mut[1]
        _, status = E.zstd_decompress(bytes(bad), rec=(ro, rl))
        assert status[0] == code


@pytest.mark.skipif(not H.have_zref(), reason="reference build not present")
def test_emu_decode_full_size_blocks():
    """1 MiB frame of 128 KiB blocks (5-byte literal headers, long offsets, a last sequence that
    needs no bits): too large to commit, written by the reference on the spot."""
    level, chunk, thunk = cases.ZCASES["z_text_1m_l1"]
    data = thunk()
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, chunk, threads=2, level=level)
    assert rv == 0
    out, status = E.zstd_decompress(st)
    assert (status == 0).all() and out == data


# ------------------------------------------------------------------------------- encoder kernels
ENC_CASES = {
    "empty": (131072, lambda: b""),
    "hello": (131072, lambda: b"hello world, hello world, hello!"),
    "text_300": (131072, lambda: cases.text(300, 2)),
    "text_70k": (131072, lambda: cases.text(70000, 3)),
    "text_2x128k_p5": (131072, lambda: cases.text(2 * 131072 + 5, 11)),
    "text_300k_chunk1m": (1 << 20, lambda: cases.text(300000, 5)),
    "random_140k": (131072, lambda: cases.rnd(140000, 3)),
    "zeros_300k": (1 << 20, lambda: bytes(300000)),
    "period_300": (1 << 20, lambda: cases.rep(cases.rnd(300, 9), 150000)),
    "mixed": (262144, lambda: cases.text(50000, 4) + bytes(70000) + cases.rnd(3000, 5) + cases.text(100000, 6)),
    # byte values above 128: the Huffman tree needs FSE-coded weights (RFC 8878 4.2.1.2)
    "text_high_bytes": (131072, lambda: bytes(b ^ 0x80 for b in cases.text(150000, 8))),
    "all_256_values": (1 << 20, lambda: bytes((b * 7) & 255 for b in cases.text(90000, 9))),
    "four_high_symbols": (131072, lambda: bytes(200 + (b & 3) for b in cases.rnd(100000, 5))),
    # the densest units the device encoder can write (7-byte matches, one literal between them:
    # ~16 K sequences per 128 KiB) through the decoder's look-ahead scratch
    "dense_sequences": (1 << 20, lambda: H.dense_sequences(150000)),
    # the same with literals that do not compress: raw literal sections, the blocks still form units
    "dense_sequences_raw_literals": (1 << 20, lambda: H.dense_sequences(200000, None)),
}


@pytest.mark.parametrize("name", sorted(ENC_CASES))
def test_emu_encode_decompress_identical(name):
    chunk, thunk = ENC_CASES[name]
    data = thunk()
    st = E.zstd_compress(data, chunk)
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    out, status = E.zstd_decompress(st)          # and through the emulated decoder kernels
    assert (status == 0).all() and out == data
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st, threads=2)
        assert rv == 0 and back == data
    # independent of how many persistent waves share the blocks
    assert E.zstd_compress(data, chunk, grid=1) == st


def test_emu_encoder_fits_sequence_tables_per_unit():
    """A unit (the blocks of one 128 KiB of input) with enough sequences describes its own FSE tables
    in its first block (modes 0xA8: three Compressed) and the others repeat them (0xFC); a small unit
    keeps the predefined tables (0).  The tables keep the predefined sizes (2^6 / 2^5 / 2^6 cells)."""
    data = cases.text(2 * 131072 + 3000, 11)
    st = E.zstd_compress(data, 1 << 20)
    blocks = H.zstd_walk_blocks(st[12:])
    modes = [b["modes"] for b in blocks if b["type"] == 2]
    heads = [i for i, m in enumerate(modes) if m == 0xA8]
    assert len(heads) == 2 and heads[0] == 0                      # two full units
    assert all(m == 0xFC for m in modes[1:heads[1]]) and heads[1] >= 2
    assert modes[-1] == 0 and blocks[-1]["nseq"] < 256            # the 3000-byte tail: predefined
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    # and it pays: the bench-like text shrinks by more than it did with predefined tables (2.29)
    big = cases.text(1 << 20, 5)
    assert len(big) / len(E.zstd_compress(big, 131072, grid=8)) > 2.40


@pytest.mark.parametrize("level", [3, 10])
@pytest.mark.parametrize("name", ["text_2x128k_p5", "text_300k_chunk1m", "mixed", "period_300", "dense_sequences", "empty"])
def test_emu_encode_level_tiers_decompress_identical(name, level):
    """levels 3-9 and 10-22 run the encoder's larger-table tiers (gpumt_zstd_level_tier): the same bar"""
    chunk, thunk = ENC_CASES[name]
    data = thunk()
    st = E.zstd_compress(data, chunk, level=level)
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    out, status = E.zstd_decompress(st)
    assert (status == 0).all() and out == data
    assert E.zstd_compress(data, chunk, grid=1, level=level) == st


def test_emu_many_small_chunks_matches_end_with_their_block():
    """1 KiB chunks (tests/test_gpu_zstd.py::test_more_records_than_one_launch_slice on the CPU): a repeat-offset match
    measured 16 bytes ahead must end with its block, not in the next chunk's bytes (round 5: it did not, the literal
    copy behind the last match ran negative)"""
    data = cases.text(300 * 1024 + 77, 33)
    st = E.zstd_compress(data, 1024, grid=8)
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    out, status = E.zstd_decompress(st)
    assert (status == 0).all() and out == data
    # structured input: the repeat offsets are used and decode (RFC 8878 3.1.1.5 history, unknown at a unit's start)
    import json
    js = ("".join(json.dumps({"id": i, "name": "user%d" % (i * 7919 % 1000), "tags": ["a", "b", "c"][:i % 4],
                             "score": (i * 31) % 100 / 10, "active": i % 3 == 0}) + "\n" for i in range(9000))).encode()
    for chunk in (1 << 20, 200000, 4096):
        st = E.zstd_compress(js, chunk, grid=5)
        assert H.oracle_zstdmt_decompress(st, len(js) + 64) == js
        out, status = E.zstd_decompress(st)
        assert (status == 0).all() and out == js
    assert len(js) / len(E.zstd_compress(js, 1 << 20, grid=5)) > 9.0   # 7.9 without repeat offsets


def test_emu_ratio_is_monotone_in_level():
    """the reference hands `level` to ZSTD_compress (lib/zstd-mt_compress.c:285): a higher level must not
    compress worse, and the tiers must be worth having on text"""
    data = cases.text(1 << 20, 5)
    sizes = [len(E.zstd_compress(data, 1 << 20, grid=8, level=lv)) for lv in (1, 3, 10)]
    assert sizes[0] > sizes[1] > sizes[2]
    assert sizes[0] / sizes[2] > 1.04
    # same tier, same bytes
    assert len(E.zstd_compress(data, 1 << 20, grid=8, level=9)) == sizes[1]
    assert len(E.zstd_compress(data, 1 << 20, grid=8, level=22)) == sizes[2]


@pytest.mark.parametrize("seed", range(4))
def test_emu_fuzz_encode_and_decode(seed):
    """Structured soup through the emulated encoder, the oracle and the emulated decoder; and, where
    the reference build is present, reference-written streams of the same data through the decoder."""
    import random
    rng = random.Random(3000 + seed)
    data = H.soup(rng, rng.choice([7, 300, 70000, 140000, rng.randrange(1, 260000)]))
    chunk = rng.choice([65536, 131072, 1 << 20])
    st = E.zstd_compress(data, chunk, level=rng.choice([1, 1, 5, 19]))
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    out, status = E.zstd_decompress(st)
    assert (status == 0).all() and out == data
    if H.have_zref():
        rv, rst, _, _ = H.zstdmt_compress_via(H.zref(), data, chunk, threads=2, level=rng.choice([1, 3, 9]))
        assert rv == 0
        out, status = E.zstd_decompress(rst)
        assert (status == 0).all() and out == data


def test_emu_checksummed_frames():
    """Frames with an XXH64 content checksum (zstd CLI flavour): verified on the device path."""
    data = cases.text(150000, 31)
    fr = H.libzstd_frame(data, level=3, checksum=1)
    if fr is None:
        pytest.skip("libzstd not present")
    st = H.mt_record(fr) + H.mt_record(H.libzstd_frame(data[:777], 1, 1)) + H.mt_record(H.libzstd_frame(b"", 1, 1))
    out, status = E.zstd_decompress(st)
    assert status.tolist() == [0, 0, 0] and out == data + data[:777]
    bad = bytearray(st)
    bad[12 + len(fr) - 1] ^= 0x10           # last checksum byte of the first frame
    out, status = E.zstd_decompress(bytes(bad))
    assert status.tolist() == [5, 0, 0]     # GPUMT_ST_BAD_CHECKSUM


def test_emu_corrupt_unit_streams():
    """Streams of the device encoder form units (one tree, up to 16 blocks decoded side by side):
    flips anywhere in such a record must leave the verdict -- and the content, when the stream stays
    valid -- equal to the oracle's, whichever block the damage lands in."""
    data = cases.text(140000, 21)
    st = E.zstd_compress(data, 1 << 20)
    assert E.zstd_decompress(st)[0] == data
    rng = np.random.default_rng(11)
    ro, rl = np.array([0], np.uint64), np.array([len(st)], np.uint32)
    for pos in sorted(set(rng.integers(12, len(st), 20).tolist())):   # (120 flips on the GPU: test_gpu_zstd.py)
        bad = bytearray(st)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = H.oracle_zstdmt_decompress(bad, len(data) + 64)
        out, status = E.zstd_decompress(bad, rec=(ro, rl))
        if want is None:
            assert status[0] != 0, f"flip at {pos}: oracle rejects, kernel accepted"
        else:
            assert status[0] == 0 and out == want, f"flip at {pos}"


# ------------------------------------------------------------------------------- sequence pre-pass
def _both_ways(stream, rec=None):
    """decode with the sequence pre-pass (default) and without it; returns (out, status, marks) of the first after
    checking that the second says the same"""
    import os
    out, status = E.zstd_decompress(stream, rec=rec)
    marks = E.zstd_last_marks()
    os.environ["EMU_ZSTD_SEQ"] = "1"
    try:
        out1, status1 = E.zstd_decompress(stream, rec=rec)
        assert E.zstd_last_marks() == 0
    finally:
        del os.environ["EMU_ZSTD_SEQ"]
    assert status.tolist() == status1.tolist() and out == out1
    return out, status, marks


@pytest.mark.skipif(not H.have_zref(), reason="reference build not present")
@pytest.mark.parametrize("level", [1, 3, 19])
def test_emu_seq_prepass_reference_frames(level):
    """frames of several blocks written by the reference (every block with its own fitted tables): the pre-pass
    (zstd_dec_seq.hip) decodes the sequences of all blocks side by side, the frame decoder only executes them"""
    data = cases.text(700000, 41) + bytes(5000) + cases.rnd(2000, 3) + cases.text(300000, 42)
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, 1 << 20, threads=2, level=level)
    assert rv == 0
    blocks = H.zstd_walk_blocks(st[12:])
    assert len([b for b in blocks if b["type"] == 2 and b.get("nseq")]) >= 6
    out, status, marks = _both_ways(st)
    assert (status == 0).all() and out == data
    assert marks >= 6, "the pre-pass did not take the frame"


@pytest.mark.skipif(not H.have_zref(), reason="reference build not present")
def test_emu_seq_prepass_more_blocks_than_one_group():
    """a 2.5 MiB frame = 20 blocks = three groups of eight; repetitive stretches make the writer use repeat /
    predefined / RLE table modes, also across the boundary of a group"""
    parts = [cases.text(300000, 7), cases.rep(cases.rnd(300, 9), 500000), cases.text(200000, 8),
             cases.rep(cases.rnd(5000, 10), 700000), bytes(200000), cases.text(700000, 9)]
    data = b"".join(parts)
    for level in (1, 5):
        rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, 4 << 20, threads=1, level=level)
        assert rv == 0
        modes = [b.get("modes") for b in H.zstd_walk_blocks(st[12:]) if b["type"] == 2 and b.get("nseq")]
        assert len(modes) > 16
        out, status, marks = _both_ways(st)
        assert (status == 0).all() and out == data
        assert marks > 8, (marks, modes)


@pytest.mark.skipif(not H.have_zref(), reason="reference build not present")
def test_emu_seq_prepass_leaves_what_it_cannot_hold():
    """one sequence per 8 bytes of content is what a record's region holds: denser frames are decoded ahead as far as
    they fit (or not at all) and the frame decoder does the rest; frames with more blocks than the region's header
    has words are left alone"""
    dense = H.dense_sequences(600000)
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), dense, 1 << 20, threads=1, level=1)
    assert rv == 0
    out, status, marks = _both_ways(st)
    assert (status == 0).all() and out == dense
    # 70 empty raw blocks spliced in front of the first block: 64 header words do not reach the frame's end
    data = cases.text(400000, 12)
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, 1 << 20, threads=1, level=1)
    fr = st[12:]
    fhd = fr[4]
    hdr_len = 5 + (0 if (fhd >> 5) & 1 else 1) + ((1 if (fhd >> 5) & 1 else 0), 2, 4, 8)[fhd >> 6]
    fr2 = fr[:hdr_len] + bytes(3) * 70 + fr[hdr_len:]
    st2 = H.mt_record(fr2)
    assert H.oracle_zstdmt_decompress(st2, len(data) + 64) == data
    out, status, marks = _both_ways(st2)
    assert (status == 0).all() and out == data and marks == 0


@pytest.mark.skipif(not H.have_zref(), reason="reference build not present")
def test_emu_seq_prepass_corrupt_streams():
    """damage anywhere in a frame the pre-pass takes: verdict, and content where the stream stays valid, are the
    oracle's -- with and without the pre-pass"""
    data = cases.text(500000, 33)
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, 1 << 20, threads=1, level=1)
    assert rv == 0
    rng = np.random.default_rng(5)
    ro, rl = np.array([0], np.uint64), np.array([len(st)], np.uint32)
    for pos in sorted(set(rng.integers(12, len(st), 10).tolist())):   # (more flips: tools/emu_fuzz_zstd_seq.py, test_gpu_zstd.py)
        bad = bytearray(st)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = H.oracle_zstdmt_decompress(bad, len(data) + 64)
        out, status, _ = _both_ways(bad, rec=(ro, rl))
        if want is None:
            assert status[0] != 0, f"flip at {pos}: oracle rejects, kernel accepted"
        else:
            assert status[0] == 0 and out == want, f"flip at {pos}"


@pytest.mark.parametrize("repeat", [0, 2, 1])
def test_emu_seq_prepass_rle_mode_tables(repeat):
    """blocks whose three sequence tables are in RLE mode (and blocks that repeat them): hand-made, libzstd does not write
    them on ordinary data; the oracle (and, where built, the reference library) must read the frame the same way"""
    fr, content = H.zstd_rle_mode_frame(4, repeat)
    st = H.mt_record(fr)
    assert H.oracle_zstdmt_decompress(st, len(content) + 64) == content
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st, threads=1)
        assert rv == 0 and back == content
    out, status, marks = _both_ways(st)
    # (a second block that repeats all three tables looks like a unit of the device encoder: left to the decoder)
    assert (status == 0).all() and out == content and marks == (0 if repeat == 1 else 4)
    # one flipped bit in every block's sequences section: a wrong extra bit changes a length or an offset, the frame then
    # misses its content size or reaches outside it -- verdict (and content, should it survive) as the oracle's
    at = 12 + 9
    ro, rl = np.array([0], np.uint64), np.array([len(st)], np.uint32)
    for blk in H.zstd_walk_blocks(fr):
        if blk["type"] == 2:
            bad = bytearray(st)
            bad[at + 3 + blk["size"] - 3] ^= 0x04
            bad = bytes(bad)
            want = H.oracle_zstdmt_decompress(bad, len(content) + 64)
            out, status, _ = _both_ways(bad, rec=(ro, rl))
            assert (status[0] != 0) if want is None else (status[0] == 0 and out == want)
        at += 3 + (1 if blk["type"] == 1 else blk["size"])


@pytest.mark.parametrize("repeat_from,marks_want", [(0, 2), (2, 0)])
def test_emu_seq_prepass_stops_where_its_region_is_full(repeat_from, marks_want):
    """four-byte sequences: a record's region (one sequence per 8 bytes of content) holds the first two blocks' and not the
    third's.  The decoder goes on by itself from there -- if that block describes its tables itself; if it repeats the
    tables of a block decoded ahead, the pre-pass drops all its marks and the decoder does the whole frame"""
    fr, content = H.zstd_rle_mode_frame(4, repeat_from, seq=(1, 3, 1), nseq=20000)
    st = H.mt_record(fr)
    assert H.oracle_zstdmt_decompress(st, len(content) + 64) == content
    out, status, marks = _both_ways(st)
    assert (status == 0).all() and out == content
    assert marks == marks_want
