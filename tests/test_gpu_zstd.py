"""zstd-mt decode on the device through the C ABI (include/gpumt.h gpumt_zstd_*), against the
streams the reference wrote (tests/golden/zstd) and, where the reference build travelled
(oracle/_ref), against fresh streams at several levels.  Byte-exact decoded content."""
import json
import os

import numpy as np
import pytest

import emu_driver as E
import helpers as H
from golden import cases

pytestmark = pytest.mark.gpu

ZDIR = os.path.join(H.GOLDEN_DIR, "zstd")
MAN = json.load(open(os.path.join(ZDIR, "manifest.json")))["cases"]


@pytest.fixture(scope="module", params=[0, 1], ids=["seq", "noseq"])
def eng(request):
    """every test runs with the sequence pre-pass (zstd_dec_seq.hip) in front of the frame decoder, the default, and
    without it: the two must agree with the oracle on every verdict and every byte"""
    import zstdmt_amd as z
    e = z.Engine(0)
    e.set_variant("zstd_seq", request.param)
    yield e
    e.close()


def _stream(name):
    ent = MAN[name]
    if "out_file" in ent:
        return open(os.path.join(ZDIR, ent["out_file"]), "rb").read()
    if not H.have_zref():
        pytest.skip("stream not committed (size) and no reference build on this box")
    level, chunk, thunk = cases.ZCASES[name]
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), thunk(), chunk, threads=2, level=level)
    assert rv == 0
    return st


@pytest.mark.parametrize("name", sorted(MAN))
def test_decode_golden(eng, name):
    ent = MAN[name]
    st = _stream(name)
    ro, rl = E.walk_records(st)
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert status.tolist() == [0] * ent["frames"]
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]
    assert out == H.oracle_zstdmt_decompress(st, ent["in_len"] + 64)


@pytest.mark.skipif(not H.have_zref(), reason="reference build not on this box")
@pytest.mark.parametrize("level,chunk", [(1, 0), (1, 131072), (3, 0), (5, 65536), (9, 0), (19, 300000)])
def test_decode_live_reference_streams(eng, level, chunk):
    data = cases.text(6 * 1048576 + 12345, 40 + level) + cases.rnd(70000, level) + bytes(200000) + \
        cases.text(300000, 9)
    rv, st, _, stats = H.zstdmt_compress_via(H.zref(), data, chunk, threads=8, level=level)
    assert rv == 0
    ro, rl = E.walk_records(st)
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert (status == 0).all() and len(rl) == stats[0]
    assert out == data


def test_corrupt_streams_status(eng):
    """Flipped bits: per-record status agrees with the oracle's verdict, output stays in bounds
    (Engine.decompress_bytes checks the guard band)."""
    st = _stream("z_text_64k_l1")
    n = MAN["z_text_64k_l1"]["in_len"]
    rng = np.random.default_rng(3)
    ro, rl = np.array([0], np.uint64), np.array([len(st)], np.uint32)
    for pos in sorted(set(rng.integers(12, len(st), 40).tolist())):
        bad = bytearray(st)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = H.oracle_zstdmt_decompress(bad, n + 64)
        out, status = eng.decompress_bytes(bad, ro, rl, codec="zstd")
        if want is None:
            assert status[0] != 0
        else:
            assert status[0] == 0 and out == want


def test_corrupt_unit_streams_status(eng):
    """The same on a record of the device encoder, whose blocks are decoded a unit (one tree, up to
    16 blocks) at a time: damage in any block leaves verdict and content equal to the oracle's."""
    data = cases.text(600000, 21)
    st, ro, rl = eng.compress_bytes(data, 1 << 20, codec="zstd")
    assert len(rl) == 1
    rng = np.random.default_rng(5)
    for pos in sorted(set(rng.integers(12, len(st), 120).tolist())):
        bad = bytearray(st)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = H.oracle_zstdmt_decompress(bad, len(data) + 64)
        out, status = eng.decompress_bytes(bad, ro, rl, codec="zstd")
        if want is None:
            assert status[0] != 0, pos
        else:
            assert status[0] == 0 and out == want, pos


@pytest.mark.parametrize("lits", [b"aaaabbcd", None])
def test_dense_units_and_their_damage(eng, lits):
    """Dense sequence lists with Huffman-coded and with raw literal sections (both form units in
    the decoder): content, and verdict parity with the oracle under bit flips."""
    data = H.dense_sequences(900000, lits)
    st, ro, rl = eng.compress_bytes(data, 1 << 20, codec="zstd")
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert (status == 0).all() and out == data
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    rng = np.random.default_rng(9)
    for pos in sorted(set(rng.integers(12, len(st), 60).tolist())):
        bad = bytearray(st)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = H.oracle_zstdmt_decompress(bad, len(data) + 64)
        out, status = eng.decompress_bytes(bad, ro, rl, codec="zstd")
        if want is None:
            assert status[0] != 0, pos
        else:
            assert status[0] == 0 and out == want, pos


def test_more_records_than_one_launch_slice(eng):
    """The decoder's per-record scratch is sized for 16 384 records; a batch beyond that goes in
    slices (gpumt_zstd_decompress_batch) and every record still lands where it belongs."""
    data = cases.text(20000 * 1024 + 77, 33)
    st, ro, rl = eng.compress_bytes(data, 1024, codec="zstd")
    assert len(rl) == 20001
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert (status == 0).all() and out == data


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_encode_is_decompress_identical(eng, seed):
    import random
    rng = random.Random(1000 + seed)
    n = rng.choice([0, 1, 6, 7, 8, 255, 256, 257, 4095, 4096, 65535, 65536, 131071, 131072, 131073,
                    200000, rng.randrange(1, 600000), rng.randrange(1, 3000000)])
    chunk = rng.choice([65536, 131072, 131072, 100000, 262144, 1 << 20])
    data = H.soup(rng, n)
    st, ro, rl = eng.compress_bytes(data, chunk, codec="zstd", level=rng.choice([1, 1, 4, 12]))
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert (status == 0).all() and out == data
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st, threads=2)
        assert rv == 0 and back == data


def test_repeat_offsets_on_structured_data(eng):
    """RFC 8878 repeat offsets (round 5): records whose matches sit at one distance -- JSON lines -- are coded with
    Offset_Value 1, at every chunk size incl. units that start with an unknown history; the oracle, the device decoder
    and the reference library read them back, and they pay"""
    import json
    js = ("".join(json.dumps({"id": i, "name": "user%d" % (i * 7919 % 1000), "tags": ["a", "b", "c"][:i % 4],
                             "score": (i * 31) % 100 / 10, "active": i % 3 == 0}) + "\n" for i in range(40000))).encode()
    for chunk, level in ((1 << 20, 1), (200000, 1), (4096, 1), (1 << 20, 5)):
        st, ro, rl = eng.compress_bytes(js, chunk, codec="zstd", level=level)
        assert H.oracle_zstdmt_decompress(st, len(js) + 64) == js
        out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
        assert (status == 0).all() and out == js
        if H.have_zref():
            rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st, threads=2)
            assert rv == 0 and back == js
    st, _, _ = eng.compress_bytes(js, 1 << 20, codec="zstd", level=1)
    assert len(js) / len(st) > 9.5   # 9.2 without repeat offsets (emulator, 7 / 7), 10.6 with (6 / 6)


def test_level_tiers_ratio_monotone_and_decompress_identical(eng):
    """level reaches the encoder (the reference hands it to ZSTD_compress, lib/zstd-mt_compress.c:285): three tiers,
    each decompress-identical, ratio monotone in level"""
    data = cases.text(4 << 20, 77)
    sizes = []
    for lv in (1, 3, 10, 22):
        st, ro, rl = eng.compress_bytes(data, 1 << 20, codec="zstd", level=lv)
        assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
        out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
        assert (status == 0).all() and out == data
        sizes.append(len(st))
    assert sizes[0] > sizes[1] > sizes[2] == sizes[3]


@pytest.mark.skipif(not H.have_zref(), reason="reference build not on this box")
@pytest.mark.parametrize("seed", range(8))
def test_fuzz_decode_reference_streams(eng, seed):
    import random
    rng = random.Random(2000 + seed)
    data = H.soup(rng, rng.randrange(1, 2500000))
    level = rng.choice([1, 1, 2, 3, 5, 8, 13, 19])
    chunk = rng.choice([0, 65536, 131072, 300000, 1 << 20])
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, chunk, threads=4, level=level)
    assert rv == 0
    ro, rl = E.walk_records(st)
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert (status == 0).all() and out == data


def test_checksummed_frames(eng):
    """Frames with an XXH64 content checksum (zstd CLI flavour): verified on the device."""
    data = cases.text(3000000, 31)
    parts = [data, data[:777], b"", cases.rnd(50000, 4)]
    frames = [H.libzstd_frame(p, lvl, 1) for p, lvl in zip(parts, (3, 1, 1, 5))]
    if frames[0] is None:
        pytest.skip("libzstd not present")
    st = b"".join(H.mt_record(f) for f in frames)
    ro, rl = E.walk_records(st)
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert status.tolist() == [0, 0, 0, 0] and out == b"".join(parts)
    bad = bytearray(st)
    bad[12 + len(frames[0]) - 2] ^= 0x01
    out, status = eng.decompress_bytes(bytes(bad), ro, rl, codec="zstd")
    assert status.tolist() == [5, 0, 0, 0]


def test_chunk_beyond_128_mib_has_a_window_descriptor(eng):
    """A single-segment frame of more than 128 MiB is refused by libzstd's streaming decoder (window
    above 2^27), which is what the reference's pt_decompress runs: such chunks get a 128 KiB
    Window_Descriptor instead, and the reference library must decode them."""
    import ctypes as C
    T = C.CDLL(os.path.join(H.ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
    T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
    n = (129 << 20) + 4321
    buf = np.empty(n, np.uint8)
    T.zmt_gen_text(buf.ctypes.data, n, 20260926, 0, 8)
    data = buf.tobytes()
    st, ro, rl = eng.compress_bytes(data, n, codec="zstd")
    assert len(rl) == 1 and st[12:16] == b"\x28\xb5\x2f\xfd" and st[16] == 0x80 and st[17] == 0x38
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert (status == 0).all() and out == data
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st)
        assert rv == 0 and back == data


@pytest.mark.skipif(not H.have_zref(), reason="reference build not on this box")
def test_seq_prepass_frames_of_many_blocks_and_their_damage(eng):
    """4 MiB chunks = 32 blocks a frame = four groups of the pre-pass; repetitive stretches make the writer use repeat /
    predefined / RLE table modes, also across a group's boundary; then 60 damaged copies of one record, each judged
    like the oracle judges it"""
    parts = [cases.text(3000000, 7), cases.rep(cases.rnd(300, 9), 1500000), cases.text(200000, 8),
             cases.rep(cases.rnd(5000, 10), 2700000), bytes(900000), cases.text(2500000, 9), H.dense_sequences(700000)]
    data = b"".join(parts)
    for level in (1, 5):
        rv, st, _, _ = H.zstdmt_compress_via(H.zref(), data, 4 << 20, threads=4, level=level)
        assert rv == 0
        ro, rl = E.walk_records(st)
        out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
        assert (status == 0).all() and out == data
    rec = st[int(ro[1]):int(ro[1]) + int(rl[1])]
    r1 = (np.array([0], np.uint64), np.array([len(rec)], np.uint32))
    rng = np.random.default_rng(77)
    for pos in sorted(set(rng.integers(12, len(rec), 60).tolist())):
        bad = bytearray(rec)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        want = H.oracle_zstdmt_decompress(bad, (4 << 20) + 64)
        out, status = eng.decompress_bytes(bad, r1[0], r1[1], codec="zstd")
        if status[0] == 7 and want is not None:
            continue                # the damage took the content size away: the probe leaves the frame to the caller
        if want is None:
            assert status[0] != 0, f"flip at {pos}: oracle rejects, device accepted"
        else:
            assert status[0] == 0 and out == want, f"flip at {pos}"


@pytest.mark.parametrize("repeat_from", [0, 2])
def test_rle_mode_sequence_tables(eng, repeat_from):
    """hand-made blocks whose three sequence tables are in RLE mode, and blocks that repeat them (H.zstd_rle_mode_frame):
    libzstd does not write these on ordinary data, the format allows them"""
    fr, content = H.zstd_rle_mode_frame(6, repeat_from)
    st = H.mt_record(fr) + H.mt_record(fr)
    assert H.oracle_zstdmt_decompress(st, 2 * len(content) + 64) == content + content
    ro, rl = E.walk_records(st)
    out, status = eng.decompress_bytes(st, ro, rl, codec="zstd")
    assert status.tolist() == [0, 0] and out == content + content
