"""Kernel logic vs the oracle, on the CPU fiber harness (tests/emu): the same HIP sources that
ship, compiled as host C++.  Small sizes only -- the real parity run is tests/test_gpu_*.py."""
import json
import os

import numpy as np
import pytest

import emu_driver as E
import helpers as H
from cases import CASES, rnd, text

with open(os.path.join(H.GOLDEN_DIR, "manifest.json")) as _f:
    MAN = json.load(_f)["cases"]

SMALL = [n for n in sorted(CASES) if MAN[n]["in_len"] <= 140000]


def test_emu_xxh32():
    import xxhash
    lens = [0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 1000, 4096, 65537]
    blobs = [H.lcg(n, n + 1) for n in lens]
    buf = np.frombuffer(b"".join(blobs) + b"\0" * 16, np.uint8)
    off = np.cumsum([0] + lens[:-1]).astype(np.uint64)
    got = E.xxh32_batch(buf, off, np.array(lens, np.uint32))
    want = [xxhash.xxh32(b, seed=0).intdigest() for b in blobs]
    assert got.tolist() == want


@pytest.mark.parametrize("name", SMALL)
def test_emu_compress_bit_exact(name):
    chunk, thunk = CASES[name]
    data = thunk()
    stream, rec_off, rec_len = E.compress(data, chunk)
    assert len(stream) == MAN[name]["out_len"]
    assert H.sha256(stream) == MAN[name]["out_sha256"]


def _hc_inputs():
    t = text(9000)
    return {
        "empty": (65536, b""),
        "abc_12": (65536, b"abcabcabcabc"),
        "abc_13": (65536, b"abcabcabcabca"),
        "text_9000": (65536, t),
        "zeros_70000": (131072, bytes(70000)),                  # linked second block, long match
        "period_7": (65536, (b"abcdefg" * 1500)[:10000]),       # overlapping matches
        "rnd_3000": (65536, rnd(3000, 5)),                      # does not shrink: stored block
        "text_rnd_text": (65536, t[:3000] + rnd(1500, 9) + t[:3000]),
        "text_4x2500": (2500, t + t[:1000]),                    # ragged chunks, persistent grid of 2
        "lazy_shapes": (65536, b"".join(t[i * 37:i * 37 + 60 + i % 40] + t[:i % 23] for i in range(120))),
        # runs of one byte value of many lengths between text: the pattern analysis of level 9
        "runs": (65536, b"".join(t[i * 11:i * 11 + 9] + bytes([65 + i % 3]) * (3 + (i * 7) % 90) for i in range(90))
                 + bytes(3000) + t[:50] + bytes(700) + b"x" + bytes(900)),
    }


def _hc_cases():
    """every input at the two ends of the hash-chain levels, the levels in between on three of them"""
    names = sorted(_hc_inputs())
    return [(n, lv) for n in names for lv in (3, 9)] + [(n, lv) for n in ("text_9000", "runs", "text_4x2500") for lv in (5, 8)]


@pytest.mark.parametrize("name,level", _hc_cases())
def test_emu_hc_compress_bit_exact(name, level):
    """lz4_enc_hc.hip on the emulator against the LZ4HC oracle (levels 3..9 = hash chain)."""
    chunk, data = _hc_inputs()[name]
    stream, rec_off, rec_len = E.compress(data, chunk, level)
    assert stream == H.oracle_compress_level(data, chunk, level)


def _opt_inputs():
    t = text(9000)
    return {
        "abc_13": (65536, b"abcabcabcabca"),
        "text_3000": (65536, t[:3000]),
        "zeros_5000": (65536, bytes(5000)),
        "period_7": (65536, (b"abcdefg" * 700)[:4000]),
        "runs": (65536, b"".join(t[i * 11:i * 11 + 9] + bytes([65 + i % 3]) * (3 + (i * 7) % 90) for i in range(40))
                 + bytes(1500) + t[:50] + bytes(400)),
        "rnd_1500": (65536, rnd(1500, 5)),
        "two_chunks": (2500, t[:4200]),
    }


@pytest.mark.parametrize("level", [10, 11, 12])
@pytest.mark.parametrize("name", sorted(_opt_inputs()))
def test_emu_hc_optimal_bit_exact(name, level):
    """levels 10..12 (LZ4HC optimal parser) of lz4_enc_hc.hip on the emulator against the oracle"""
    chunk, data = _opt_inputs()[name]
    stream, rec_off, rec_len = E.compress(data, chunk, level)
    assert stream == H.oracle_compress_level(data, chunk, level)


def test_emu_hc_optimal_1mib_chunk():
    """level 10 at a 1 MiB chunk on the emulator: 272 KiB of input = five linked blocks with one table set carried across
    them (chunks above 256 KiB were untested at levels >= 10).  About a minute of one core; the whole MiB, levels 10..12,
    runs on the device (tests/test_gpu_lz4.py::test_hc_optimal_levels_at_a_1mib_chunk)"""
    data = text(272 << 10, seed=77)
    stream, rec_off, rec_len = E.compress(data, 1 << 20, 10)
    assert stream == H.oracle_compress_level(data, 1 << 20, 10)


# decoder variants of the emulator API: 0 | ring << 4 = frames + parse4 + copy3 with a 4 / 8 / 16 KiB ring
# (plain 0 = the product's default, 4 KiB), 1 = frame-serial
DEC_VARIANTS = [0, 1, 0 | 13 << 4, 0 | 14 << 4]


@pytest.mark.parametrize("variant", DEC_VARIANTS)
@pytest.mark.parametrize("name", SMALL)
def test_emu_decompress(name, variant):
    chunk, thunk = CASES[name]
    data = thunk()
    stream = H.oracle_compress(data, chunk)
    out, status = E.decompress(stream, variant)
    assert status.tolist() == [0] * len(status)
    assert out == data


@pytest.mark.parametrize("variant", DEC_VARIANTS)
@pytest.mark.parametrize("mutate,code", [("magic", 2), ("hc", 2), ("blocksize", 3), ("checksum", 5),
                                         ("skipmagic", 1), ("skiplen", 1), ("offset0", 3)])
def test_emu_corrupt_streams(mutate, code, variant):
    """Same corruptions as tests/test_gpu_lz4.py, on the fiber harness (scratch starts as garbage)."""
    data = text(131072) + text(70000, seed=5)
    s = bytearray(H.oracle_compress(data, 131072))
    if mutate == "magic":
        s[12] ^= 1
    elif mutate == "hc":
        s[12 + 14] ^= 0x10
    elif mutate == "blocksize":
        s[12 + 15 + 2] ^= 0x40
    elif mutate == "checksum":
        import struct
        c0 = struct.unpack_from("<I", s, 8)[0]
        s[12 + c0 - 1] ^= 0x80
    elif mutate == "skipmagic":
        s[0] ^= 1
    elif mutate == "skiplen":
        s[4] = 8
    elif mutate == "offset0":
        p = 12 + 15 + 4
        lit = s[p] >> 4
        q = p + 1
        if lit == 15:
            while s[q] == 255:
                lit += 255
                q += 1
            lit += s[q]
            q += 1
        s[q + lit] = 0
        s[q + lit + 1] = 0
    good = H.oracle_compress(data, 131072)
    rec = E.walk_records(good)   # record boundaries of the uncorrupted stream
    out, status = E.decompress(bytes(s), variant, rec=rec)
    assert status.tolist() == [code, 0]
    assert out[131072:] == data[131072:]   # the healthy record still decodes


@pytest.mark.parametrize("variant", DEC_VARIANTS)
def test_emu_block_checksum_and_dictid_frames(variant):
    """frames with block checksums / a dictionary id (liblz4-written fixtures): the pipeline hands them
    to the wave-per-record decoder, which verifies every block's XXH32"""
    d = os.path.join(H.GOLDEN_DIR, "lz4f_flags")
    man = json.load(open(os.path.join(d, "manifest.json")))["cases"]
    for name, e in man.items():
        rec = open(os.path.join(d, name + ".rec"), "rb").read()
        out, status = E.decompress(rec, variant)
        assert status.tolist() == [0], (name, status)
        assert H.sha256(out) == e["content_sha256"], name
        if e["flg"] & 0x10:
            bad = bytearray(rec)
            bad[12 + 40] ^= 0x01
            _, status = E.decompress(bytes(bad), variant, rec=E.walk_records(rec))
            assert status.tolist() == [5], (name, status)   # GPUMT_ST_BAD_CHECKSUM


def test_emu_hc_stored_block_after_an_attempt_that_wrote_matches():
    """A last chunk of 45 bytes whose HC attempt emits a sequence and then fails to shrink: the block is
    stored raw over the bytes the attempt wrote.  (tools/emu_fuzz_hc.py seed 166: without a wave sync in
    front of the raw copy the emulator let late stores of the attempt land on top of it.)"""
    import random
    rng = random.Random(166 * 4099 + 5)
    level = rng.choice([3, 4, 5, 6, 7, 8, 9, 9, 10, 11, 12])
    n = rng.choice([rng.randrange(1, 40000), rng.randrange(1, 3000), 65536 + rng.randrange(0, 50), rng.randrange(60000, 140000)])
    assert rng.randrange(5) == 3 and (level, n) == (7, 65581)
    t = bytearray(text(n, seed=rng.randrange(1 << 30)))
    for _ in range(rng.randrange(1, 15)):
        a = rng.randrange(0, n - 50)
        ln = rng.randrange(1, min(3000, n - a))
        b = rng.randrange(0, n - ln)
        t[b:b + ln] = t[a:a + ln] if rng.random() < 0.7 else bytes([rng.randrange(256)]) * ln
    data = bytes(t)
    tail = data[65536:]
    assert len(tail) == 45
    for lv in (3, 7, 9, 10, 12):
        got = E.compress(tail, 65536, lv)[0]
        assert got == H.oracle_compress_level(tail, 65536, lv), lv
        assert got[12 + 15:12 + 19] == (45 | 0x80000000).to_bytes(4, "little")     # stored block
    assert E.compress(data, 65536, 7)[0] == H.oracle_compress_level(data, 65536, 7)


def test_emu_fast_encoder_stored_block_after_an_attempt_that_wrote_matches():
    """The same for the level-1 encoder (tools/emu_fuzz_small.py seeds 1 and 280): 18 bytes with a 4-byte repeat
    -- the attempt emits a sequence, does not shrink, and the block is stored."""
    for data in (b"Tie sie sspfr mdot", b"abcdabcd-abcdabcd", text(4096, 1) + b"Tie sie sspfr mdot, sie sspfr"):
        for chunk in (65536, 4096):
            assert E.compress(data, chunk, 1)[0] == H.oracle_compress(data, chunk), (data[-20:], chunk)


def _block_kinds(stream):
    """(stored?, size) of every block of every record of an lz4-mt stream"""
    import struct
    out, i = [], 0
    while i < len(stream):
        c = struct.unpack_from("<I", stream, i + 8)[0]
        p = i + 12 + 15
        while True:
            bh = struct.unpack_from("<I", stream, p)[0]
            p += 4
            if bh == 0:
                break
            out.append((bh >> 31, bh & 0x7FFFFFFF))
            p += bh & 0x7FFFFFFF
        i += 12 + c
    return out


def test_emu_fast_encoder_blocks_near_the_output_limit():
    """Random blocks with a stretch of compressible bytes sized so that the savings land within a few bytes of what the
    literal runs' length bytes cost: blocks that barely fit and blocks that barely do not (LZ4F stores those raw, SURVEY
    Appendix B).  The encoder advances and tests its output position sequence by sequence as the reference does -- one
    compare covers both of the reference's tests in the common case, the two tests decide near the limit (round 5) -- so
    every stream must equal the oracle's, and both outcomes must occur."""
    import random
    stored = fit = 0
    for seed in range(1000, 1030):
        rng = random.Random(777000 + seed)
        n = rng.choice([65536, 131072, 65536 + rng.randrange(20, 60000)])
        base = bytearray(rng.getrandbits(8) for _ in range(n))
        words = [bytes(rng.getrandbits(8) for _ in range(rng.randrange(3, 9))) for _ in range(40)]
        for blk in range(0, n, 65536):
            blen = min(65536, n - blk)
            if blen < 400:
                continue
            target = blen // 255 + rng.randrange(-6, 30)
            span = int(target * rng.uniform(2.0, 7.0)) + rng.randrange(0, 40)
            at = blk + rng.randrange(0, max(1, blen - span - 8))
            t = bytearray()
            while len(t) < span:
                t += rng.choice(words)
            base[at:at + span] = t[:span]
        data, chunk = bytes(base), rng.choice([65536, 131072, 1 << 20])
        want = H.oracle_compress(data, chunk)
        assert E.compress(data, chunk)[0] == want, seed
        for st, _ in _block_kinds(want):
            stored += st
            fit += 1 - st
    assert stored >= 10 and fit >= 10, (stored, fit)


from cases import enc3_path_inputs as _enc3_path_inputs


@pytest.mark.parametrize("chunk", [65536, 131072, 262144, 100000])
@pytest.mark.parametrize("name", sorted(_enc3_path_inputs()))
def test_emu_enc3_paths_bit_exact(name, chunk):
    data = _enc3_path_inputs()[name]
    got, _, _ = E.compress(data, chunk)
    assert got == H.oracle_compress(data, chunk)
    out, status = E.decompress(got)
    assert status.tolist() == [0] * len(status) and out == data


@pytest.mark.parametrize("chunk", [65536, 131072, 262144])
def test_emu_probe_batch_encoder_variant_is_still_bit_exact(chunk, monkeypatch):
    """lz4_enc3.hip (round 5's probe batches) stays in the tree as a variant of the window encoder lz4_enc5.hip
    (GPUMT_LZ4_ENC=3 / ZMT_EMU_LZ4_ENC=3): both must write the oracle's bytes on the path inputs and on windows full of
    twins (few distinct 5-byte strings: every window holds positions with equal hashes, chains of three and more)"""
    import random
    rng = random.Random(4242)
    words = [bytes(rng.randrange(97, 101) for _ in range(rng.randrange(2, 7))) for _ in range(12)]
    twins = b"".join(rng.choice(words) for _ in range(30000))
    inputs = dict(_enc3_path_inputs(), twins=twins, zeros_and_text=bytes(5000) + twins[:20000] + bytes(70000))
    for name, data in sorted(inputs.items()):
        want = H.oracle_compress(data, chunk)
        monkeypatch.setenv("ZMT_EMU_LZ4_ENC", "3")
        assert E.compress(data, chunk)[0] == want, ("enc3", name)
        monkeypatch.setenv("ZMT_EMU_LZ4_ENC", "5")
        assert E.compress(data, chunk)[0] == want, ("enc5", name)

