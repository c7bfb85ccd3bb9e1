"""Parity tests proper: the HIP path, through the C ABI (include/gpumt.h), against the oracle and
the committed golden vectors.  Bit-exact for frames, byte-exact for decoded content."""
import json
import os

import numpy as np
import pytest

import helpers as H
from cases import CASES, KNOWN_HEX, rnd, text

pytestmark = pytest.mark.gpu

# decoder kernel variants under test (include/gpumt.h gpumt_set_variant "lz4_dec")
# 0 | ring << 4 = frames + parse4 + copy3 pipeline with a 4 / 8 / 16 KiB LDS ring per wave ("lz4_ring" = 12 / 13 / 14;
# the default is 12), 1 = serial wave-per-record decoder
VARIANTS = [0 | 12 << 4, 1, 0 | 13 << 4, 0 | 14 << 4]


def set_dec(eng, v):
    eng.set_variant("lz4_dec", v & 15)
    eng.set_variant("lz4_ring", (v >> 4) or 12)

with open(os.path.join(H.GOLDEN_DIR, "manifest.json")) as _f:
    MAN = json.load(_f)["cases"]


@pytest.fixture(scope="module")
def eng():
    import zstdmt_amd as z
    e = z.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_compress_golden(eng, name):
    chunk, thunk = CASES[name]
    data = thunk()
    stream, rec_off, rec_len = eng.compress_bytes(data, chunk)
    e = MAN[name]
    assert len(stream) == e["out_len"]
    assert H.sha256(stream) == e["out_sha256"]
    assert len(rec_len) == e["frames"]
    if name in KNOWN_HEX:
        assert stream.hex() == KNOWN_HEX[name]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", sorted(CASES))
def test_decompress_golden(eng, name, variant):
    chunk, thunk = CASES[name]
    data = thunk()
    stream = H.oracle_compress(data, chunk)
    assert H.sha256(stream) == MAN[name]["out_sha256"]
    import emu_driver as E
    ro, rl = E.walk_records(stream)
    set_dec(eng, variant)
    try:
        out, status = eng.decompress_bytes(stream, ro, rl)
    finally:
        set_dec(eng, 0)
    assert status.tolist() == [0] * len(status)
    assert out == data


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_vs_oracle(eng, seed):
    import random
    from test_oracle_vs_ref import _mix
    rng = random.Random(1000 + seed)
    n = rng.randrange(1, 3_000_000)
    chunk = rng.choice([65536, 131072, 131072, 100000, 262144, 1 << 20])
    data = _mix(rng, n)
    want = H.oracle_compress(data, chunk)
    stream, ro, rl = eng.compress_bytes(data, chunk)
    assert stream == want
    for variant in VARIANTS:
        set_dec(eng, variant)
        out, status = eng.decompress_bytes(stream, ro, rl)
        set_dec(eng, 0)
        assert not status.any() and out == data


@pytest.mark.parametrize("chunk", [65536, 131072, 262144])
def test_fast_encoder_paths_vs_oracle(eng, chunk):
    """the inputs aimed at the paths of lz4_enc3.hip (re-match probe in lane 0, quick extension and its fall-backs,
    ring restarts; tests/golden/cases.py::enc3_path_inputs), each repeated over several chunks"""
    from cases import enc3_path_inputs
    for name, d in sorted(enc3_path_inputs().items()):
        data = d * 3 + d[: len(d) // 3]
        want = H.oracle_compress(data, chunk)
        stream, ro, rl = eng.compress_bytes(data, chunk)
        assert stream == want, name
        out, status = eng.decompress_bytes(stream, ro, rl)
        assert not status.any() and out == data, name


def test_probe_batch_encoder_variant_equals_window_encoder(eng):
    """gpumt_set_variant("lz4_enc", 3) selects round 5's probe-batch encoder (lz4_enc3.hip); the default is the window
    encoder (lz4_enc5.hip).  Both are the reference's parse: same bytes on text, on windows full of equal hashes and on
    incompressible input, at the three table modes (chunks of 64 KiB, 128 KiB, 1 MiB)"""
    import random
    rng = random.Random(4242)
    words = [bytes(rng.randrange(97, 101) for _ in range(rng.randrange(2, 7))) for _ in range(12)]
    twins = b"".join(rng.choice(words) for _ in range(60000))
    data = text(700000, 11) + twins + rnd(200000, 5) + bytes(100000) + text(300000, 12)
    try:
        for chunk in (65536, 131072, 1 << 20):
            want = H.oracle_compress(data, chunk)
            for v in (3, 0):
                eng.set_variant("lz4_enc", v)
                stream, _, _ = eng.compress_bytes(data, chunk)
                assert stream == want, (v, chunk)
    finally:
        eng.set_variant("lz4_enc", 0)


with open(os.path.join(H.GOLDEN_DIR, "lz4hc", "manifest.json")) as _f:
    HCMAN = json.load(_f)


def test_fast_encoder_blocks_near_the_output_limit(eng):
    """tests/test_emu_kernels.py::test_emu_fast_encoder_blocks_near_the_output_limit on the device: random blocks with a
    compressible stretch sized so that they barely fit or barely do not (LZ4F stores those raw); the encoder's collected
    emit advances and tests the output position sequence by sequence as the reference does, so every stream equals the
    oracle's"""
    import random
    import struct
    stored = fit = 0
    for seed in range(1000, 1040):
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
        stream, _, _ = eng.compress_bytes(data, chunk)
        assert stream == H.oracle_compress(data, chunk), seed
        # (stored?, size) of every block of the DEVICE's stream: both outcomes must occur
        i = 0
        while i < len(stream):
            c = struct.unpack_from("<I", stream, i + 8)[0]
            p = i + 12 + 15
            while True:
                bh = struct.unpack_from("<I", stream, p)[0]
                p += 4
                if bh == 0:
                    break
                stored += bh >> 31
                fit += 1 - (bh >> 31)
                p += bh & 0x7FFFFFFF
            i += 12 + c
    assert stored >= 10 and fit >= 10, (stored, fit)


@pytest.mark.parametrize("level", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_hc_compress_golden(eng, level):
    """LZ4HC levels 3..12 against the digests the reference build wrote (tests/golden/gen_golden_lz4hc.py)."""
    n = 0
    for name, e in HCMAN["cases"].items():
        chunk, thunk = CASES[name]
        want = e["levels"].get(str(level))
        if want is None or (level >= 10 and chunk > 131072):   # a 1 MiB chunk is one wave for seconds up there
            continue
        stream, ro, rl = eng.compress_bytes(thunk(), chunk, level=level)
        assert (len(stream), H.sha256(stream)) == (want["out_len"], want["out_sha256"]), name
        n += 1
    assert n >= 20


@pytest.mark.parametrize("level", [10, 11, 12])
def test_hc_optimal_levels_at_a_1mib_chunk(eng, level):
    """levels 10..12 (the optimal parser) with ONE wave walking a 1 MiB chunk of 16 linked blocks: slow on the device
    (seconds per chunk -- a wave per chunk and a 4 096-position price table), bit-exact all the same"""
    data = text(1 << 20, seed=77) + text(300000, seed=78)
    stream, ro, rl = eng.compress_bytes(data, 1 << 20, level=level)
    assert stream == H.oracle_compress_level(data, 1 << 20, level)
    out, status = eng.decompress_bytes(stream, ro, rl)
    assert not status.any() and out == data


@pytest.mark.parametrize("seed", range(6))
def test_hc_fuzz_vs_oracle(eng, seed):
    import random
    from test_oracle_vs_ref import _mix
    rng = random.Random(7000 + seed)
    n = rng.randrange(1, 2_000_000)
    chunk = rng.choice([65536, 131072, 100000, 262144, 1 << 20])
    level = rng.choice([3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
    chunk = min(chunk, 262144) if level >= 10 else chunk
    data = _mix(rng, n)
    stream, ro, rl = eng.compress_bytes(data, chunk, level=level)
    assert stream == H.oracle_compress_level(data, chunk, level)
    out, status = eng.decompress_bytes(stream, ro, rl)
    assert not status.any() and out == data


@pytest.mark.parametrize("seed", range(4))
def test_hc_level9_and_up_runs_vs_oracle(eng, seed):
    """byte runs of every length: the repeated-pattern shortcut of level 9 and of the optimal parser"""
    import random
    from test_oracle_vs_ref import _runs
    rng = random.Random(9100 + seed)
    data = _runs(rng, rng.randrange(200_000, 1_500_000))
    chunk = rng.choice([65536, 131072, 262144])
    level = [9, 10, 11, 12][seed]
    stream, ro, rl = eng.compress_bytes(data, chunk, level=level)
    assert stream == H.oracle_compress_level(data, chunk, level)
    out, status = eng.decompress_bytes(stream, ro, rl)
    assert not status.any() and out == data


def test_hc_many_chunks_persistent_grid(eng):
    """More chunks than GPUMT_LZ4HC_WAVES: every wave resets its tables and takes several chunks."""
    data = text(40 << 20)
    chunk = 16384
    stream, ro, rl = eng.compress_bytes(data, chunk, level=3)
    assert len(rl) == 2560
    for i in (0, 1, 2047, 2048, 2559):
        rec = stream[int(ro[i]):int(ro[i]) + int(rl[i])]
        assert rec == H.oracle_compress_level(data[i * chunk:(i + 1) * chunk], chunk, 3)
    out, status = eng.decompress_bytes(stream, ro, rl)
    assert not status.any() and out == data
    assert eng.L.gpumt_lz4_level_supported(12) == 1 and eng.L.gpumt_lz4_level_supported(13) == 0


def test_config1_random_64m(eng):
    """BASELINE config 1 shape: 64 MiB of PRNG bytes, default 4 MiB chunks -> 16 frames of 64 stored
    blocks; output = input + 16*(12+15+64*4+4+4) bytes."""
    data = rnd(64 << 20, 7)
    stream, ro, rl = eng.compress_bytes(data, 4 << 20)
    assert len(rl) == 16 and len(stream) == (64 << 20) + 4656
    assert stream == H.oracle_compress(data, 4 << 20)
    out, status = eng.decompress_bytes(stream, ro, rl)
    assert not status.any() and out == data


def test_text_256m_roundtrip_and_checksum_of_checksums(eng):
    """Larger-than-oracle-friendly size: size-independent properties (round trip; per-chunk XXH32
    of the decoded data equals the checksum each frame carries) + oracle spot-check of 4 chunks."""
    n = 256 << 20
    data = text(n)
    stream, ro, rl = eng.compress_bytes(data, 131072)
    out, status = eng.decompress_bytes(stream, ro, rl)
    assert not status.any()
    assert out == data
    for i in (0, 1, 1000, len(rl) - 1):
        rec = stream[int(ro[i]):int(ro[i]) + int(rl[i])]
        assert rec == H.oracle_compress(data[i * 131072:(i + 1) * 131072], 131072)


def test_text_1g_every_record_vs_oracle(eng):
    """configs[1] at 1 GiB of the bench text (the C generator bench.py uses), 128 KiB chunks: EVERY record the device
    writes is compared with the oracle's -- the oracle compresses 64 MiB slices on the host's threads (the C restatement
    releases the GIL), records are independent, so the concatenation is the whole stream -- and the per-record XXH32 of
    both streams must agree record by record; then the device decodes its own stream back."""
    import ctypes as C
    import os
    from concurrent.futures import ThreadPoolExecutor
    import xxhash
    T = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zstdmt_amd", "lib", "libzmt_tools.so"))
    T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
    n, chunk, piece = 1 << 30, 131072, 64 << 20
    buf = np.empty(n, np.uint8)
    T.zmt_gen_text(buf.ctypes.data, n, 20260926, 0, min(32, os.cpu_count() or 1))
    data = buf.tobytes()
    stream, ro, rl = eng.compress_bytes(data, chunk)
    assert len(rl) == n // chunk
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        want = list(ex.map(lambda a: H.oracle_compress(data[a:a + piece], chunk), range(0, n, piece)))
    per = piece // chunk
    bad = []
    for k, w in enumerate(want):
        lo, hi = int(ro[k * per]), int(ro[(k + 1) * per])
        got = stream[lo:hi]
        if got != w:
            # name the first differing record of the slice (XXH32 per record, as the streams carry no index)
            off = 0
            for i in range(per):
                ln = int(rl[k * per + i])
                if xxhash.xxh32(got[off:off + ln]).intdigest() != xxhash.xxh32(w[off:off + ln]).intdigest():
                    bad.append(k * per + i)
                    break
                off += ln
    assert not bad, "records that differ from the oracle: %r" % bad[:8]
    assert sum(len(w) for w in want) == len(stream)
    out, status = eng.decompress_bytes(stream, ro, rl)
    assert not status.any() and out == data


@pytest.mark.parametrize("mutate,code", [("magic", 2), ("hc", 2), ("blocksize", 3), ("checksum", 5),
                                         ("skipmagic", 1), ("skiplen", 1), ("offset0", 3),
                                         ("csize", 4)])
def test_corrupt_streams_rejected(eng, mutate, code):
    data = text(131072)
    s = bytearray(H.oracle_compress(data, 131072))
    if mutate == "magic":
        s[12] ^= 1
    elif mutate == "hc":
        s[12 + 14] ^= 0x10
    elif mutate == "blocksize":
        s[12 + 15 + 2] ^= 0x40
    elif mutate == "checksum":
        s[-1] ^= 0x80
    elif mutate == "skipmagic":
        s[0] ^= 1
    elif mutate == "skiplen":
        s[4] = 8
    elif mutate == "csize":
        pass
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
    ro = np.array([0], np.uint64)
    rl = np.array([len(s)], np.uint32)
    if mutate == "csize":
        # claim one byte less than the frame really holds, with a valid header checksum
        import xxhash
        s[12 + 6] = (s[12 + 6] - 1) & 0xFF
        s[12 + 14] = (xxhash.xxh32(bytes(s[12 + 4:12 + 14]), seed=0).intdigest() >> 8) & 0xFF
    for variant in VARIANTS:
        set_dec(eng, variant)
        out, status = eng.decompress_bytes(bytes(s), ro, rl)
        set_dec(eng, 0)
        assert status.tolist() == [code], (variant, status)
    assert H.oracle_decompress(bytes(s), 131072) is None   # the oracle rejects it too


def test_xxh32_batch(eng):
    import xxhash
    import ctypes as C
    lens = [0, 1, 15, 16, 17, 100, 4096, 65537, 131072]
    blobs = [H.lcg(n, n + 1) for n in lens]
    buf = b"".join(blobs)
    off = np.cumsum([0] + lens[:-1]).astype(np.uint64)
    d_buf = eng.upload(buf)
    d_off = eng.upload(off)
    d_len = eng.upload(np.array(lens, np.uint32))
    d_h = eng.alloc(len(lens) * 4)
    eng._ck(eng.L.gpumt_xxh32_batch(eng.h, d_buf.ptr, d_off.ptr, d_len.ptr, len(lens), d_h.ptr, 0), "xxh")
    got = eng.download(d_h, len(lens) * 4, np.uint32).tolist()
    assert got == [xxhash.xxh32(b, seed=0).intdigest() for b in blobs]


@pytest.mark.parametrize("variant", VARIANTS)
def test_block_checksum_and_dictid_frames(eng, variant):
    """LZ4F features lz4-mt never writes but liblz4 decodes (fixtures written by liblz4 1.9.3,
    tests/golden/gen_golden_lz4f_flags.py): block checksums are verified on the device, a dictionary
    id is skipped; a damaged block reports GPUMT_ST_BAD_CHECKSUM."""
    import emu_driver as E
    d = os.path.join(H.GOLDEN_DIR, "lz4f_flags")
    man = json.load(open(os.path.join(d, "manifest.json")))["cases"]
    set_dec(eng, variant)
    try:
        for name, e in man.items():
            rec = open(os.path.join(d, name + ".rec"), "rb").read()
            ro, rl = E.walk_records(rec)
            out, status = eng.decompress_bytes(rec, ro, rl)
            assert status.tolist() == [0], (name, status)
            assert H.sha256(out) == e["content_sha256"], name
            if e["flg"] & 0x10:
                bad = bytearray(rec)
                bad[12 + 40] ^= 0x01
                _, status = eng.decompress_bytes(bytes(bad), ro, rl)
                assert status.tolist() == [5], (name, status)
    finally:
        set_dec(eng, 0)
