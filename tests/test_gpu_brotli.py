"""brotli-mt decode on the device through the C ABI (include/gpumt.h gpumt_brotli_decompress_batch),
against the streams the reference wrote (tests/golden/brotli), fresh streams from the reference
build / the image's libbrotli where they are on this box, and the oracle's verdict on corrupted
streams.  Byte-exact decoded content."""
import json
import os
import random

import numpy as np
import pytest

import emu_driver as E
import helpers as H
from golden import cases

pytestmark = pytest.mark.gpu

BDIR = os.path.join(H.GOLDEN_DIR, "brotli")
MAN = json.load(open(os.path.join(BDIR, "manifest.json")))["cases"]
needs_lib = pytest.mark.skipif(not H.have_libbrotli(), reason="libbrotli 1.0.9 not on this box")
needs_ref = pytest.mark.skipif(not H.have_bref(), reason="reference build not on this box")


# decoder variants (gpumt_set_variant("brotli_dec", v)): 2 = zmt_brotli_dec4_kernel (four records per wave) first and
# the general kernel for what it hands over -- what a large batch takes by itself; 1 = the general kernel alone -- what a
# small batch takes.  Every test runs both.
@pytest.fixture(scope="module", params=[2, 1], ids=["dec4", "general"])
def eng(request):
    import zstdmt_amd as z
    e = z.Engine(0)
    e.set_variant("brotli_dec", request.param)
    yield e
    e.close()


def _stream(name):
    ent = MAN[name]
    if "out_file" in ent:
        return open(os.path.join(BDIR, ent["out_file"]), "rb").read()
    if not H.have_bref():
        pytest.skip("stream not committed (size) and no reference build on this box")
    level, chunk, thunk = cases.BCASES[name]
    rv, st, _, _ = H.brotlimt_compress_via(H.bref(), thunk(), chunk, threads=2, level=level)
    assert rv == 0
    return st


@pytest.mark.parametrize("name", sorted(MAN))
def test_decode_golden(eng, name):
    ent = MAN[name]
    st = _stream(name)
    ro, rl, cap = E.walk_brotli_records(st)
    recs, status = eng.brotli_decompress_bytes(st, ro, rl, cap)
    assert status.tolist() == [0] * ent["frames"]
    out = b"".join(recs)
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]


@needs_ref
@pytest.mark.parametrize("level,chunk", [(0, 0), (1, 0), (1, 131072), (2, 65536), (3, 0), (4, 1 << 20), (5, 1 << 20),
                                         (6, 0), (9, 1 << 21), (11, 1 << 19)])
def test_decode_live_reference_streams(eng, level, chunk):
    n = 1 << 20 if level >= 10 else 5 << 20
    data = cases.text(n + 12345, 40 + level) + cases.rnd(70000, level) + bytes(200000) + \
        cases.english(300000, 9 + level)
    rv, st, _, stats = H.brotlimt_compress_via(H.bref(), data, chunk, threads=16, level=level)
    assert rv == 0
    ro, rl, cap = E.walk_brotli_records(st)
    assert len(rl) == stats[0]
    recs, status = eng.brotli_decompress_bytes(st, ro, rl, cap)
    assert (status == 0).all()
    assert b"".join(recs) == data


@needs_lib
@pytest.mark.parametrize("quality", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
def test_decode_libbrotli_streams(eng, quality):
    """raw encoder streams at every quality and several window sizes, many records per launch"""
    rng = random.Random(quality)
    payloads, plain = [], []
    for lgwin in (24, 22, 16, 10):
        for kind in range(5):
            n = rng.randrange(1, 60000 if quality >= 10 else 200000)
            d = [cases.text, cases.english][kind & 1](n, rng.randrange(1 << 30))
            if kind == 4:
                d = H.soup(rng, n)
            payloads.append(H.libbrotli_compress(d, quality, lgwin))
            plain.append(d)
    st = b"".join(H.brotli_record(p, (len(d) >> 16) + 1) for p, d in zip(payloads, plain))
    ro, rl, cap = E.walk_brotli_records(st)
    recs, status = eng.brotli_decompress_bytes(st, ro, rl, cap)
    assert (status == 0).all()
    assert recs == plain


@needs_lib
@pytest.mark.parametrize("quality", [1, 5, 9, 11])
def test_corrupt_streams_same_verdict_as_oracle(eng, quality):
    d = cases.english(30000, 61) + cases.text(20000, 62)
    st = H.libbrotli_compress(d, quality, 22)
    rng = random.Random(quality)
    variants = []
    for _ in range(300):
        s = bytearray(st)
        for _ in range(rng.choice([1, 1, 2, 3])):
            s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
        variants.append(bytes(s))
    for cut in (1, 2, 3, 50, len(st) // 2):
        variants.append(st[:len(st) - cut])
    stream = b"".join(H.brotli_record(s, 1) for s in variants)
    ro, rl, cap = E.walk_brotli_records(stream)
    recs, status = eng.brotli_decompress_bytes(stream, ro, rl, cap)
    n_ok = 0
    for s, o, stc in zip(variants, recs, status):
        want = H.oracle_brotli_decompress(s, 65536)
        if isinstance(want, int):
            assert stc != 0
        else:
            assert stc == 0 and o == want
            n_ok += 1
    assert n_ok < len(variants)


def test_capacity_is_the_hint(eng):
    st = _stream("b_text_3000_l1")
    ro, rl, cap = E.walk_brotli_records(st)
    recs, status = eng.brotli_decompress_bytes(st, ro, rl, np.array([2048], np.uint32))
    assert status[0] == 4
    recs, status = eng.brotli_decompress_bytes(st, ro, rl, np.array([3000], np.uint32))
    assert status[0] == 0 and len(recs[0]) == 3000


def test_more_records_than_resident_waves(eng):
    """the decoder is a persistent grid: 6000 small records exercise the per-wave record loop"""
    base = [_stream(n) for n in ("b_hello", "b_one", "b_text_3000_l1", "b_empty")]
    want = [cases.BCASES[n][2]() for n in ("b_hello", "b_one", "b_text_3000_l1", "b_empty")]
    st = b"".join(base[i % 4] for i in range(6000))
    ro, rl, cap = E.walk_brotli_records(st)
    recs, status = eng.brotli_decompress_bytes(st, ro, rl, cap)
    assert (status == 0).all()
    assert all(recs[i] == want[i % 4] for i in range(6000))


@needs_lib
def test_large_batch_takes_dec4_and_hands_over():
    """the product's own rule (variant 0): a batch of more records than the one-record kernel keeps resident goes to
    zmt_brotli_dec4_kernel, which decodes the quality 0..4 streams itself and hands the others (context modelling) and
    nothing else over; a damaged record gets the oracle's verdict, the rest of the batch is untouched by it"""
    import zstdmt_amd as z
    e = z.Engine(0)
    try:
        datas = [cases.text(3000, 1), cases.english(9000, 2), b"", cases.text(70000, 3), bytes(5000), cases.text(200, 4)]
        quals = [1, 5, 2, 0, 9, 4]
        streams = [H.libbrotli_compress(d, quality=q, lgwin=22) for d, q in zip(datas, quals)]
        bad = bytearray(streams[3])
        bad[len(bad) // 2] ^= 0x5A
        want_bad = H.oracle_brotli_decompress(bytes(bad), ((len(datas[3]) >> 16) + 1) << 16)   # the record's capacity
        n = 6000                                             # > 16 waves x 256 CUs
        parts = []
        for i in range(n):
            k = i % 6
            parts.append(H.brotli_record(bytes(bad) if i == 4005 else streams[k], (len(datas[k]) >> 16) + 1))
        st = b"".join(parts)
        ro, rl, cap = E.walk_brotli_records(st)
        recs, status = e.brotli_decompress_bytes(st, ro, rl, cap)
        for i in range(n):
            if i == 4005:                                  # (4005 % 6 == 3: the damaged copy sits in a slot of its own capacity)
                assert (status[i] != 0) if isinstance(want_bad, int) else (status[i] == 0 and recs[i] == want_bad)
            else:
                assert status[i] == 0 and recs[i] == datas[i % 6], i
    finally:
        e.close()


def test_garbage_streams_same_verdict_as_oracle(eng):
    """Records of random bytes (and of random bytes behind a plausible header): nothing hangs, nothing
    is written outside the record's slot, and the verdict -- in the rare accepted case also the
    content -- is the oracle's."""
    rng = random.Random(99)
    variants = []
    for i in range(3000):
        n = rng.choice([1, 2, 3, 5, 8, 17, 40, 100, 300, 1000])
        body = bytes(rng.randrange(256) for _ in range(n))
        if i % 3 == 0:      # WBITS + a compressed meta-block header of a few hundred bytes
            body = bytes([0x1B, 0x0A + (rng.randrange(4) << 4), rng.randrange(4)]) + body
        elif i % 3 == 1:    # all-zero or all-one tails
            body = body[:n // 2] + bytes([rng.choice([0, 0xFF])]) * (n - n // 2)
        variants.append(body)
    stream = b"".join(H.brotli_record(s, 1) for s in variants)
    ro, rl, cap = E.walk_brotli_records(stream)
    recs, status = eng.brotli_decompress_bytes(stream, ro, rl, cap)
    for s, o, stc in zip(variants, recs, status):
        want = H.oracle_brotli_decompress(s, 65536)
        if isinstance(want, int):
            assert stc != 0, s.hex()
        else:
            assert stc == 0 and o == want, s.hex()
