"""snappy.hip under the fiber emulator (tests/emu): the encoder's streams must decode to the input with
the oracle, the emulated decoder and -- where present -- libsnappy; the decoder must give the oracle's
verdict and bytes on golden, hand-built and damaged streams."""
import json
import os
import random

import numpy as np
import pytest

import emu_driver as E
import helpers as H
from golden import cases
from test_snappy_oracle import BAD, HAND, MAN, SDIR

@pytest.fixture(params=["element", "batched"], autouse=True)
def decoder_variant(request, monkeypatch):
    """Every test runs with both decoders of snappy.hip: zmt_snappy_dec_kernel (one element at a time, the
    default) and zmt_snappy_dec2_kernel (64 elements per batch, gpumt_set_variant("snappy_dec", 1))."""
    monkeypatch.setenv("EMU_SNAPPY_DEC", "1" if request.param == "batched" else "0")


ENC_CASES = {
    "empty": (65536, lambda: b""),
    "one": (65536, lambda: b"x"),
    "hello": (65536, lambda: b"hello hello hello hello, hello!"),
    "text_300": (65536, lambda: cases.text(300, 2)),
    "text_70k": (65536, lambda: cases.text(70000, 3)),                 # two records, ragged
    "text_exact_2x64k": (65536, lambda: cases.text(131072, 4)),
    "text_300k_chunk128k": (131072, lambda: cases.text(300000, 5)),    # records of two 64 KiB blocks
    "text_chunk_4k": (4096, lambda: cases.text(30000, 6)),
    "random_100k": (65536, lambda: cases.rnd(100000, 3)),              # does not shrink: long literals
    "zeros_200k": (65536, lambda: bytes(200000)),                      # matches of a whole block
    "period_300": (1 << 20, lambda: cases.rep(cases.rnd(300, 9), 150000)),
    "period_3": (65536, lambda: b"abc" * 30000),
    "mixed": (32768, lambda: cases.text(50000, 4) + bytes(70000) + cases.rnd(3000, 5) + cases.text(100000, 6)),
    "dense_sequences": (65536, lambda: H.dense_sequences(100000)),     # a match every few bytes
    "long_literal_runs": (65536, lambda: b"".join(cases.rnd(200, i) + cases.text(40, 1) for i in range(300))),
}


def _payloads(st):
    ro, rl, hints = E.walk_snappy_records(st)
    return [st[o:o + n] for o, n in zip(ro.tolist(), rl.tolist())], hints


@pytest.mark.parametrize("name", sorted(ENC_CASES))
def test_emu_encode_decompress_identical(name):
    chunk, thunk = ENC_CASES[name]
    data = thunk()
    st = E.snappy_compress(data, chunk)
    assert H.oracle_snappymt_decompress(st, len(data) + 64) == data
    pl, hints = _payloads(st)
    frames = max(1, -(-len(data) // chunk))
    assert len(pl) == frames
    # the reference's hint: 64 KiB units of the chunk size, (n >> 16) + 1 for a short last chunk
    for i, h in enumerate(hints):
        n = min(chunk, len(data) - i * chunk)
        assert h == ((n >> 16) + 1 if n < chunk else chunk >> 16)
    bound = 32 + chunk + chunk // 6
    assert all(len(p) <= bound for p in pl)                       # snappy_max_compressed_length
    if H.have_libsnappy():
        assert b"".join(H.libsnappy_decompress(p, chunk) for p in pl) == data
    out, status = E.snappy_decompress(st)
    assert (status == 0).all() and out == data
    assert E.snappy_compress(data, chunk, grid=1) == st            # independent of the persistent grid


def test_emu_encoder_compresses():
    data = cases.text(1 << 20, 5)
    st = E.snappy_compress(data, 65536, grid=8)
    assert len(data) / len(st) > 1.6
    if H.have_libsnappy():
        assert len(st) < 1.06 * len(H.snappymt_stream(data, 65536))


@pytest.mark.parametrize("name", sorted(MAN))
def test_emu_decode_golden(name):
    ent = MAN[name]
    if "out_file" not in ent:
        pytest.skip("digest-only case")
    st = open(os.path.join(SDIR, ent["out_file"]), "rb").read()
    out, status = E.snappy_decompress(st, grid=3)
    assert (status == 0).all() and len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]


def _one(payload, cap):
    st = H.snappy_record(payload, 1)
    return E.snappy_decompress(st, caps=[cap])


@pytest.mark.parametrize("name", sorted(HAND))
def test_emu_decode_hand_built_elements(name):
    payload, want = HAND[name]
    out, status = _one(payload, len(want))
    assert status[0] == 0 and out == want
    out, status = _one(payload, len(want) - 1)
    assert status[0] == 4                                          # GPUMT_ST_SIZE_MISMATCH


@pytest.mark.parametrize("name", sorted(n for n in BAD if BAD[n]))
def test_emu_decode_rejects(name):
    out, status = _one(BAD[name], 4096)
    assert status[0] == 3 and out == b""                           # GPUMT_ST_BAD_BLOCK


@pytest.mark.parametrize("seed", range(12))
def test_emu_damaged_streams_get_the_oracles_verdict(seed):
    rng = random.Random(8100 + seed)
    n = rng.choice([5, 300, 4096, 70000, rng.randrange(1, 100000)])
    data = H.soup(rng, n) if seed % 3 == 0 else cases.text(n, seed)
    own = seed % 2 == 0 or not H.have_libsnappy()
    st = E.snappy_compress(data, 65536) if own else H.snappymt_stream(data, 65536)
    pl, _ = _payloads(st)
    for _ in range(12):
        p = bytearray(rng.choice(pl))
        for _ in range(rng.randrange(1, 3)):
            k = rng.randrange(len(p))
            p[k] = p[k] ^ (1 << rng.randrange(8)) if rng.random() < 0.6 else rng.randrange(256)
        if rng.random() < 0.1:
            p = p[:rng.randrange(1, len(p) + 1)]
        p = bytes(p)
        cap = 65536 + 4096
        want = H.oracle_snappy_decompress(p, cap)
        out, status = _one(p, cap)
        if want is None:
            assert status[0] != 0 and out == b""
        else:
            assert status[0] == 0 and out == want


def test_emu_records_side_by_side():
    """Good and bad records in one launch keep their own verdicts and places."""
    a, b = cases.text(5000, 1), cases.rnd(700, 2)
    good_a, good_b = H.libsnappy_compress(a) if H.have_libsnappy() else _payloads(E.snappy_compress(a, 65536))[0][0], \
        _payloads(E.snappy_compress(b, 65536))[0][0]
    bad = bytearray(good_a)
    bad[len(bad) // 2] ^= 0x55
    st = H.snappy_record(good_a, 1) + H.snappy_record(bytes(bad), 1) + H.snappy_record(good_b, 1) + H.snappy_record(b"\x00", 1)
    out, status = E.snappy_decompress(st, grid=2, caps=[5000, 5000, 700, 0])
    want_mid = H.oracle_snappy_decompress(bytes(bad), 5000)
    assert status[0] == 0 and status[2] == 0 and status[3] == 0
    assert (status[1] == 0) == (want_mid is not None)
    assert out == a + (want_mid or b"") + b
