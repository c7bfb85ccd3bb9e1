"""brotli decoder kernel (zstdmt_amd/csrc/hip/brotli_dec.hip) compiled for the host fiber emulator,
against the golden streams of the reference build and the oracle.  CPU only; the same checks run on
the device in test_gpu_brotli.py at full sizes."""
import json
import os
import random

import numpy as np
import pytest

import emu_driver as E
import helpers as H
from golden import cases

GDIR = os.path.join(H.GOLDEN_DIR, "brotli")
with open(os.path.join(GDIR, "manifest.json")) as f:
    MAN = json.load(f)["cases"]

SMALL = ["b_empty", "b_one", "b_hello", "b_text_3000_l1", "b_text_64k_l0", "b_english_l11", "b_period_300"]
needs_lib = pytest.mark.skipif(not H.have_libbrotli(), reason="libbrotli 1.0.9 not in this image")


@pytest.mark.parametrize("name", SMALL)
def test_golden_streams(name):
    ent = MAN[name]
    with open(os.path.join(GDIR, ent["out_file"]), "rb") as f:
        st = f.read()
    recs, status = E.brotli_decompress(st)
    assert (status == 0).all()
    out = b"".join(recs)
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]


def test_multi_record_persistent_waves():
    """five records on two persistent waves: scratch and register state carry nothing over"""
    names = ["b_hello", "b_english_l11", "b_one", "b_text_3000_l1", "b_empty"]
    st = b""
    want = []
    for n in names:
        with open(os.path.join(GDIR, MAN[n]["out_file"]), "rb") as f:
            st += f.read()
        want.append(H.sha256(cases.BCASES[n][2]()))
    recs, status = E.brotli_decompress(st, grid=2)
    assert (status == 0).all()
    assert [H.sha256(r) for r in recs] == want


@needs_lib
@pytest.mark.parametrize("seed", range(4))
def test_four_streams_per_wave_mixed_records(seed):
    """zmt_brotli_dec4_kernel runs four records per wave in lockstep: records of very different lengths and shapes in
    one wave -- empty, tiny, 100 KiB+, streams it decodes itself (qualities 0..4) next to streams it hands over to the
    general kernel (context modelling, qualities 5+) and a damaged one -- must not disturb each other"""
    import random
    rng = random.Random(900 + seed)
    parts, want = [], []
    nrec = rng.choice([3, 5, 6, 9])
    for i in range(nrec):
        n = rng.choice([0, 1, 5, 100, 3000, 20000, 70000, rng.randrange(1, 150000)])
        d = cases.text(n, seed=rng.randrange(1 << 30)) if rng.random() < 0.7 else H.soup(rng, n)
        q = rng.choice([0, 1, 1, 2, 3, 4, 5, 9])
        st = bytearray(H.libbrotli_compress(d, quality=q, lgwin=rng.choice([16, 18, 22])))
        if i == 1 and len(st) > 20:
            st[len(st) // 2] ^= 0x55                      # one damaged record among them
            d = H.oracle_brotli_decompress(bytes(st), ((n >> 16) + 1) << 16)    # (the record's capacity; a negative status when the oracle rejects it)
        parts.append(H.brotli_record(bytes(st), (n >> 16) + 1))
        want.append(d)
    recs, status = E.brotli_decompress(b"".join(parts), grid=rng.choice([1, 2, 3]))
    for i in range(nrec):
        if isinstance(want[i], int):                        # the oracle's negative status: it rejects the damaged record
            assert status[i] != 0
        else:
            assert status[i] == 0 and recs[i] == want[i], i


@needs_lib
@pytest.mark.parametrize("quality", [0, 2, 5, 9, 11])
def test_live_streams_match_oracle(quality):
    d = cases.english(14000, 40 + quality) + cases.text(6000, 41) + bytes(300) + cases.rnd(500, 42)
    for lgwin in (22, 10):
        st = H.libbrotli_compress(d, quality, lgwin)
        assert H.oracle_brotli_decompress(st, len(d)) == d
        recs, status = E.brotli_decompress(H.brotli_record(st, 1))
        assert status[0] == 0 and recs[0] == d


@needs_lib
def test_corrupt_streams_same_verdict():
    d = cases.english(5000, 51) + cases.text(3000, 52)
    st = H.libbrotli_compress(d, 6, 22)
    rng = random.Random(5)
    recs = []
    for _ in range(24):
        s = bytearray(st)
        for _ in range(rng.choice([1, 2])):
            s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
        recs.append(bytes(s))
    stream = b"".join(H.brotli_record(s, 1) for s in recs)
    outs, status = E.brotli_decompress(stream, grid=3)
    for s, o, stc in zip(recs, outs, status):
        want = H.oracle_brotli_decompress(s, 65536)
        if isinstance(want, int):
            assert stc != 0
        else:
            assert stc == 0 and o == want


def test_capacity_and_truncation():
    ent = MAN["b_text_3000_l1"]
    with open(os.path.join(GDIR, ent["out_file"]), "rb") as f:
        st = f.read()
    ro, rl, cap = E.walk_brotli_records(st)
    cap2 = np.array([2048], np.uint32)  # capacity below the decoded size
    recs, status = E.brotli_decompress(st, rec=(ro, rl, cap2))
    assert status[0] == 4
    rl2 = rl.copy()
    rl2[0] -= 7  # payload cut short
    recs, status = E.brotli_decompress(st, rec=(ro, rl2, cap))
    assert status[0] == 3


# ---- encoder kernel: decompress-identical, checked by the oracle (and libbrotlidec where present) ----
ENC = {
    "empty": (b"", 65536), "one": (b"a", 65536), "hello": (b"hello world hello world hello world", 65536),
    "text_3000": (cases.text(3000, 5), 65536),
    "text_66k": (cases.text(66000, 6), 131072),            # MLEN - 1 >= 65536: 5 nibbles
    "english_40k": (cases.english(40000, 2), 65536),       # MLEN - 1 < 65536: 4 nibbles
    "zeros": (bytes(30000), 131072),
    "random": (cases.rnd(8000, 3), 65536),                 # does not shrink: uncompressed meta-block
    "two_blocks_two_chunks": (cases.text(140000, 7), 131072),
    "one_symbol_then_text": (b"z" * 5000 + cases.text(5000, 9), 65536),
    # one command and one distance in the meta-block: both alphabets get a zero-bit code
    "single_command": (bytes((b & 15) * 17 for b in cases.rnd(384, 5)) + b"\xff" * 616, 65536),
}


@pytest.mark.parametrize("name", sorted(ENC))
def test_encoder_round_trip(name):
    data, chunk = ENC[name]
    st = E.brotli_compress(data, chunk, grid=2)
    assert H.oracle_brotlimt_decompress(st, len(data) + 65536) == data
    if H.have_libbrotli():
        ro, rl, cap = E.walk_brotli_records(st)
        got = b"".join(H.libbrotli_decompress(st[int(o):int(o) + int(n)], int(c)) for o, n, c in zip(ro, rl, cap))
        assert got == data
    # what the encoder writes, the emulated decoder reads
    recs, status = E.brotli_decompress(st, grid=2)
    assert (status == 0).all() and b"".join(recs) == data


@pytest.mark.parametrize("level", [5, 11])
@pytest.mark.parametrize("name", ["two_blocks_two_chunks", "zeros", "random", "single_command"])
def test_encoder_quality_tiers_round_trip(name, level):
    """qualities 4-8 and 9-11 run the encoder's larger-table tiers (gpumt_brotli_level_tier): the same bar"""
    data, chunk = ENC[name]
    st = E.brotli_compress(data, chunk, grid=2, level=level)
    assert H.oracle_brotlimt_decompress(st, len(data) + 65536) == data
    recs, status = E.brotli_decompress(st, grid=2)
    assert (status == 0).all() and b"".join(recs) == data
    assert E.brotli_compress(data, chunk, grid=1, level=level) == st


def test_encoder_ratio_is_monotone_in_quality():
    """the reference hands the level to BrotliEncoderCompress (lib/brotli-mt_compress.c:269-272)"""
    data = cases.text(1 << 20, 5)
    sizes = [len(E.brotli_compress(data, 1 << 20, grid=8, level=q)) for q in (1, 5, 11)]
    assert sizes[0] > sizes[1] > sizes[2] and sizes[0] / sizes[2] > 1.04
    assert len(E.brotli_compress(data, 1 << 20, grid=8, level=3)) == sizes[0]
    assert len(E.brotli_compress(data, 1 << 20, grid=8, level=8)) == sizes[1]


def test_encoder_is_deterministic_across_grids():
    data = cases.text(140000, 17)
    assert E.brotli_compress(data, 65536, grid=1) == E.brotli_compress(data, 65536, grid=3)
