"""Live cross-check of the oracle against the reference's own lz4-mt code (oracle/_ref), where
that has been built (this container; the .so also travels to the GPU box).  Skipped otherwise --
the committed golden vectors (test_oracle_golden.py) are the portable pin."""
import random

import pytest

import helpers as H
from cases import rnd, text

pytestmark = pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")


def _mix(rng, n):
    """Borderline-compressible soup: exercises the stored-vs-compressed bail-out points."""
    out = bytearray()
    while len(out) < n:
        k = rng.choice([0, 1, 2, 3, 4])
        m = rng.randrange(1, 5000)
        if k == 0:
            out += rnd(m, rng.randrange(1 << 30))
        elif k == 1:
            out += bytes([rng.randrange(256)]) * m
        elif k == 2:
            out += text(m, seed=rng.randrange(1 << 30))
        elif k == 3:
            unit = rnd(rng.randrange(1, 40), rng.randrange(1 << 30))
            out += (unit * (m // len(unit) + 1))[:m]
        else:
            back = rng.randrange(1, len(out) + 1) if out else 0
            if back:
                start = len(out) - back
                out += out[start:start + m]
    return bytes(out[:n])


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_compress_bit_exact(seed):
    rng = random.Random(seed)
    n = rng.choice([0, 1, 11, 12, 13, 14, 65535, 65536, 65537, 65548, 131071, 131072, 200000,
                    rng.randrange(1, 400000)])
    chunk = rng.choice([65536, 131072, 131072, 100000, 262144, 1 << 20])
    data = _mix(rng, n)
    rv, s_ref, _, _ = H.lz4mt_compress_via(H.ref(), data, chunk, threads=rng.choice([1, 3]))
    assert rv == 0
    assert H.oracle_compress(data, chunk) == s_ref
    rv, back, _, _ = H.lz4mt_decompress_via(H.ref(), s_ref, threads=2)
    assert rv == 0 and back == data
    assert H.oracle_decompress(s_ref, max(n, 65536)) == data


@pytest.mark.parametrize("frac", [0.93, 0.97, 0.99, 0.995, 1.0])
def test_borderline_blocks(frac):
    """Blocks whose LZ4 size lands around len-1: the capacity checks decide stored vs compressed."""
    rng = random.Random(int(frac * 1000))
    n = 131072
    k = int(n * frac)
    data = bytearray(rnd(n, 42))
    # sprinkle short repeats so the encoder saves a few bytes here and there
    for _ in range((n - k) // 6):
        p = rng.randrange(100, n - 16)
        q = rng.randrange(0, p - 8)
        data[p:p + 8] = data[q:q + 8]
    data = bytes(data)
    rv, s_ref, _, _ = H.lz4mt_compress_via(H.ref(), data, 131072, threads=1)
    assert rv == 0 and H.oracle_compress(data, 131072) == s_ref


def test_mt_variants_agree():
    data = text(1 << 20) + rnd(100000, 2) + bytes(300000)
    s = H.oracle_compress(data, 131072)
    import ctypes as C
    lib = H.oracle()
    cap = lib.zo_lz4mt_compress_bound(len(data), 131072)
    out = C.create_string_buffer(cap)
    n = lib.zo_lz4mt_compress_mt(data, len(data), 131072, out, cap, 4)
    assert out.raw[:n] == s
    back = C.create_string_buffer(len(data))
    m = lib.zo_lz4mt_decompress_mt(s, len(s), back, len(data), 4)
    assert m == len(data) and back.raw == data


@pytest.mark.parametrize("level", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("seed", range(6))
def test_fuzz_compress_hc_bit_exact(seed, level):
    """LZ4 HC levels that run the hash-chain parser (lz4hc_oracle.c; level 9 adds the repeated-pattern
    analysis) against the reference
    library (liblz4 1.9.3 behind lib/lz4-mt_compress.c:281), incl. linked blocks and stored blocks."""
    rng = random.Random(7000 + 31 * seed + level)
    n = rng.choice([0, 1, 12, 13, 14, 65536, 65537, 131072, 131073, rng.randrange(1, 500000),
                    rng.randrange(1, 500000)])
    chunk = rng.choice([65536, 131072, 131072, 100000, 262144, 1 << 20])
    data = _mix(rng, n)
    rv, s_ref, _, _ = H.lz4mt_compress_via(H.ref(), data, chunk, threads=rng.choice([1, 3]), level=level)
    assert rv == 0
    assert H.oracle_compress_level(data, chunk, level) == s_ref
    assert H.oracle_decompress(s_ref, max(n, 65536)) == data


def _runs(rng, n):
    """byte runs of every length next to short repeats and text: what the pattern analysis of level 9 is for"""
    out = bytearray()
    while len(out) < n:
        k = rng.randrange(6)
        if k == 0:
            out += bytes([rng.randrange(256)]) * rng.choice([3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 300, 5000, 70000])
        elif k == 1:
            out += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 6)))
        elif k == 2:
            out += text(rng.randrange(1, 400), rng.randrange(1 << 20))
        elif k == 3:
            out += (bytes([rng.randrange(256)]) * rng.randrange(1, 5) + bytes([rng.randrange(256)])) * rng.randrange(1, 40)
        elif k == 4 and len(out) > 10:
            a = rng.randrange(len(out))
            out += out[a:a + rng.randrange(4, 300)]
        else:
            out += b"\0" * rng.randrange(1, 2000)
    return bytes(out[:n])


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_level9_runs(seed):
    rng = random.Random(9900 + seed)
    n = rng.choice([rng.randrange(1, 400000), 65536, 131072, 200000])
    chunk = rng.choice([65536, 131072, 100000, 262144])
    data = _runs(rng, n)
    for level in (9, rng.choice([3, 5, 8]), rng.choice([10, 11, 12])):
        rv, s_ref, _, _ = H.lz4mt_compress_via(H.ref(), data, chunk, threads=1, level=level)
        assert rv == 0
        assert H.oracle_compress_level(data, chunk, level) == s_ref, (level, n, chunk)


@pytest.mark.parametrize("level", [3, 8, 9, 10, 11, 12])
def test_hc_text_and_runs(level):
    data = text(700000, 21) + bytes(200000) + (text(97, 22) * 3000) + rnd(90000, 23) + text(150000, 24)
    for chunk in (131072, 4 << 20):
        rv, s_ref, _, _ = H.lz4mt_compress_via(H.ref(), data, chunk, threads=2, level=level)
        assert rv == 0 and H.oracle_compress_level(data, chunk, level) == s_ref
