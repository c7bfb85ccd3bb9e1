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


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("name", SMALL)
def test_emu_decompress(name, variant):
    chunk, thunk = CASES[name]
    data = thunk()
    stream = H.oracle_compress(data, chunk)
    out, status = E.decompress(stream, variant)
    assert status.tolist() == [0] * len(status)
    assert out == data
