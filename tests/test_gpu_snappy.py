"""snappy-mt on the device through the C ABI: SNAPPYMT_* of libzstdmt_amd.so (include/snappy-mt.h) over
the kernels of snappy.hip, against the committed vectors (payloads written by libsnappy 1.1.8), the
oracle, and the image's libsnappy where it is on this box.  Decoder: byte-exact content and the
oracle's verdicts; encoder: decompress-identical."""
import ctypes as C
import json
import os
import random
import struct

import pytest

import helpers as H
from golden import cases
from test_emu_snappy import ENC_CASES
from test_snappy_oracle import BAD, HAND, MAN, SDIR

pytestmark = pytest.mark.gpu

ERR = lambda e: C.c_size_t(-e).value  # noqa: E731
E_MEM, E_READ, E_WRITE, E_DATA, E_FC, E_FD, E_PARAM, E_LIB, E_CANCEL = range(1, 10)


@pytest.fixture(scope="module")
def lib():
    from zstdmt_amd._native import lib_path
    return H.bind_lz4mt(C.CDLL(lib_path()), "SNAPPYMT_")


# every shipped decoder kernel: 0 = zmt_snappy_dec_kernel (element by element), 1 = zmt_snappy_dec2_kernel
# (batched).  gpumt_open reads GPUMT_SNAPPY_DEC when a context is created, i.e. per SNAPPYMT_createDCtx.
VARIANTS = (0, 1)


@pytest.fixture(params=VARIANTS, ids=lambda v: "dec%d" % v)
def dec(request, monkeypatch):
    monkeypatch.setenv("GPUMT_SNAPPY_DEC", str(request.param))
    return request.param


@pytest.mark.parametrize("name", sorted(MAN))
def test_decompress_golden(lib, dec, name):
    ent = MAN[name]
    if "out_file" in ent:
        st = open(os.path.join(SDIR, ent["out_file"]), "rb").read()
    elif H.have_libsnappy():
        from gen_golden_snappy import SCASES
        st = H.snappymt_stream(SCASES[name][1](), ent["chunk"])
    else:
        pytest.skip("stream not committed (size) and no libsnappy on this box")
    rv, out, io, stats = H.snappymt_decompress_via(lib, st, threads=4)
    assert rv == 0 and len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]
    assert stats == (ent["frames"], len(st), ent["in_len"])
    assert len(io.writes) == ent["frames"]


@pytest.mark.parametrize("name", sorted(ENC_CASES))
def test_compress_is_decompress_identical(lib, name):
    chunk, thunk = ENC_CASES[name]
    data = thunk()
    rv, st, io, stats = H.snappymt_compress_via(lib, data, chunk, threads=4)
    assert rv == 0
    frames = max(1, -(-len(data) // chunk))
    assert stats == (frames, len(data), len(st)) and len(io.writes) == frames
    assert all(want == chunk for want, _ in io.reads)
    assert H.oracle_snappymt_decompress(st, len(data) + 64) == data
    if H.have_libsnappy():
        at, parts = 0, []
        while at < len(st):
            csz = struct.unpack_from("<I", st, at + 8)[0]
            parts.append(H.libsnappy_decompress(st[at + 16:at + 16 + csz], chunk))
            at += 16 + csz
        assert b"".join(parts) == data
    rv, back, _, dstats = H.snappymt_decompress_via(lib, st, threads=4)
    assert rv == 0 and back == data and dstats == (frames, len(st), len(data))


def test_large_default_chunks_many_batches(lib, dec):
    """200 MiB at the default 64 KiB chunk: 3 200 records, several device batches each way."""
    data = cases.text(200 << 20, 61)
    rv, st, io, stats = H.snappymt_compress_via(lib, data, 0, threads=8)
    assert rv == 0 and stats[0] == 3200 and len(data) / len(st) > 1.6
    for i in (0, 1023, 1024, 3199):
        off = sum(io.writes[:i])
        assert H.oracle_snappymt_decompress(st[off:off + io.writes[i]], 65536) == data[i * 65536:(i + 1) * 65536]
    rv, out, _, dstats = H.snappymt_decompress_via(lib, st, threads=8)
    assert rv == 0 and out == data and dstats == (3200, len(st), len(data))
    if H.have_libsnappy():
        ref = H.snappymt_stream(data[:32 << 20], 65536)
        rv, out, _, _ = H.snappymt_decompress_via(lib, ref, threads=8)
        assert rv == 0 and out == data[:32 << 20]


def test_large_chunks(lib):
    data = cases.text(9 << 20, 62) + cases.rnd(1 << 20, 3) + bytes(3 << 20)
    rv, st, _, stats = H.snappymt_compress_via(lib, data, 4 << 20, threads=2)
    assert rv == 0 and stats[0] == 4
    assert H.oracle_snappymt_decompress(st, len(data) + 64) == data
    rv, out, _, _ = H.snappymt_decompress_via(lib, st, threads=2)
    assert rv == 0 and out == data


@pytest.mark.parametrize("name", sorted(HAND))
def test_hand_built_elements(lib, dec, name):
    payload, want = HAND[name]
    rv, out, _, _ = H.snappymt_decompress_via(lib, H.snappy_record(payload, 1) * 3, threads=2)
    assert rv == 0 and out == want * 3


@pytest.mark.parametrize("name", sorted(n for n in BAD if BAD[n]))
def test_rejects(lib, dec, name):
    good = H.snappy_record(HAND["copy4"][0], 1)
    rv, out, _, _ = H.snappymt_decompress_via(lib, good + H.snappy_record(BAD[name], 1) + good, threads=2)
    assert rv == ERR(E_FD)


@pytest.mark.parametrize("seed", range(10))
def test_damaged_streams_get_the_oracles_verdict(lib, dec, seed):
    rng = random.Random(9100 + seed)
    data = H.soup(rng, rng.randrange(1, 400000)) if seed % 3 == 0 else cases.text(rng.randrange(1, 400000), seed)
    own = seed % 2 == 0 or not H.have_libsnappy()
    if own:
        rv, st, _, _ = H.snappymt_compress_via(lib, data, 65536, threads=2)
        assert rv == 0
    else:
        st = H.snappymt_stream(data, 65536)
    for _ in range(10):
        bad = bytearray(st)
        k = rng.randrange(16, len(bad))
        bad[k] = bad[k] ^ (1 << rng.randrange(8)) if rng.random() < 0.6 else rng.randrange(256)
        bad = bytes(bad)
        want = H.oracle_snappymt_decompress(bad, len(data) + 70000)
        rv, out, _, _ = H.snappymt_decompress_via(lib, bad, threads=3)
        if want is None:
            assert lib.SNAPPYMT_isError(rv)
        else:
            assert rv == 0 and out == want


def test_callback_errors_and_arguments(lib):
    data = cases.text(400000, 63)
    rv, st, _, _ = H.snappymt_compress_via(lib, data, 65536, threads=2)
    assert rv == 0
    io = H.MemIO(st, fail_read_at=2, read_rv=-2)
    ctx = lib.SNAPPYMT_createDCtx(2, 0)
    assert lib.SNAPPYMT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(E_CANCEL)
    lib.SNAPPYMT_freeDCtx(ctx)
    io = H.MemIO(data, fail_write_at=3, write_rv=-1)
    ctx = lib.SNAPPYMT_createCCtx(2, 0, 65536)
    assert lib.SNAPPYMT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_READ)   # sic: mt_error
    io = H.MemIO(data)
    assert lib.SNAPPYMT_compressCCtx(ctx, C.byref(io.rdwr)) == 0 and io.result() == st
    lib.SNAPPYMT_freeCCtx(ctx)
    assert not lib.SNAPPYMT_createCCtx(0, 1, 0) and not lib.SNAPPYMT_createDCtx(129, 0)
    rv, _, _, _ = H.snappymt_decompress_via(lib, st[:-3])
    assert rv == ERR(E_DATA)
    rv, _, _, _ = H.snappymt_decompress_via(lib, b"\x28\xb5\x2f\xfd" + st)
    assert rv == ERR(E_DATA)


def test_command_line_tool(tmp_path):
    """snappy-mt / unsnappy-mt / snappycat-mt (programs/zmt_cli.c -DZMT_SNAPPY; reference programs/snappy-mt.c:
    suffix .snp, levels 0..1, default 0)."""
    import subprocess
    BIN = os.path.join(H.ROOT, "zstdmt_amd", "bin")
    data = cases.text(3 << 20, 64) + bytes(100000)
    f = tmp_path / "blob"
    f.write_bytes(data)
    p = subprocess.run([os.path.join(BIN, "snappy-mt"), "-T4", "-B", str(f)], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    out = (tmp_path / "blob.snp").read_bytes()
    assert not f.exists() and H.oracle_snappymt_decompress(out, len(data) + 64) == data
    lines = p.stderr.decode().splitlines()
    frames = -(-len(data) // 65536)
    assert lines[0] == "Level;Threads;InSize;OutSize;Frames" and lines[1] == f"0;4;{len(data)};{len(out)};{frames}"
    p = subprocess.run([os.path.join(BIN, "snappycat-mt"), str(tmp_path / "blob.snp")], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout == data
    p = subprocess.run([os.path.join(BIN, "unsnappy-mt"), str(tmp_path / "blob.snp")], capture_output=True, timeout=300)
    assert p.returncode == 0 and f.read_bytes() == data
    p = subprocess.run([os.path.join(BIN, "snappy-mt"), "-2", "-c"], input=b"x", capture_output=True, timeout=60)
    assert p.returncode != 0                                   # level above LEVEL_MAX
