"""The host engines on CPU: LZ4MT_* / ZSTDCB_* / BROTLIMT_* / SNAPPYMT_* of tests/emu/libzstdmt_emu_host.so -- the
unchanged engine sources (zstdmt_amd/csrc/host/*.c) linked over an emulated device boundary
(tests/emu/emu_gpumt.cpp: the kernels compiled as host C++ under the fiber emulator) -- driven through
the reference's callback protocol like the `-m gpu` API tests, on inputs small enough for the
emulator.  GPUMT_BATCH_KB makes a device batch hold a handful of records, so the three-role pipeline,
the slot dealing over several (emulated) devices and the incremental plain-stream paths all run
several rounds.  TEST HARNESS ONLY: the product library binds the same engines to gpumt.hip."""
import ctypes as C
import os
import struct
import subprocess
import threading

import pytest

import helpers as H
from golden import cases

EMU_DIR = os.path.join(H.ROOT, "tests", "emu")
ERR = lambda e: C.c_size_t(-e).value  # noqa: E731  (size_t)-enum
# lz4-mt / brotli-mt codes; zstd-mt has init_missing = 2 in between (lib/zstd-mt.h:69-81)
E_MEM, E_READ, E_WRITE, E_DATA, E_FC, E_FD, E_PARAM, E_LIB, E_CANCEL = range(1, 10)
Z_MEM, Z_INIT, Z_READ, Z_WRITE, Z_DATA, Z_FC, Z_FD, Z_PARAM, Z_LIB, Z_CANCEL = range(1, 11)

CHUNK = 4096


@pytest.fixture(scope="module")
def lib():
    H.locked_make(EMU_DIR, "libzstdmt_emu_host.so", stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    old = os.environ.get("GPUMT_BATCH_KB")
    os.environ["GPUMT_BATCH_KB"] = "16"      # read once, at the library's first batch
    L = C.CDLL(os.path.join(EMU_DIR, "libzstdmt_emu_host.so"))
    for pfx in ("LZ4MT_", "ZSTDCB_", "BROTLIMT_", "SNAPPYMT_"):
        H.bind_lz4mt(L, pfx)
    L.emu_gpumt_launches.restype = C.c_ulonglong
    L.emu_gpumt_launches.argtypes = [C.c_int]
    L.emu_gpumt_opened.restype = C.c_ulonglong
    # prime the cached batch size while the variable is set
    rv, _, _, _ = H.lz4mt_compress_via(L, b"prime", CHUNK)
    assert rv == 0
    yield L
    if old is None:
        os.environ.pop("GPUMT_BATCH_KB", None)
    else:
        os.environ["GPUMT_BATCH_KB"] = old


def _mixed(n, seed):
    t = cases.text(n, seed)
    return t[:n // 2] + bytes(n // 8) + cases.rnd(n // 8, seed) + t[n // 2:n // 2 + n // 4]


# ------------------------------------------------------------------------------------------ lz4-mt
LZ4_CASES = {
    "empty": (CHUNK, lambda: b""),
    "hello_5": (CHUNK, lambda: b"hello"),
    "one_full_chunk": (CHUNK, lambda: cases.text(CHUNK, 1)),
    "ragged_11_chunks": (CHUNK, lambda: _mixed(44000, 2)),          # 3 batches of 4 records
    "exact_8_chunks": (CHUNK, lambda: cases.text(8 * CHUNK, 3)),    # input ends on a batch boundary
    "chunk_larger_than_batch": (40000, lambda: cases.text(90000, 4)),  # one record per batch
}


@pytest.mark.parametrize("name", sorted(LZ4_CASES))
@pytest.mark.parametrize("threads", [1, 3])
def test_lz4mt_compress_matches_oracle(lib, name, threads):
    chunk, thunk = LZ4_CASES[name]
    data = thunk()
    rv, stream, io, stats = H.lz4mt_compress_via(lib, data, chunk, threads=threads, level=1)
    assert rv == 0
    assert stream == H.oracle_compress(data, chunk)
    frames = max(1, -(-len(data) // chunk))
    assert stats == (frames, len(data), len(stream))
    # one fn_read of exactly `inputsize` per chunk, one fn_write per record, in order
    assert all(want == chunk for want, _ in io.reads) and len(io.writes) == frames
    rv, out, io, dstats = H.lz4mt_decompress_via(lib, stream, threads=threads)
    assert rv == 0 and out == data and dstats == (frames, len(stream), len(data))
    wants = [w for w, _ in io.reads]
    assert wants[0] == 4 and wants[1] == 8 and wants[-1] == 12 and io.reads[-1][1] == 0


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["empty", "hello_5", "ragged_11_chunks", "exact_8_chunks"])
def test_lz4mt_callback_trace_equals_reference(lib, name):
    """Same requests, same sizes, same order as the reference library (T = 1: deterministic)."""
    chunk, thunk = LZ4_CASES[name]
    data = thunk()
    rv_r, s_r, io_r, st_r = H.lz4mt_compress_via(H.ref(), data, chunk, threads=1)
    rv_o, s_o, io_o, st_o = H.lz4mt_compress_via(lib, data, chunk, threads=1)
    assert (rv_o, s_o, st_o) == (rv_r, s_r, st_r)
    assert io_o.reads == io_r.reads and io_o.writes == io_r.writes
    rv_r, d_r, io_r, st_r = H.lz4mt_decompress_via(H.ref(), s_r, threads=1)
    rv_o, d_o, io_o, st_o = H.lz4mt_decompress_via(lib, s_r, threads=1)
    assert (rv_o, d_o, st_o) == (rv_r, d_r, st_r)
    assert io_o.reads == io_r.reads and io_o.writes == io_r.writes


@pytest.mark.parametrize("level", [2, 3, 6, 9, 10, 11, 12])
def test_lz4mt_levels(lib, level):
    data = _mixed(30000, 10 + level)
    rv, stream, _, stats = H.lz4mt_compress_via(lib, data, CHUNK, threads=2, level=level)
    assert rv == 0 and stream == H.oracle_compress_level(data, CHUNK, level)
    if H.have_ref():
        rv_r, s_r, _, st_r = H.lz4mt_compress_via(H.ref(), data, CHUNK, threads=1, level=level)
        assert rv_r == 0 and s_r == stream and st_r == stats


def test_lz4mt_arguments(lib):
    assert not lib.LZ4MT_createCCtx(0, 1, 0) and not lib.LZ4MT_createCCtx(129, 1, 0)
    assert not lib.LZ4MT_createCCtx(1, 0, 0) and not lib.LZ4MT_createCCtx(1, 13, 0)
    assert not lib.LZ4MT_createDCtx(0, 0) and not lib.LZ4MT_createDCtx(129, 0)
    io = H.MemIO(b"x")
    assert lib.LZ4MT_compressCCtx(None, C.byref(io.rdwr)) == ERR(E_PARAM)
    assert lib.LZ4MT_decompressDCtx(None, C.byref(io.rdwr)) == ERR(E_PARAM)
    assert lib.LZ4MT_getErrorString(ERR(E_PARAM)) == b"Compression parameter is out of bound"
    assert not lib.LZ4MT_isError(0) and lib.LZ4MT_isError(ERR(E_LIB))


@pytest.mark.parametrize("rv_cb,code", [(-1, E_READ), (-2, E_CANCEL), (-3, E_MEM), (-7, E_READ)])
@pytest.mark.parametrize("at", [0, 2, 6])          # first batch, inside it, a later batch
def test_lz4mt_read_failures(lib, rv_cb, code, at):
    data = cases.text(40000, 5)
    io = H.MemIO(data, fail_read_at=at, read_rv=rv_cb)
    ctx = lib.LZ4MT_createCCtx(2, 1, CHUNK)
    assert lib.LZ4MT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(code)
    lib.LZ4MT_freeCCtx(ctx)
    stream = H.oracle_compress(data, CHUNK)
    io = H.MemIO(stream, fail_read_at=at + 1, read_rv=rv_cb)
    ctx = lib.LZ4MT_createDCtx(2, 0)
    assert lib.LZ4MT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(code)
    lib.LZ4MT_freeDCtx(ctx)


@pytest.mark.parametrize("at", [0, 3, 7])
def test_lz4mt_write_failures_and_context_reuse(lib, at):
    """The reference routes write failures through mt_error(): -1 surfaces as read_fail.  A context
    that failed serves the next call; its counters carry on."""
    data = cases.text(40000, 6)
    io = H.MemIO(data, fail_write_at=at, write_rv=-1)
    ctx = lib.LZ4MT_createCCtx(2, 1, CHUNK)
    assert lib.LZ4MT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_READ)
    assert len(io.writes) == at
    io = H.MemIO(data)
    assert lib.LZ4MT_compressCCtx(ctx, C.byref(io.rdwr)) == 0
    assert io.result() == H.oracle_compress(data, CHUNK)
    lib.LZ4MT_freeCCtx(ctx)
    stream = io.result()
    io = H.MemIO(stream, fail_write_at=at, write_rv=-2)
    ctx = lib.LZ4MT_createDCtx(2, 0)
    assert lib.LZ4MT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(E_CANCEL)
    lib.LZ4MT_freeDCtx(ctx)


def test_dctx_serves_one_stream_like_reference(lib):
    """The reference picks the "magic already read" header path by its frame counter
    (lib/lz4-mt_decompress.c:200, likewise zstd-mt), and the counters carry over: a second stream on
    the same DCtx is a data error there.  Here a DCtx decodes stream after stream, counters carrying
    on as they do for a CCtx."""
    data = cases.text(12000, 8)
    st = H.oracle_compress(data, CHUNK)
    rv, zst, _, _ = H.zstdmt_compress_via(lib, data, ZCHUNK)
    assert rv == 0
    rv, bst, _, _ = H.brotlimt_compress_via(lib, data, BCHUNK)
    assert rv == 0
    libs = [(lib, "LZ4MT_", st, 0), (lib, "ZSTDCB_", zst, 0), (lib, "BROTLIMT_", bst, 0)]
    if H.have_ref():
        libs.append((H.ref(), "LZ4MT_", st, ERR(E_DATA)))
    if H.have_zref():
        libs.append((H.zref(), "ZSTDCB_", zst, ERR(Z_DATA)))
    for L, pfx, stream, second in libs:
        g = lambda n: getattr(L, pfx + n)  # noqa: E731
        ctx = g("createDCtx")(2, 0)
        io = H.MemIO(stream)
        assert g("decompressDCtx")(ctx, C.byref(io.rdwr)) == 0 and io.result() == data
        one = (g("GetFramesDCtx")(ctx), g("GetInsizeDCtx")(ctx), g("GetOutsizeDCtx")(ctx))
        io = H.MemIO(stream)
        assert g("decompressDCtx")(ctx, C.byref(io.rdwr)) == second
        if second == 0:
            assert io.result() == data
            two = (g("GetFramesDCtx")(ctx), g("GetInsizeDCtx")(ctx), g("GetOutsizeDCtx")(ctx))
            assert two == tuple(2 * v for v in one)
        g("freeDCtx")(ctx)


@pytest.mark.parametrize("mutate,code", [("badmagic", E_DATA), ("notlz4", E_DATA), ("skiplen", E_DATA),
                                         ("truncated", E_DATA), ("frame", E_LIB), ("checksum", E_LIB),
                                         ("secondmagic", E_DATA), ("late_record", E_LIB)])
def test_lz4mt_decompress_errors(lib, mutate, code):
    data = cases.text(40000, 7)
    s = bytearray(H.oracle_compress(data, CHUNK))
    c0 = struct.unpack_from("<I", s, 8)[0]
    if mutate == "badmagic":
        s[0] ^= 1
    elif mutate == "notlz4":
        s[:4] = b"\x28\xb5\x2f\xfd"
    elif mutate == "skiplen":
        s[4] = 8
    elif mutate == "truncated":
        s = s[:-10]
    elif mutate == "frame":
        s[12] ^= 1
    elif mutate == "checksum":
        s[-1] ^= 0x40
    elif mutate == "secondmagic":
        s[12 + c0] ^= 1
    elif mutate == "late_record":          # a record of the third batch
        at = 0
        for _ in range(9):
            at += 12 + struct.unpack_from("<I", s, at + 8)[0]
        s[at + 12] ^= 1
    rv, out, io, _ = H.lz4mt_decompress_via(lib, bytes(s), threads=2)
    assert rv == ERR(code), (rv, lib.LZ4MT_getErrorString(rv))
    if mutate == "late_record":            # what was written before the error surfaced is a prefix, whole records
        got = io.result()
        assert len(got) % CHUNK == 0 and len(got) <= 9 * CHUNK and got == data[:len(got)]


def _plain_lz4():
    a = cases.text(30000, 81)
    b = cases.rnd(3000, 82) + bytes(20000) + cases.text(30000, 83)
    return {
        "linked_64k_blocks": ([H.liblz4_frame(a + b, block_id=4)], a + b),
        "independent_blocks": ([H.liblz4_frame(b, block_id=4, linked=0)], b),
        "with_content_size": ([H.liblz4_frame(a, content_size=1)], a),
        "no_checksum_hc": ([H.liblz4_frame(a, checksum=0, level=9)], a),
        "block_checksums": ([H.liblz4_frame(a, block_checksum=1)], a),
        "three_frames": ([H.liblz4_frame(a), H.liblz4_frame(b, block_id=5), H.liblz4_frame(a[:10])], a + b + a[:10]),
        "skippable_between": ([H.liblz4_frame(a), b"\x5A\x2A\x4D\x18" + (5).to_bytes(4, "little") + b"hello",
                               H.liblz4_frame(b)], a + b),
        "empty_frame": ([H.liblz4_frame(b""), H.liblz4_frame(a)], a),
        "frame_then_records": ([H.oracle_compress(a, CHUNK)[12:]], a),
    }


@pytest.mark.skipif(H.liblz4_frame(b"x") is None, reason="liblz4 not on this box")
@pytest.mark.parametrize("name", sorted(_plain_lz4()) if H.liblz4_frame(b"x") is not None else [])
def test_lz4mt_plain_lz4_streams(lib, name):
    """.lz4 files without lz4-mt records (lib/lz4-mt_decompress.c:391-483); with 16 KiB batches every
    case takes several rounds of the incremental reader."""
    frames, plain = _plain_lz4()[name]
    st = b"".join(frames)
    rv, out, io, stats = H.lz4mt_decompress_via(lib, st, threads=4)
    assert rv == 0 and out == plain
    assert stats == (0, len(st), len(plain))
    if H.have_ref():
        rv_r, out_r, _, _ = H.lz4mt_decompress_via(H.ref(), st, threads=4)
        assert rv_r == 0 and out_r == out
    for bad in (st[:len(st) // 2], st[:-2]):
        if name in ("no_checksum_hc",) and bad == st[:-2]:
            continue                            # the end mark is all that is cut: still an error below
        rv, _, _, _ = H.lz4mt_decompress_via(lib, bad)
        assert lib.LZ4MT_isError(rv)


def test_lz4mt_callback_threads(lib):
    data = cases.text(5 * CHUNK + 99, 9)
    rv, stream, _, _ = H.lz4mt_compress_via(lib, data, CHUNK, threads=2)
    assert rv == 0
    me = threading.get_ident()
    rv, out, io, _ = H.lz4mt_decompress_via(lib, stream, threads=1)
    assert rv == 0 and out == data
    assert io.read_threads == {me} and io.write_threads == {me}
    rv, out, io, _ = H.lz4mt_decompress_via(lib, stream, threads=4)
    assert rv == 0 and out == data
    assert len(io.write_threads) == 1 and me not in io.write_threads


@pytest.mark.parametrize("devices,slots", [("all", None), ("1,0", "5"), ("0", "2"), (None, "3")])
def test_slots_dealt_over_devices(lib, monkeypatch, devices, slots):
    """GPUMT_DEVICES deals the batch slots out round-robin over the devices of a context (mt_host.h);
    the emulated boundary reports two devices.  Same bytes for every setting, and with two devices
    both of them have launched kernels."""
    if devices is None:
        monkeypatch.delenv("GPUMT_DEVICES", raising=False)
    else:
        monkeypatch.setenv("GPUMT_DEVICES", devices)
    if slots is None:
        monkeypatch.delenv("GPUMT_SLOTS", raising=False)
    else:
        monkeypatch.setenv("GPUMT_SLOTS", slots)
    data = _mixed(44000, 12)
    before = [lib.emu_gpumt_launches(d) for d in (0, 1)]
    rv, stream, _, _ = H.lz4mt_compress_via(lib, data, CHUNK, threads=2)
    assert rv == 0 and stream == H.oracle_compress(data, CHUNK)
    rv, out, _, _ = H.lz4mt_decompress_via(lib, stream, threads=2)
    assert rv == 0 and out == data
    used = [lib.emu_gpumt_launches(d) - before[d] for d in (0, 1)]
    if devices in ("all", "1,0"):
        assert used[0] > 0 and used[1] > 0
    else:
        assert used[0] > 0 and used[1] == 0


def test_bad_device_list_fails_cleanly(lib, monkeypatch):
    monkeypatch.setenv("GPUMT_DEVICES", "0,7")
    assert not lib.LZ4MT_createCCtx(2, 1, CHUNK)
    assert not lib.ZSTDCB_createDCtx(2, 0)
    monkeypatch.setenv("GPUMT_DEVICES", "zero")
    assert not lib.BROTLIMT_createCCtx(2, 3, CHUNK)


# ----------------------------------------------------------------------------------------- zstd-mt
ZCHUNK = 8192


@pytest.mark.parametrize("name,thunk", [("empty", lambda: b""), ("hello", lambda: b"hello world, hello!"),
                                        ("ragged", lambda: _mixed(30000, 21)),
                                        ("exact", lambda: cases.text(4 * ZCHUNK, 22))])
def test_zstdmt_compress_is_decompress_identical(lib, name, thunk):
    data = thunk()
    rv, st, io, stats = H.zstdmt_compress_via(lib, data, ZCHUNK, threads=3, level=1)
    assert rv == 0
    frames = max(1, -(-len(data) // ZCHUNK))
    assert stats == (frames, len(data), len(st))
    assert all(want == ZCHUNK for want, _ in io.reads) and len(io.writes) == frames
    at = 0
    for w in io.writes:                     # record framing (lib/zstd-mt_compress.c:296-302)
        assert struct.unpack_from("<III", st, at) == (0x184D2A50, 4, w - 12)
        assert st[at + 12:at + 16] == bytes([0x28, 0xB5, 0x2F, 0xFD])
        at += w
    assert at == len(st)
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    rv, back, _, dstats = H.zstdmt_decompress_via(lib, st, threads=3)
    assert rv == 0 and back == data and dstats == (frames, len(st), len(data))
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st, threads=2)
        assert rv == 0 and back == data


def test_zstdmt_level_reaches_the_encoder(lib):
    """ZSTDCB_createCCtx(level) is not just a chunk size: levels 1-2 / 3-9 / 10-22 select the encoder's tiers
    (the reference hands level to ZSTD_compress, lib/zstd-mt_compress.c:285)"""
    data = cases.text(300000, 23)
    sizes = {}
    for level in (1, 3, 19):
        rv, st, _, _ = H.zstdmt_compress_via(lib, data, 131072, threads=2, level=level)
        assert rv == 0 and H.oracle_zstdmt_decompress(st, len(data) + 64) == data
        sizes[level] = len(st)
    assert sizes[1] > sizes[3] > sizes[19]


def _strip_eof(reads):
    r = list(reads)
    while r and r[-1][1] == 0:
        r.pop()
    return r


@pytest.mark.skipif(not H.have_zref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("level", [1, 7])
def test_zstdmt_decompress_reference_streams(lib, level):
    data = _mixed(40000, 23)
    rv, st, io_c, _ = H.zstdmt_compress_via(H.zref(), data, ZCHUNK, threads=1, level=level)
    assert rv == 0
    rv_r, d_r, io_r, st_r = H.zstdmt_decompress_via(H.zref(), st, threads=2)
    rv_o, d_o, io_o, st_o = H.zstdmt_decompress_via(lib, st, threads=2)
    assert rv_r == 0 and d_r == data
    assert (rv_o, d_o, st_o) == (rv_r, d_r, st_r)
    assert _strip_eof(io_o.reads) == _strip_eof(io_r.reads) and io_o.writes == io_r.writes
    # and the compress side reads and writes like the reference (T = 1)
    rv, st2, io_o, _ = H.zstdmt_compress_via(lib, data, ZCHUNK, threads=1, level=level)
    assert rv == 0 and io_o.reads == io_c.reads and len(io_o.writes) == len(io_c.writes)
    # old "zstdmt style" prefix: a 9-byte empty frame in front (lib/zstd-mt_decompress.c:225-249)
    pre = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00]) + st
    rv_r, d_r, io_r, st_r = H.zstdmt_decompress_via(H.zref(), pre, threads=2)
    rv_o, d_o, io_o, st_o = H.zstdmt_decompress_via(lib, pre, threads=2)
    assert (rv_o, d_o, st_o) == (rv_r, d_r, st_r)
    assert _strip_eof(io_o.reads) == _strip_eof(io_r.reads) and io_o.writes == io_r.writes


def test_zstdmt_errors(lib):
    data = cases.text(40000, 24)
    rv, st, _, _ = H.zstdmt_compress_via(lib, data, ZCHUNK, threads=2)
    assert rv == 0
    rv, _, _, _ = H.zstdmt_decompress_via(lib, b"\x00" * 64)
    assert rv == ERR(Z_DATA)
    rv, _, _, _ = H.zstdmt_decompress_via(lib, st[:-7])
    assert rv == ERR(Z_DATA)
    c0 = struct.unpack_from("<I", st, 8)[0]
    bad = bytearray(st)
    bad[12 + c0] ^= 0xFF
    rv, _, _, _ = H.zstdmt_decompress_via(lib, bytes(bad))
    assert rv == ERR(Z_DATA)
    bad = bytearray(st)
    bad[16] ^= 0x08                          # frame header descriptor of the first record
    rv, _, _, _ = H.zstdmt_decompress_via(lib, bytes(bad))
    assert lib.ZSTDCB_isError(rv)
    io = H.MemIO(st, fail_read_at=2, read_rv=-2)
    ctx = lib.ZSTDCB_createDCtx(2, 0)
    assert lib.ZSTDCB_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(Z_CANCEL)
    lib.ZSTDCB_freeDCtx(ctx)
    io = H.MemIO(st, fail_write_at=3, write_rv=-1)
    ctx = lib.ZSTDCB_createDCtx(2, 0)
    assert lib.ZSTDCB_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(Z_READ)  # sic: mt_error
    lib.ZSTDCB_freeDCtx(ctx)
    io = H.MemIO(data, fail_read_at=3, read_rv=-3)
    ctx = lib.ZSTDCB_createCCtx(2, 1, ZCHUNK)
    assert lib.ZSTDCB_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(Z_MEM)
    io = H.MemIO(data)                       # counters restart on the next call (zstd-mt_compress.c:337-341)
    assert lib.ZSTDCB_compressCCtx(ctx, C.byref(io.rdwr)) == 0
    assert lib.ZSTDCB_GetInsizeCCtx(ctx) == len(data) and lib.ZSTDCB_GetFramesCCtx(ctx) == 5
    lib.ZSTDCB_freeCCtx(ctx)
    assert not lib.ZSTDCB_createCCtx(0, 1, 0) and not lib.ZSTDCB_createCCtx(1, 23, 0)
    assert lib.ZSTDCB_getErrorString(ERR(Z_DATA)) == b"Malformed input"


def _plain_zst():
    a = cases.text(30000, 81)
    b = cases.rnd(3000, 82) + bytes(20000) + cases.text(30000, 83)
    return {
        "one_frame": ([H.libzstd_frame(a, 3)], a),
        "no_content_size": ([H.libzstd_frame(a, 1, content_size=0)], a),
        "checksum_no_size": ([H.libzstd_frame(b, 5, checksum=1, content_size=0)], b),
        "three_frames": ([H.libzstd_frame(a, 1), H.libzstd_frame(b, 9, content_size=0), H.libzstd_frame(a[:1000], 19)],
                         a + b + a[:1000]),
        "skippable_between": ([H.libzstd_frame(a, 2), b"\x5A\x2A\x4D\x18" + (7).to_bytes(4, "little") + b"comment",
                               H.libzstd_frame(b, 2)], a + b),
        "tiny": ([H.libzstd_frame(b"abc", 1)], b"abc"),
        "empty_with_size": ([H.libzstd_frame(b"", 1), H.libzstd_frame(a, 1)], a),
    }


@pytest.mark.skipif(H.libzstd_frame(b"x") is None, reason="libzstd not on this box")
@pytest.mark.parametrize("name", sorted(_plain_zst()) if H.libzstd_frame(b"x") is not None else [])
def test_zstdmt_plain_zst_streams(lib, name):
    """.zst files without zstd-mt records (lib/zstd-mt_decompress.c:552-687), several rounds each."""
    frames, plain = _plain_zst()[name]
    st = b"".join(frames)
    rv, out, io, stats = H.zstdmt_decompress_via(lib, st, threads=4)
    assert rv == 0 and out == plain
    assert stats == (0, len(st), len(plain))
    assert max(io.writes, default=0) <= 131072
    if H.have_zref():
        rv_r, out_r, _, stats_r = H.zstdmt_decompress_via(H.zref(), st, threads=4)
        assert rv_r == 0 and out_r == out and stats_r == stats
    if len(st) > 40:
        rv, _, _, _ = H.zstdmt_decompress_via(lib, st[:len(st) // 2])
        assert lib.ZSTDCB_isError(rv)


# --------------------------------------------------------------------------------------- brotli-mt
BCHUNK = 8192


@pytest.mark.parametrize("name,thunk", [("empty", lambda: b""), ("hello", lambda: b"hello world, hello!"),
                                        ("ragged", lambda: _mixed(30000, 31)),
                                        ("exact", lambda: cases.text(4 * BCHUNK, 32))])
def test_brotlimt_compress_is_decompress_identical(lib, name, thunk):
    data = thunk()
    rv, st, io, stats = H.brotlimt_compress_via(lib, data, BCHUNK, threads=3, level=3)
    assert rv == 0
    frames = max(1, -(-len(data) // BCHUNK)) if data else stats[0]
    assert stats == (frames, len(data), len(st))
    assert len(io.writes) == frames
    at = 0
    for w in io.writes:                      # 16-byte record headers (lib/brotli-mt_compress.c:285-304)
        magic, eight, csz, br, hint = struct.unpack_from("<IIIHH", st, at)
        assert (magic, eight, csz, br) == (0x184D2A50, 8, w - 16, 0x5242)
        at += w
    assert at == len(st)
    assert H.oracle_brotlimt_decompress(st, len(data) + 64) == data
    rv, back, _, dstats = H.brotlimt_decompress_via(lib, st, threads=3)
    assert rv == 0 and back == data and dstats == (frames, len(st), len(data))
    if H.have_bref():
        rv, back, _, _ = H.brotlimt_decompress_via(H.bref(), st, threads=2)
        assert rv == 0 and back == data


@pytest.mark.skipif(not H.have_bref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("level", [0, 5, 11])
def test_brotlimt_decompress_reference_streams(lib, level):
    """The reference's hint is inputsize >> 16 for a full chunk (lib/brotli-mt_compress.c:294-304), so
    its own streams only decode for chunks that are multiples of 64 KiB: 65536 here."""
    data = _mixed(150000, 33)
    rv, st, io_c, _ = H.brotlimt_compress_via(H.bref(), data, 65536, threads=1, level=level)
    assert rv == 0
    rv_r, d_r, io_r, st_r = H.brotlimt_decompress_via(H.bref(), st, threads=2)
    rv_o, d_o, io_o, st_o = H.brotlimt_decompress_via(lib, st, threads=2)
    assert rv_r == 0 and d_r == data
    assert (rv_o, d_o, st_o) == (rv_r, d_r, st_r)
    assert _strip_eof(io_o.reads) == _strip_eof(io_r.reads) and io_o.writes == io_r.writes
    rv, st2, io_o, _ = H.brotlimt_compress_via(lib, data, 65536, threads=1, level=level)
    assert rv == 0 and io_o.reads == io_c.reads and len(io_o.writes) == len(io_c.writes)


def test_brotlimt_errors(lib):
    data = cases.text(30000, 34)
    rv, st, _, _ = H.brotlimt_compress_via(lib, data, BCHUNK, threads=2)
    assert rv == 0
    rv, _, _, _ = H.brotlimt_decompress_via(lib, b"\x00" * 64)
    assert rv == ERR(E_DATA)
    rv, _, _, _ = H.brotlimt_decompress_via(lib, st[:-7])
    assert rv == ERR(E_DATA)
    bad = bytearray(st)
    bad[12] ^= 1                             # the "BR" mark of the first record
    rv, _, _, _ = H.brotlimt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_DATA)
    bad = bytearray(st)
    bad[14] = 0                              # output hint 0: the stream does not fit
    bad[15] = 0
    rv, _, _, _ = H.brotlimt_decompress_via(lib, bytes(bad))
    assert lib.BROTLIMT_isError(rv)
    io = H.MemIO(st, fail_read_at=2, read_rv=-2)
    ctx = lib.BROTLIMT_createDCtx(2, 0)
    assert lib.BROTLIMT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(E_CANCEL)
    lib.BROTLIMT_freeDCtx(ctx)
    io = H.MemIO(data, fail_write_at=3, write_rv=-3)
    ctx = lib.BROTLIMT_createCCtx(2, 3, BCHUNK)
    assert lib.BROTLIMT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_MEM)
    lib.BROTLIMT_freeCCtx(ctx)
    assert not lib.BROTLIMT_createCCtx(0, 1, 0) and not lib.BROTLIMT_createCCtx(1, 12, 0)


# --------------------------------------------------------------------------------------- snappy-mt
SCHUNK = 4096


@pytest.mark.parametrize("name,chunk,thunk", [("empty", SCHUNK, lambda: b""), ("hello", SCHUNK, lambda: b"hello world, hello!"),
                                              ("ragged", SCHUNK, lambda: _mixed(30000, 41)),
                                              ("exact", SCHUNK, lambda: cases.text(8 * SCHUNK, 42)),
                                              ("default_64k", 0, lambda: cases.text(150000, 43))])
def test_snappymt_compress_is_decompress_identical(lib, name, chunk, thunk):
    data = thunk()
    rv, st, io, stats = H.snappymt_compress_via(lib, data, chunk, threads=3, level=0)
    assert rv == 0
    eff = chunk or 65536                     # SNAPPY_IN_ALLOC_SIZE, lib/snappy-mt_compress.c:12,102
    frames = max(1, -(-len(data) // eff))
    assert stats == (frames, len(data), len(st))
    assert all(want == eff for want, _ in io.reads) and len(io.writes) == frames
    at = 0
    for i, w in enumerate(io.writes):        # 16-byte record headers (lib/snappy-mt_compress.c:280-300)
        magic, eight, csz, sp, hint = struct.unpack_from("<IIIHH", st, at)
        n = min(eff, len(data) - i * eff)
        assert (magic, eight, csz, sp) == (0x184D2A50, 8, w - 16, 0x5053)
        assert hint == ((n >> 16) + 1 if n < eff else eff >> 16)
        at += w
    assert at == len(st)
    assert H.oracle_snappymt_decompress(st, len(data) + 64) == data
    rv, back, io, dstats = H.snappymt_decompress_via(lib, st, threads=3)
    assert rv == 0 and back == data and dstats == (frames, len(st), len(data))
    # reads of pt_read: 4-byte sniff, 12 more for the first header, payload, then 16 + payload, EOF
    wants = [w for w, _ in io.reads]
    assert wants[0] == 4 and wants[1] == 12 and wants[-1] == 16 and io.reads[-1][1] == 0
    assert len(io.writes) == frames
    # the same stream whatever the level argument (unused, lib/snappy-mt_compress.c:80,96)
    rv, st9, _, _ = H.snappymt_compress_via(lib, data, chunk, threads=1, level=9)
    assert rv == 0 and st9 == st


@pytest.mark.skipif(not H.have_libsnappy(), reason="libsnappy not on this box")
@pytest.mark.parametrize("threads", [1, 4])
def test_snappymt_decompress_foreign_streams(lib, threads):
    """Payloads written by libsnappy 1.1.8, framed as the reference does; a zero hint changes nothing
    (the output is sized from the stream's preamble, lib/snappy-mt_decompress.c:234-238,262-267)."""
    data = _mixed(40000, 44)
    st = H.snappymt_stream(data, SCHUNK)
    rv, out, io, stats = H.snappymt_decompress_via(lib, st, threads=threads)
    assert rv == 0 and out == data and stats == (10, len(st), len(data))
    if threads == 1:
        me = threading.get_ident()
        assert io.read_threads == {me} and io.write_threads == {me}
    nohint = bytearray(st)
    at = 0
    while at < len(nohint):
        nohint[at + 14:at + 16] = b"\0\0"
        at += 16 + struct.unpack_from("<I", nohint, at + 8)[0]
    rv, out, _, _ = H.snappymt_decompress_via(lib, bytes(nohint), threads=threads)
    assert rv == 0 and out == data


def test_snappymt_batched_decoder(lib, monkeypatch):
    """The second decoder of snappy.hip (64 elements per batch) behind the same API."""
    monkeypatch.setenv("EMU_SNAPPY_DEC", "1")
    data = _mixed(60000, 46) + bytes(9000) + b"abc" * 3000
    rv, st, _, _ = H.snappymt_compress_via(lib, data, 16384, threads=2)
    assert rv == 0
    rv, out, _, stats = H.snappymt_decompress_via(lib, st, threads=3)
    assert rv == 0 and out == data and stats[0] == -(-len(data) // 16384)
    bad = bytearray(st)
    bad[40] ^= 0x10
    want = H.oracle_snappymt_decompress(bytes(bad), len(data) + 64)
    rv, out, _, _ = H.snappymt_decompress_via(lib, bytes(bad), threads=3)
    assert (rv == 0 and out == want) if want is not None else rv == ERR(E_FD)


def test_snappymt_errors(lib):
    data = cases.text(40000, 45)
    rv, st, _, _ = H.snappymt_compress_via(lib, data, SCHUNK, threads=2)
    assert rv == 0
    rv, _, _, _ = H.snappymt_decompress_via(lib, b"\x00" * 64)
    assert rv == ERR(E_DATA)
    rv, _, _, _ = H.snappymt_decompress_via(lib, st[:-7])
    assert rv == ERR(E_DATA)                 # "needed more bytes"
    bad = bytearray(st)
    bad[12] ^= 1                             # the "SP" mark of the first record
    rv, _, _, _ = H.snappymt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_DATA)
    c0 = struct.unpack_from("<I", st, 8)[0]
    bad = bytearray(st)
    bad[16 + c0] ^= 1                        # skippable magic of the second record
    rv, _, _, _ = H.snappymt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_DATA)
    bad = bytearray(st)
    bad[16] = 0xFF                           # preamble of the first stream: claims far more than it holds
    bad[17] = 0xFF
    rv, _, _, _ = H.snappymt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_FD)                   # frame_decompress, as a failing snappy_uncompress (:350-357)
    bad = bytearray(st)
    bad[16 + c0 // 2] ^= 0x5A                # inside the first stream
    rv, out, _, _ = H.snappymt_decompress_via(lib, bytes(bad))
    want = H.oracle_snappymt_decompress(bytes(bad), len(data) + 64)
    assert (rv == 0 and out == want) if want is not None else rv == ERR(E_FD)
    io = H.MemIO(st, fail_read_at=2, read_rv=-2)
    ctx = lib.SNAPPYMT_createDCtx(2, 0)
    assert lib.SNAPPYMT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(E_CANCEL)
    lib.SNAPPYMT_freeDCtx(ctx)
    io = H.MemIO(data, fail_write_at=3, write_rv=-3)
    ctx = lib.SNAPPYMT_createCCtx(2, 0, SCHUNK)
    assert lib.SNAPPYMT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_MEM)
    lib.SNAPPYMT_freeCCtx(ctx)
    assert not lib.SNAPPYMT_createCCtx(0, 1, 0) and not lib.SNAPPYMT_createCCtx(129, 1, 0)
    ctx = lib.SNAPPYMT_createCCtx(1, 77, 0)  # any level
    assert ctx
    lib.SNAPPYMT_freeCCtx(ctx)
    assert not lib.SNAPPYMT_createDCtx(0, 0)
    io = H.MemIO(b"x")
    assert lib.SNAPPYMT_compressCCtx(None, C.byref(io.rdwr)) == ERR(E_PARAM)
    assert lib.SNAPPYMT_decompressDCtx(None, C.byref(io.rdwr)) == ERR(E_PARAM)
    assert lib.SNAPPYMT_getErrorString(ERR(E_DATA)) == b"Malformed input"
    assert lib.SNAPPYMT_getErrorString(ERR(E_CANCEL)) == b"Unspecified snappy error code"
    assert lib.BROTLIMT_getErrorString(ERR(E_CANCEL)) == b"Unspecified brotli error code"
