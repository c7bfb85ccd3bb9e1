"""The drop-in boundary: LZ4MT_* of libzstdmt_amd.so (include/lz4-mt.h) driven through the same
callback protocol as the reference (lib/lz4-mt.h), checked against the golden vectors, the oracle
and -- where oracle/_ref is present -- the reference library itself, call for call."""
import ctypes as C
import json
import os

import pytest

import helpers as H
from cases import CASES, rnd, text

pytestmark = pytest.mark.gpu

with open(os.path.join(H.GOLDEN_DIR, "manifest.json")) as _f:
    MAN = json.load(_f)["cases"]

ERR = lambda e: C.c_size_t(-e).value  # noqa: E731  (size_t)-enum
E_MEM, E_READ, E_WRITE, E_DATA, E_FC, E_FD, E_PARAM, E_LIB, E_CANCEL = range(1, 10)


@pytest.fixture(scope="module")
def lib():
    from zstdmt_amd._native import lib_path
    return H.bind_lz4mt(C.CDLL(lib_path()))


@pytest.mark.parametrize("name", sorted(CASES))
def test_compress_matches_reference_golden(lib, name):
    chunk, thunk = CASES[name]
    data = thunk()
    rv, stream, io, stats = H.lz4mt_compress_via(lib, data, chunk, threads=4, level=1)
    e = MAN[name]
    assert rv == 0
    assert len(stream) == e["out_len"] and H.sha256(stream) == e["out_sha256"]
    # statistics as the reference reports them (frames written, bytes read, bytes written)
    assert stats == (e["frames"], e["insize"], e["outsize"])
    # one fn_read of exactly `inputsize` per chunk (+ the terminating empty read), one fn_write
    # per record, in order
    assert all(want == chunk for want, _ in io.reads)
    assert len(io.writes) == e["frames"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_decompress_golden(lib, name):
    chunk, thunk = CASES[name]
    data = thunk()
    stream = H.oracle_compress(data, chunk)
    rv, out, io, stats = H.lz4mt_decompress_via(lib, stream, threads=2)
    e = MAN[name]
    assert rv == 0 and out == data
    assert stats == (e["frames"], e["d_insize"], e["d_outsize"])
    # read pattern of pt_read: 4-byte sniff, 8-byte rest of the first header, payload, then
    # 12-byte headers + payloads, then the empty read that signals EOF
    wants = [w for w, _ in io.reads]
    assert wants[0] == 4 and wants[1] == 8
    assert wants[-1] == 12 and io.reads[-1][1] == 0


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["empty", "hello_5", "text_3x128k_p100", "zeros_262149",
                                  "text_300k_chunk64k", "rnd_512k_chunk256k"])
def test_callback_trace_equals_reference(lib, name):
    """Same requests, same sizes, same order as the reference library (T=1: deterministic)."""
    chunk, thunk = CASES[name]
    data = thunk()
    ref = H.ref()
    rv_r, s_r, io_r, st_r = H.lz4mt_compress_via(ref, data, chunk, threads=1)
    rv_o, s_o, io_o, st_o = H.lz4mt_compress_via(lib, data, chunk, threads=1)
    assert (rv_o, s_o, st_o) == (rv_r, s_r, st_r)
    assert io_o.reads == io_r.reads and io_o.writes == io_r.writes
    rv_r, d_r, io_r, st_r = H.lz4mt_decompress_via(ref, s_r, threads=1)
    rv_o, d_o, io_o, st_o = H.lz4mt_decompress_via(lib, s_r, threads=1)
    assert (rv_o, d_o, st_o) == (rv_r, d_r, st_r)
    assert io_o.reads == io_r.reads and io_o.writes == io_r.writes


@pytest.mark.parametrize("level", [3, 4, 6, 8, 9, 10, 12])
def test_hc_levels_match_oracle_and_reference(lib, level):
    """Levels 3..12 (LZ4HC hash chain, the CLI default is 3): same bytes as the oracle, and as the
    reference library where it is built; the stream decodes back."""
    data = text(700000) + rnd(70000, 3) + bytes(200000) + text(300000)[::-1]
    chunk = 262144
    rv, stream, io, stats = H.lz4mt_compress_via(lib, data, chunk, threads=3, level=level)
    assert rv == 0
    assert stream == H.oracle_compress_level(data, chunk, level)
    if H.have_ref():
        rv_r, s_r, _, st_r = H.lz4mt_compress_via(H.ref(), data, chunk, threads=1, level=level)
        assert rv_r == 0 and s_r == stream and st_r == stats
    rv, out, _, _ = H.lz4mt_decompress_via(lib, stream, threads=2)
    assert rv == 0 and out == data


def test_large_pipeline_many_batches(lib):
    """> 2 device batches each way (64 MiB per batch): exercises the double-buffered pipeline."""
    data = text(200 << 20)
    rv, stream, io, stats = H.lz4mt_compress_via(lib, data, 131072, threads=8)
    assert rv == 0 and stats[0] == 1600
    for i in (0, 511, 512, 1599):   # batch boundaries included
        off = sum(io.writes[:i])
        assert stream[off:off + io.writes[i]] == H.oracle_compress(data[i * 131072:(i + 1) * 131072], 131072)
    rv, out, _, dstats = H.lz4mt_decompress_via(lib, stream, threads=8)
    assert rv == 0 and out == data and dstats[0] == 1600


def test_config1_default_chunk(lib):
    """BASELINE config 1 through the API: inputsize 0 -> 4 MiB chunks, 64 MiB of PRNG bytes."""
    data = rnd(64 << 20, 7)
    rv, stream, io, stats = H.lz4mt_compress_via(lib, data, 0, threads=4)
    assert rv == 0 and stats == (16, 64 << 20, (64 << 20) + 4656)
    assert all(w == 4 << 20 for w, _ in io.reads)
    rv, out, _, _ = H.lz4mt_decompress_via(lib, stream, threads=4)
    assert rv == 0 and out == data


def test_create_argument_checks(lib):
    assert not lib.LZ4MT_createCCtx(0, 1, 0)
    assert not lib.LZ4MT_createCCtx(129, 1, 0)
    assert not lib.LZ4MT_createCCtx(1, 0, 0)
    assert not lib.LZ4MT_createCCtx(1, 13, 0)
    assert not lib.LZ4MT_createDCtx(0, 0)
    assert not lib.LZ4MT_createDCtx(129, 0)
    io = H.MemIO(b"x")
    assert lib.LZ4MT_compressCCtx(None, C.byref(io.rdwr)) == ERR(E_PARAM)
    assert lib.LZ4MT_decompressDCtx(None, C.byref(io.rdwr)) == ERR(E_PARAM)
    # every level the reference accepts is served (12 = LZ4HC optimal parser)
    rv, s12, _, _ = H.lz4mt_compress_via(lib, b"abc" * 50, 131072, threads=1, level=12)
    assert rv == 0 and s12 == H.oracle_compress_level(b"abc" * 50, 131072, 12)
    assert lib.LZ4MT_getErrorString(ERR(E_PARAM)) == b"Compression parameter is out of bound"
    assert not lib.LZ4MT_isError(0)


@pytest.mark.parametrize("rv_cb,code", [(-1, E_READ), (-2, E_CANCEL), (-3, E_MEM), (-7, E_READ)])
def test_read_failures(lib, rv_cb, code):
    data = text(400000)
    io = H.MemIO(data, fail_read_at=2, read_rv=rv_cb)
    ctx = lib.LZ4MT_createCCtx(2, 1, 131072)
    assert lib.LZ4MT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(code)
    lib.LZ4MT_freeCCtx(ctx)
    stream = H.oracle_compress(data, 131072)
    io = H.MemIO(stream, fail_read_at=3, read_rv=rv_cb)
    ctx = lib.LZ4MT_createDCtx(2, 0)
    assert lib.LZ4MT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(code)
    lib.LZ4MT_freeDCtx(ctx)


def test_write_failure_maps_like_reference(lib):
    """The reference routes write failures through mt_error(): -1 surfaces as read_fail."""
    data = text(400000)
    io = H.MemIO(data, fail_write_at=1, write_rv=-1)
    ctx = lib.LZ4MT_createCCtx(2, 1, 131072)
    assert lib.LZ4MT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_READ)
    lib.LZ4MT_freeCCtx(ctx)


@pytest.mark.parametrize("mutate,code", [("badmagic", E_DATA), ("notlz4", E_DATA), ("skiplen", E_DATA),
                                         ("truncated", E_DATA), ("frame", E_LIB), ("checksum", E_LIB),
                                         ("secondmagic", E_DATA)])
def test_decompress_errors(lib, mutate, code):
    data = text(300000)
    s = bytearray(H.oracle_compress(data, 131072))
    if mutate == "badmagic":
        s[0] ^= 1                      # neither skippable nor LZ4F magic
    elif mutate == "notlz4":
        s[:4] = b"\x28\xb5\x2f\xfd"    # a zstd magic
    elif mutate == "skiplen":
        s[4] = 8
    elif mutate == "truncated":
        s = s[:-10]
    elif mutate == "frame":
        s[12] ^= 1                     # LZ4F magic inside the first record
    elif mutate == "checksum":
        s[-1] ^= 0x40
    elif mutate == "secondmagic":
        import struct
        c0 = struct.unpack_from("<I", s, 8)[0]
        s[12 + c0] ^= 1                # skippable magic of the second record
    rv, out, _, _ = H.lz4mt_decompress_via(lib, bytes(s), threads=2)
    assert rv == ERR(code), (rv, lib.LZ4MT_getErrorString(rv))
    assert lib.LZ4MT_isError(rv)


# ---- plain .lz4 streams (the reference's single-threaded path, lib/lz4-mt_decompress.c:391-483) ----
def _plain_lz4_cases():
    a = text(300000, 81)
    b = rnd(50000, 82) + bytes(200000) + text(700000, 83)
    big = text(9 << 20, 84)
    return {
        "tool_defaults": ([H.liblz4_frame(big)], big),                       # 4 MiB linked blocks, no content size
        "independent_256k": ([H.liblz4_frame(b, block_id=5, linked=0)], b),
        "with_content_size": ([H.liblz4_frame(a, content_size=1)], a),
        "no_checksum_hc": ([H.liblz4_frame(a, checksum=0, level=9)], a),
        "three_frames": ([H.liblz4_frame(a), H.liblz4_frame(b, block_id=6), H.liblz4_frame(a[:10])], a + b + a[:10]),
        "skippable_between": ([H.liblz4_frame(a), b"\x5A\x2A\x4D\x18" + (5).to_bytes(4, "little") + b"hello",
                               H.liblz4_frame(b)], a + b),
        "empty_frame": ([H.liblz4_frame(b""), H.liblz4_frame(a)], a),
        # a bare frame followed by lz4-mt records: the record headers are skippable frames
        "frame_then_records": ([H.oracle_compress(a, 131072)[12:]], a),
    }


@pytest.mark.skipif(H.liblz4_frame(b"x") is None, reason="liblz4 not on this box")
@pytest.mark.parametrize("name", ["tool_defaults", "independent_256k", "with_content_size", "no_checksum_hc",
                                  "three_frames", "skippable_between", "empty_frame", "frame_then_records"])
def test_decompress_plain_lz4_streams(lib, name):
    frames, plain = _plain_lz4_cases()[name]
    st = b"".join(frames)
    rv, out, io, stats = H.lz4mt_decompress_via(lib, st, threads=4)
    assert rv == 0 and out == plain
    assert stats == (0, len(st), len(plain))            # st_decompress counts no frames
    if H.have_ref():
        rv_r, out_r, _, stats_r = H.lz4mt_decompress_via(H.ref(), st, threads=4)
        assert rv_r == 0 and out_r == out
        # the reference counts every input byte twice on this path (once when read, :467, once when
        # consumed, :441); reported here once -- INTEGRATION.md
        assert stats_r == (0, 2 * len(st), len(plain))


@pytest.mark.skipif(H.liblz4_frame(b"x") is None, reason="liblz4 not on this box")
def test_plain_lz4_errors(lib):
    a = text(200000, 91)
    f = H.liblz4_frame(a)
    for bad in (f[:len(f) // 2], f[:-2]):               # truncated inside a block / inside the checksum
        rv, _, _, _ = H.lz4mt_decompress_via(lib, bad)
        assert rv == ERR(E_LIB)
    dmg = bytearray(f)
    dmg[-1] ^= 0x20                                     # content checksum
    rv, _, _, _ = H.lz4mt_decompress_via(lib, bytes(dmg))
    assert rv == ERR(E_LIB)
    dmg = bytearray(f)
    dmg[len(f) // 2] ^= 0xFF
    dmg[len(f) // 2 + 1] ^= 0xFF
    rv, out, _, _ = H.lz4mt_decompress_via(lib, bytes(dmg))
    assert rv == ERR(E_LIB) or out != a                # damaged tokens: rejected, or caught by the checksum
    bc = H.liblz4_frame(a, block_checksum=1)            # block checksums are verified on the device
    rv, out, _, _ = H.lz4mt_decompress_via(lib, bc)
    assert rv == 0 and out == a
    dmg = bytearray(bc)
    dmg[len(bc) // 2] ^= 0x01
    rv, _, _, _ = H.lz4mt_decompress_via(lib, bytes(dmg))
    assert rv == ERR(E_LIB)


def test_callback_threads(lib):
    """threads == 1: every decompress callback runs on the calling thread (reference:
    lib/lz4-mt_decompress.c:528-534); otherwise one reader and one writer thread of the library."""
    import threading
    data = text(5 * 131072 + 99)
    rv, stream, _, _ = H.lz4mt_compress_via(lib, data, 131072, threads=2)
    assert not lib.LZ4MT_isError(rv)
    me = threading.get_ident()
    rv, out, io, _ = H.lz4mt_decompress_via(lib, stream, threads=1)
    assert not lib.LZ4MT_isError(rv) and out == data
    assert io.read_threads == {me} and io.write_threads == {me}
    rv, out, io, _ = H.lz4mt_decompress_via(lib, stream, threads=4)
    assert not lib.LZ4MT_isError(rv) and out == data
    assert len(io.write_threads) == 1 and me not in io.write_threads


def _cpulist(text_):
    cpus = set()
    for part in text_.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def test_reader_and_writer_threads_run_next_to_the_device(lib, monkeypatch):
    """mt_bind_near: the library's reader / writer threads (never the caller's) bind themselves to the CPUs of the host NUMA
    node the device hangs off -- where its pinned batch buffers live -- within the mask the process has; GPUMT_NUMA=0 leaves
    them alone.  The callbacks report the affinity mask they run under."""
    import zstdmt_amd as z
    eng = z.Engine(0)
    node = eng.L.gpumt_host_node(eng.h)
    me = frozenset(os.sched_getaffinity(0))
    data = text(5 * 131072 + 99)
    monkeypatch.setenv("GPUMT_NUMA", "0")
    rv, stream, io, _ = H.lz4mt_compress_via(lib, data, 131072, threads=2)
    assert not lib.LZ4MT_isError(rv) and io.read_cpus == {me} and io.write_cpus == {me}
    monkeypatch.delenv("GPUMT_NUMA")
    rv, stream2, io, _ = H.lz4mt_compress_via(lib, data, 131072, threads=2)
    assert not lib.LZ4MT_isError(rv) and stream2 == stream
    rv, out, iod, _ = H.lz4mt_decompress_via(lib, stream, threads=4)
    assert not lib.LZ4MT_isError(rv) and out == data
    path = "/sys/devices/system/node/node%d/cpulist" % node
    want = me
    if node >= 0 and os.path.exists(path):
        near = frozenset(_cpulist(open(path).read())) & me
        want = near if near else me
    for got in (io.read_cpus, io.write_cpus, iod.write_cpus):
        assert got == {want}, (node, sorted(want)[:4], [sorted(g)[:4] for g in got])
    # (the decompressor sniffs the stream's first four bytes on the caller's thread before its pipeline starts)
    assert iod.read_cpus - {me} == {want} - {me} and want in iod.read_cpus | {me}
    assert frozenset(os.sched_getaffinity(0)) == me   # the caller's own thread is never touched
