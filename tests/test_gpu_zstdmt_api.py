"""The drop-in boundary for zstd-mt: ZSTDCB_* of libzstdmt_amd.so (include/zstd-mt.h) driven through
the reference's callback protocol (lib/zstd-mt.h), against streams the reference wrote and -- where
oracle/_ref travelled -- the reference library itself, call for call.  Bar: decompress-identical."""
import ctypes as C
import json
import os

import pytest

import helpers as H
from golden import cases

pytestmark = pytest.mark.gpu

ZDIR = os.path.join(H.GOLDEN_DIR, "zstd")
MAN = json.load(open(os.path.join(ZDIR, "manifest.json")))["cases"]
ERR = lambda e: C.c_size_t(-e).value  # noqa: E731  (size_t)-enum
E_MEM, E_INIT, E_READ, E_WRITE, E_DATA, E_FC, E_FD, E_PARAM, E_LIB, E_CANCEL = range(1, 11)


@pytest.fixture(scope="module")
def lib():
    from zstdmt_amd._native import lib_path
    return H.bind_lz4mt(C.CDLL(lib_path()), "ZSTDCB_")


def _stream(name):
    ent = MAN[name]
    if "out_file" in ent:
        return open(os.path.join(ZDIR, ent["out_file"]), "rb").read()
    if not H.have_zref():
        pytest.skip("stream not committed (size) and no reference build on this box")
    level, chunk, thunk = cases.ZCASES[name]
    rv, st, _, _ = H.zstdmt_compress_via(H.zref(), thunk(), chunk, threads=2, level=level)
    assert rv == 0
    return st


def _strip_eof(reads):
    r = list(reads)
    while r and r[-1][1] == 0:
        r.pop()
    return r


@pytest.mark.parametrize("name", sorted(MAN))
def test_decompress_reference_streams(lib, name):
    ent = MAN[name]
    st = _stream(name)
    rv, out, io, stats = H.zstdmt_decompress_via(lib, st, threads=4)
    assert rv == 0
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]
    # frames written, bytes read (headers included), bytes written -- as the reference counts them
    assert stats == (ent["frames"], ent["d_insize"], ent["d_outsize"])
    # one fn_write per frame, in order
    assert len(io.writes) == ent["frames"] and sum(io.writes) == ent["in_len"]
    # reads: 16-byte sniff, rest of the first record, then 12 + csize per record (pt_read)
    assert io.reads[0] == (16, 16)


@pytest.mark.skipif(not H.have_zref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["z_empty", "z_hello", "z_text_3x128k", "z_zeros_chunks", "z_mixed"])
def test_decompress_trace_equals_reference(lib, name):
    st = _stream(name)
    rv_r, d_r, io_r, st_r = H.zstdmt_decompress_via(H.zref(), st, threads=2)
    rv_o, d_o, io_o, st_o = H.zstdmt_decompress_via(lib, st, threads=2)
    assert (rv_o, d_o, st_o) == (rv_r, d_r, st_r)
    # every reference worker ends on its own zero-length read; the requests before that are equal
    assert _strip_eof(io_o.reads) == _strip_eof(io_r.reads)
    assert io_o.writes == io_r.writes


def test_decompress_errors(lib):
    st = _stream("z_text_3x128k")
    # not a zstd-mt stream
    rv, _, _, _ = H.zstdmt_decompress_via(lib, b"\x00" * 64)
    assert rv == ERR(E_DATA)
    # truncated inside a record: "needed more bytes" -> data_error (zstd-mt_decompress.c:346-347)
    rv, _, _, _ = H.zstdmt_decompress_via(lib, st[:-7])
    assert rv == ERR(E_DATA)
    # corrupt payload -> compression_library, device status kept in the global
    bad = bytearray(st)
    bad[len(st) // 2] ^= 0x40
    rv, _, _, _ = H.zstdmt_decompress_via(lib, bytes(bad))
    assert rv == 0 or rv == ERR(E_LIB)
    # second record header damaged
    import struct
    c0 = struct.unpack_from("<I", st, 8)[0]
    bad = bytearray(st)
    bad[12 + c0] ^= 0xFF
    rv, _, _, _ = H.zstdmt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_DATA)
    # callback failures map through mt_error
    io = H.MemIO(st, fail_read_at=2, read_rv=-2)
    ctx = lib.ZSTDCB_createDCtx(2, 0)
    assert lib.ZSTDCB_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(E_CANCEL)
    lib.ZSTDCB_freeDCtx(ctx)
    io = H.MemIO(st, fail_write_at=1, write_rv=-1)
    ctx = lib.ZSTDCB_createDCtx(2, 0)
    assert lib.ZSTDCB_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(E_READ)  # sic: mt_error
    lib.ZSTDCB_freeDCtx(ctx)
    assert not lib.ZSTDCB_createDCtx(0, 0) and not lib.ZSTDCB_createDCtx(129, 0)
    assert lib.ZSTDCB_isError(ERR(E_LIB)) and not lib.ZSTDCB_isError(0)
    assert lib.ZSTDCB_getErrorString(ERR(E_DATA)) == b"Malformed input"


@pytest.mark.skipif(not H.have_zref(), reason="oracle/_ref not built")
def test_decompress_many_batches(lib):
    """> 2 device batches (64 MiB of output per batch at first): the double-buffered pipeline."""
    data = cases.text(300 << 20, 77)
    rv, st, _, stats = H.zstdmt_compress_via(H.zref(), data, 0, threads=32, level=1)
    assert rv == 0 and stats[0] == 300
    rv, out, io, dstats = H.zstdmt_decompress_via(lib, st, threads=8)
    assert rv == 0 and out == data and dstats == (300, len(st), len(data))


# ------------------------------------------------------------------------------- compression
ZC_CASES = {
    "empty": (131072, lambda: b""),
    "hello": (131072, lambda: b"hello world, hello world, hello!"),
    "text_100": (131072, lambda: cases.text(100)),
    "text_300": (131072, lambda: cases.text(300, 2)),
    "text_70k": (131072, lambda: cases.text(70000, 3)),
    "text_3x128k": (131072, lambda: cases.text(3 * 131072 + 100, 11)),
    "text_1m_default": (0, lambda: cases.text(1048576 + 77, 3)),
    "text_5m_default": (0, lambda: cases.text(5 * 1048576 + 12345, 5)),
    "random_200k": (131072, lambda: cases.rnd(200000, 3)),
    "zeros_300k": (0, lambda: bytes(300000)),
    "period_300": (0, lambda: cases.rep(cases.rnd(300, 9), 200000)),
    "period_65537": (0, lambda: cases.rep(cases.rnd(65537, 4), 400000)),
    "two_symbols": (65536, lambda: bytes(65 + (b & 1) for b in cases.rnd(70000, 13))),
    "mixed": (262144, lambda: cases.text(50000, 4) + bytes(70000) + cases.rnd(3000, 5) + cases.text(200000, 6)),
    "allbytes": (0, lambda: bytes(range(256)) * 300 + cases.text(40000, 8)),
    # byte values above 128: Huffman trees described by FSE-coded weights
    "text_high_bytes": (0, lambda: bytes(b ^ 0x80 for b in cases.text(2500000, 8))),
    "all_256_values": (0, lambda: bytes((b * 7) & 255 for b in cases.text(1500000, 9))),
    "four_high_symbols": (131072, lambda: bytes(200 + (b & 3) for b in cases.rnd(300000, 5))),
}


@pytest.mark.parametrize("name", sorted(ZC_CASES))
def test_compress_is_decompress_identical(lib, name):
    """The bar for zstd: what ZSTDCB_compressCCtx writes must decode to the input
    with the oracle, with this library, and -- where present -- with the reference + libzstd."""
    chunk, thunk = ZC_CASES[name]
    data = thunk()
    rv, st, io, stats = H.zstdmt_compress_via(lib, data, chunk, threads=4, level=1)
    assert rv == 0
    eff = chunk or (1 << 20)
    frames = max(1, -(-len(data) // eff))
    assert stats == (frames, len(data), len(st))
    assert all(want == eff for want, _ in io.reads) and len(io.writes) == frames
    # record framing byte for byte (lib/zstd-mt_compress.c:296-302)
    import struct
    at = 0
    for w in io.writes:
        magic, four, csz = struct.unpack_from("<III", st, at)
        assert (magic, four, csz) == (0x184D2A50, 4, w - 12)
        assert st[at + 12:at + 16] == bytes([0x28, 0xB5, 0x2F, 0xFD])
        at += w
    assert at == len(st)
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    rv, back, _, dstats = H.zstdmt_decompress_via(lib, st, threads=4)
    assert rv == 0 and back == data and dstats == (frames, len(st), len(data))
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), st, threads=3)
        assert rv == 0 and back == data


def test_compress_is_deterministic_and_bounded(lib):
    data = cases.text(3 << 20, 19) + cases.rnd(1 << 20, 2)
    rv, a, _, _ = H.zstdmt_compress_via(lib, data, 1 << 20, threads=2, level=1)
    rv2, b, _, _ = H.zstdmt_compress_via(lib, data, 1 << 20, threads=9, level=2)
    assert rv == 0 and rv2 == 0 and a == b        # same encoder tier: independent of threads and of the level inside it
    assert len(a) < 0.62 * (3 << 20) + (1 << 20) + 200   # text shrinks, noise is stored raw
    # the level reaches the encoder (lib/zstd-mt_compress.c:285): tiers 3-9 and 10-22 compress better
    rv3, c, _, _ = H.zstdmt_compress_via(lib, data, 1 << 20, threads=4, level=7)
    rv4, d, _, _ = H.zstdmt_compress_via(lib, data, 1 << 20, threads=4, level=19)
    assert rv3 == 0 and rv4 == 0 and len(a) > len(c) > len(d)
    for st in (c, d):
        assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data


def test_compress_argument_and_callback_errors(lib):
    assert not lib.ZSTDCB_createCCtx(0, 1, 0) and not lib.ZSTDCB_createCCtx(129, 1, 0)
    assert not lib.ZSTDCB_createCCtx(1, 0, 0) and not lib.ZSTDCB_createCCtx(1, 23, 0)
    data = cases.text(400000, 1)
    io = H.MemIO(data, fail_read_at=1, read_rv=-3)
    ctx = lib.ZSTDCB_createCCtx(2, 1, 131072)
    assert lib.ZSTDCB_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_MEM)
    lib.ZSTDCB_freeCCtx(ctx)
    io = H.MemIO(data, fail_write_at=0, write_rv=-2)
    ctx = lib.ZSTDCB_createCCtx(2, 1, 131072)
    assert lib.ZSTDCB_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_CANCEL)
    # counters restart on the next call (zstd-mt_compress.c:337-341)
    io = H.MemIO(data)
    assert lib.ZSTDCB_compressCCtx(ctx, C.byref(io.rdwr)) == 0
    assert lib.ZSTDCB_GetInsizeCCtx(ctx) == len(data) and lib.ZSTDCB_GetFramesCCtx(ctx) == 4
    lib.ZSTDCB_freeCCtx(ctx)


def test_roundtrip_many_batches(lib):
    data = cases.text(300 << 20, 78)
    rv, st, io, stats = H.zstdmt_compress_via(lib, data, 0, threads=8, level=1)
    assert rv == 0 and stats[0] == 300
    rv, out, _, dstats = H.zstdmt_decompress_via(lib, st, threads=8)
    assert rv == 0 and out == data and dstats == (300, len(st), len(data))


@pytest.mark.skipif(not H.have_zref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["z_hello", "z_text_3x128k", "z_mixed"])
def test_decompress_old_zstdmt_prefix_layout(lib, name):
    """Old "zstdmt style" streams (lib/zstd-mt_decompress.c:225-249): a 9-byte empty zstd frame in
    front of ordinary records.  Same output, requests and counters as the reference."""
    st = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00]) + _stream(name)
    rv_r, d_r, io_r, st_r = H.zstdmt_decompress_via(H.zref(), st, threads=2)
    rv_o, d_o, io_o, st_o = H.zstdmt_decompress_via(lib, st, threads=2)
    assert rv_r == 0 and len(d_r) == MAN[name]["in_len"]
    assert (rv_o, d_o, st_o) == (rv_r, d_r, st_r)
    assert _strip_eof(io_o.reads) == _strip_eof(io_r.reads) and io_o.writes == io_r.writes


def test_large_default_chunks(lib):
    """Level 12 -> default chunk 1 << (windowLog[12] + 1) = 8 MiB (lib/zstd-mt_compress.c:116-127):
    64 blocks per frame, three frames."""
    data = cases.text(20 << 20, 44) + cases.rnd(1 << 20, 9)
    rv, st, io, stats = H.zstdmt_compress_via(lib, data, 0, threads=4, level=12)
    assert rv == 0 and stats[0] == 3 and all(want == 8 << 20 for want, _ in io.reads)
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    rv, out, _, _ = H.zstdmt_decompress_via(lib, st, threads=4)
    assert rv == 0 and out == data


# ---- plain .zst streams (the reference's single-threaded path, lib/zstd-mt_decompress.c:552-687) ----
def _plain_cases():
    a = cases.text(300000, 81)
    b = cases.rnd(50000, 82) + bytes(200000) + cases.text(700000, 83)
    big = cases.text(5 << 20, 84)
    return {
        # frames of the library flavours zstd-mt never writes: no content size, checksums, high levels
        "one_frame": [H.libzstd_frame(a, 3)],
        "no_content_size": [H.libzstd_frame(a, 1, content_size=0)],
        "checksum_no_size": [H.libzstd_frame(b, 5, checksum=1, content_size=0)],
        "three_frames": [H.libzstd_frame(a, 1), H.libzstd_frame(b, 9, content_size=0), H.libzstd_frame(a[:1000], 19)],
        "skippable_between": [H.libzstd_frame(a, 2), b"\x5A\x2A\x4D\x18" + (7).to_bytes(4, "little") + b"comment",
                              H.libzstd_frame(b, 2)],
        "large_window_many_blocks": [H.libzstd_frame(big, 6, content_size=0)],
        "tiny": [H.libzstd_frame(b"abc", 1)],            # shorter than the 16-byte sniff
        "empty_with_size": [H.libzstd_frame(b"", 1), H.libzstd_frame(a, 1)],
    }, {"one_frame": a, "no_content_size": a, "checksum_no_size": b, "three_frames": a + b + a[:1000],
        "skippable_between": a + b, "large_window_many_blocks": big, "tiny": b"abc", "empty_with_size": a}


@pytest.mark.skipif(H.libzstd_frame(b"x") is None, reason="libzstd not on this box")
@pytest.mark.parametrize("name", ["one_frame", "no_content_size", "checksum_no_size", "three_frames",
                                  "skippable_between", "large_window_many_blocks", "tiny", "empty_with_size"])
def test_decompress_plain_zst_streams(lib, name):
    streams, plain = _plain_cases()
    st = b"".join(streams[name])
    rv, out, io, stats = H.zstdmt_decompress_via(lib, st, threads=4)
    assert rv == 0 and out == plain[name]
    assert stats == (0, len(st), len(plain[name]))      # st_decompress counts no frames
    assert max(io.writes, default=0) <= 131072           # pieces of ZSTD_DStreamOutSize() at most
    if H.have_zref():
        rv_r, out_r, io_r, stats_r = H.zstdmt_decompress_via(H.zref(), st, threads=4)
        assert rv_r == 0 and out_r == out and stats_r == stats
        assert _strip_eof(io_r.reads)[:2] == _strip_eof(io.reads)[:2]   # sniff, then the rest of the first buffer


@pytest.mark.skipif(H.libzstd_frame(b"x") is None, reason="libzstd not on this box")
def test_mt_records_without_content_size(lib):
    """pzstd-style records whose frames were streamed (no Frame_Content_Size): the reference grows
    its output buffer (lib/zstd-mt_decompress.c:499-522); here the block headers bound the content
    and the decoder reports the size.  Mixed with sized frames and a checksummed one."""
    parts = [cases.text(300000, 61), cases.rnd(4000, 62) + bytes(90000), cases.text(1 << 20, 63), b"tail"]
    frames = [H.libzstd_frame(parts[0], 1, content_size=0), H.libzstd_frame(parts[1], 3, checksum=1, content_size=0),
              H.libzstd_frame(parts[2], 1), H.libzstd_frame(parts[3], 1, content_size=0)]
    st = b"".join(b"\x50\x2A\x4D\x18" + (4).to_bytes(4, "little") + len(f).to_bytes(4, "little") + f for f in frames)
    rv, out, io, stats = H.zstdmt_decompress_via(lib, st, threads=4)
    assert rv == 0 and out == b"".join(parts)
    assert stats == (4, len(st), len(out))
    assert io.writes == [len(p) for p in parts]          # one fn_write per frame, its decoded size
    if H.have_zref():
        rv_r, out_r, _, stats_r = H.zstdmt_decompress_via(H.zref(), st, threads=4)
        assert rv_r == 0 and out_r == out and stats_r == stats
    # damage inside the checksummed size-less frame is still an error (block syntax or XXH64)
    dmg = bytearray(st)
    dmg[12 + len(frames[0]) + 12 + len(frames[1]) // 2] ^= 0x10
    rv, _, _, _ = H.zstdmt_decompress_via(lib, bytes(dmg), threads=4)
    assert rv == ERR(E_LIB)


@pytest.mark.skipif(H.libzstd_frame(b"x") is None, reason="libzstd not on this box")
def test_plain_zst_errors(lib):
    a = cases.text(200000, 91)
    f = H.libzstd_frame(a, 3, checksum=1)
    for bad in (f[:len(f) // 2], f[:-1]):                # truncated inside a block / inside the checksum
        rv, _, _, _ = H.zstdmt_decompress_via(lib, bad)
        assert rv == ERR(E_LIB)
    dmg = bytearray(f)
    dmg[len(f) // 2] ^= 0x40
    rv, _, _, _ = H.zstdmt_decompress_via(lib, bytes(dmg))
    assert rv == ERR(E_LIB)
    rv, _, _, _ = H.zstdmt_decompress_via(lib, f + b"garbage after the frame....")
    assert rv == ERR(E_LIB)
