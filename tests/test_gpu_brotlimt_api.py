"""The drop-in boundary for brotli-mt decompression: BROTLIMT_* of libzstdmt_amd.so
(include/brotli-mt.h) driven through the reference's callback protocol (lib/brotli-mt.h), against
streams the reference wrote and -- where oracle/_ref travelled -- the reference library itself,
call for call."""
import ctypes as C
import json
import os

import pytest

import helpers as H
from golden import cases

pytestmark = pytest.mark.gpu

BDIR = os.path.join(H.GOLDEN_DIR, "brotli")
MAN = json.load(open(os.path.join(BDIR, "manifest.json")))["cases"]
ERR = lambda e: C.c_size_t(-e).value  # noqa: E731  (size_t)-enum
E_MEM, E_READ, E_WRITE, E_DATA, E_FC, E_FD, E_PARAM, E_LIB, E_CANCEL = range(1, 10)
needs_ref = pytest.mark.skipif(not H.have_bref(), reason="reference build not on this box")


@pytest.fixture(scope="module")
def lib():
    from zstdmt_amd._native import lib_path
    return H.bind_lz4mt(C.CDLL(lib_path()), "BROTLIMT_")


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


def _strip_eof(reads):
    r = list(reads)
    while r and r[-1][1] == 0:
        r.pop()
    return r


@pytest.mark.parametrize("name", sorted(MAN))
def test_decompress_reference_streams(lib, name):
    ent = MAN[name]
    st = _stream(name)
    rv, out, io, stats = H.brotlimt_decompress_via(lib, st, threads=4)
    assert rv == 0
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]
    # frames written, bytes read (headers included), bytes written -- as the reference counts them
    assert stats == (ent["frames"], ent["d_insize"], ent["d_outsize"])
    # one fn_write per frame, in order
    assert len(io.writes) == ent["frames"] and sum(io.writes) == ent["in_len"]
    # reads: 4-byte sniff, 12 header bytes, payload; then 16 + payload per record (pt_read)
    assert io.reads[0] == (4, 4) and io.reads[1] == (12, 12)


@needs_ref
@pytest.mark.parametrize("name", ["b_empty", "b_hello", "b_text_3x128k", "b_english_chunks", "b_mixed_l7"])
def test_same_callback_trace_as_reference(lib, name):
    """T=1 on the reference is fully sequential: identical request sizes, identical writes"""
    st = _stream(name)
    rv_r, out_r, io_r, stats_r = H.brotlimt_decompress_via(H.bref(), st, threads=1)
    rv_o, out_o, io_o, stats_o = H.brotlimt_decompress_via(lib, st, threads=1)
    assert rv_r == 0 and rv_o == 0 and out_r == out_o and stats_r == stats_o
    assert _strip_eof(io_r.reads) == _strip_eof(io_o.reads)
    assert io_r.writes == io_o.writes


@needs_ref
@pytest.mark.parametrize("level,chunk,mib", [(1, 0, 9), (0, 65536, 3), (4, 1 << 20, 6), (6, 0, 7), (9, 1 << 19, 3)])
def test_many_records(lib, level, chunk, mib):
    data = cases.text((mib << 20) + 4321, 70 + level) + cases.english(400000, level)
    rv, st, _, cstats = H.brotlimt_compress_via(H.bref(), data, chunk, threads=16, level=level)
    assert rv == 0
    rv, out, io, stats = H.brotlimt_decompress_via(lib, st, threads=8)
    assert rv == 0 and out == data
    assert stats == (cstats[0], len(st), len(data))


def test_errors(lib):
    st = _stream("b_text_3x128k")
    # not a skippable frame at all
    rv, _, _, _ = H.brotlimt_decompress_via(lib, b"\x00" * 64)
    assert rv == ERR(E_DATA)
    # shorter than the sniff
    rv, _, _, _ = H.brotlimt_decompress_via(lib, st[:3])
    assert rv == ERR(E_DATA)
    # header cut: read_fail; payload cut: data_error (pt_read)
    rv, _, _, _ = H.brotlimt_decompress_via(lib, st[:10])
    assert rv == ERR(E_READ)
    rv, _, _, _ = H.brotlimt_decompress_via(lib, st[:16 + 100])
    assert rv == ERR(E_DATA)
    # bad length field / bad "BR"
    bad = bytearray(st)
    bad[4] = 4
    rv, _, _, _ = H.brotlimt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_DATA)
    bad = bytearray(st)
    bad[12] = 0x41
    rv, _, _, _ = H.brotlimt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_DATA)
    # second record's magic
    ro = 16 + int.from_bytes(st[8:12], "little")
    bad = bytearray(st)
    bad[ro] ^= 0xFF
    rv, _, _, _ = H.brotlimt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_DATA)
    # a damaged brotli stream / a hint that is too small: frame_decompress
    bad = bytearray(st)
    bad[16 + 50] ^= 0x55
    bad[16 + 51] ^= 0xAA
    rv, _, _, _ = H.brotlimt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_FD)
    bad = bytearray(st)
    bad[14] = 1  # first record decodes to 128 KiB
    rv, _, _, _ = H.brotlimt_decompress_via(lib, bytes(bad))
    assert rv == ERR(E_FD)
    assert lib.BROTLIMT_isError(rv) == 1
    assert lib.BROTLIMT_getErrorString(rv) == b"Could not decompress frame at once"


def test_callback_errors_and_arguments(lib):
    st = _stream("b_text_3x128k")
    for fail_at, code, want in ((0, -1, E_READ), (1, -2, E_CANCEL), (2, -3, E_MEM)):
        io = H.MemIO(st, fail_read_at=fail_at, read_rv=code)
        ctx = lib.BROTLIMT_createDCtx(2, 0)
        assert lib.BROTLIMT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(want)
        lib.BROTLIMT_freeDCtx(ctx)
    io = H.MemIO(st, fail_write_at=1, write_rv=-1)
    ctx = lib.BROTLIMT_createDCtx(2, 0)
    assert lib.BROTLIMT_decompressDCtx(ctx, C.byref(io.rdwr)) == ERR(E_READ)  # mt_error maps -1 to read_fail
    lib.BROTLIMT_freeDCtx(ctx)
    assert not lib.BROTLIMT_createDCtx(0, 0) and not lib.BROTLIMT_createDCtx(129, 0)
    assert lib.BROTLIMT_decompressDCtx(None, None) == ERR(E_PARAM)
    # compression side: argument validation as the reference
    assert not lib.BROTLIMT_createCCtx(0, 3, 0) and not lib.BROTLIMT_createCCtx(1, 12, 0)
    assert lib.BROTLIMT_compressCCtx(None, None) == ERR(E_PARAM)


# ---- compression: decompress-identical (brotli's bytes are version dependent) ----
ZC_CASES = {
    "empty": (b"", 0, 3), "one": (b"x", 0, 1), "hello": (b"hello world, hello world, hello!", 65536, 0),
    "text_3x128k": (cases.text(3 * 131072 + 100, 11), 131072, 4),
    "text_default_chunk": (cases.text(2500000, 12), 0, 1),        # 1 MiB chunks at level 1
    "english_64k_chunks": (cases.english(300000, 13), 65536, 6),
    "random": (cases.rnd(200000, 14), 131072, 2),
    "zeros": (bytes(700000), 0, 1),
    "mixed": (cases.text(50000, 4) + bytes(70000) + cases.rnd(3000, 5) + cases.english(100000, 6), 1 << 20, 5),
    "allbytes": (bytes(range(256)) * 300 + cases.text(40000, 8), 131072, 9),
    "two_symbols": (bytes(65 + (b & 1) for b in cases.rnd(70000, 13)), 65536, 3),
}


@pytest.mark.parametrize("name", sorted(ZC_CASES))
def test_compress_decompress_identical(lib, name):
    data, chunk, level = ZC_CASES[name]
    rv, st, io, stats = H.brotlimt_compress_via(lib, data, chunk, threads=4, level=level)
    assert rv == 0
    eff = chunk or (1 << 20) * max(level, 1)
    nrec = max(1, -(-len(data) // eff))
    assert stats == (nrec, len(data), len(st)) and len(io.writes) == nrec
    assert all(want == eff for want, _ in io.reads)
    # record headers: magic, 8, csize, "BR", hint (lib/brotli-mt_compress.c:285-304)
    ip = 0
    for i in range(nrec):
        import struct
        magic, eight, csize, br, hint = struct.unpack_from("<IIIHH", st, ip)
        clen = min(eff, len(data) - i * eff)
        assert (magic, eight, br) == (0x184D2A50, 8, 0x5242)
        assert hint == ((clen >> 16) + 1 if clen < eff else eff >> 16)
        ip += 16 + csize
    assert ip == len(st)
    # the oracle, the device decoder and -- where it travelled -- the reference library decode it
    assert H.oracle_brotlimt_decompress(st, len(data) + 65536) == data
    rv, out, _, dstats = H.brotlimt_decompress_via(lib, st, threads=4)
    assert rv == 0 and out == data and dstats == (nrec, len(st), len(data))
    if H.have_bref():
        rv, out, _, _ = H.brotlimt_decompress_via(H.bref(), st, threads=4)
        assert rv == 0 and out == data
    if len(data) > 100000 and name not in ("random",):
        assert len(st) < len(data)


@needs_ref
def test_compress_same_callback_trace_as_reference(lib):
    """request sizes, write count and counters of the reference; the payload bytes differ by design"""
    data = cases.text(5 * 65536 + 777, 21)
    rv_r, st_r, io_r, stats_r = H.brotlimt_compress_via(H.bref(), data, 65536, threads=1, level=3)
    rv_o, st_o, io_o, stats_o = H.brotlimt_compress_via(lib, data, 65536, threads=1, level=3)
    assert rv_r == 0 and rv_o == 0
    assert _strip_eof(io_r.reads) == _strip_eof(io_o.reads)
    assert len(io_r.writes) == len(io_o.writes) and stats_r[:2] == stats_o[:2]
    # same hints in the same places
    for st in (st_r, st_o):
        assert st[12:14] == b"BR"
    assert st_r[14:16] == st_o[14:16]


def test_compress_many_batches_and_determinism(lib):
    data = cases.text(40 << 20, 31)
    rv, st1, _, stats = H.brotlimt_compress_via(lib, data, 1 << 20, threads=8, level=1)
    assert rv == 0 and stats[0] == 40
    rv, st2, _, _ = H.brotlimt_compress_via(lib, data, 1 << 20, threads=2, level=1)
    assert rv == 0 and st1 == st2
    rv, out, _, _ = H.brotlimt_decompress_via(lib, st1, threads=8)
    assert rv == 0 and out == data


def test_quality_reaches_the_encoder(lib):
    """BROTLIMT_createCCtx(level) selects the device encoder's tier (0-3 / 4-8 / 9-11), as the reference hands the
    level to BrotliEncoderCompress (lib/brotli-mt_compress.c:269-272): decompress-identical, ratio monotone"""
    data = cases.text(6 << 20, 33)
    sizes = []
    for q in (1, 5, 11):
        rv, st, _, _ = H.brotlimt_compress_via(lib, data, 1 << 20, threads=4, level=q)
        assert rv == 0 and H.oracle_brotlimt_decompress(st, len(data) + 65536) == data
        rv, out, _, _ = H.brotlimt_decompress_via(lib, st, threads=4)
        assert rv == 0 and out == data
        sizes.append(len(st))
    assert sizes[0] > sizes[1] > sizes[2]


def test_compress_callback_errors(lib):
    data = cases.text(400000, 41)
    for fail_at, code, want in ((0, -1, E_READ), (1, -2, E_CANCEL), (2, -3, E_MEM)):
        io = H.MemIO(data, fail_read_at=fail_at, read_rv=code)
        ctx = lib.BROTLIMT_createCCtx(2, 1, 65536)
        assert lib.BROTLIMT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(want)
        lib.BROTLIMT_freeCCtx(ctx)
    io = H.MemIO(data, fail_write_at=1, write_rv=-1)
    ctx = lib.BROTLIMT_createCCtx(2, 1, 65536)
    assert lib.BROTLIMT_compressCCtx(ctx, C.byref(io.rdwr)) == ERR(E_READ)
    lib.BROTLIMT_freeCCtx(ctx)


def test_context_reuse(lib):
    a, b = _stream("b_text_3x128k"), _stream("b_english_chunks")
    ctx = lib.BROTLIMT_createDCtx(2, 0)
    for st, name in ((a, "b_text_3x128k"), (b, "b_english_chunks"), (a, "b_text_3x128k")):
        io = H.MemIO(st)
        assert lib.BROTLIMT_decompressDCtx(ctx, C.byref(io.rdwr)) == 0
        assert H.sha256(io.result()) == MAN[name]["in_sha256"]
    lib.BROTLIMT_freeDCtx(ctx)


def test_chunk_not_a_multiple_of_64k(lib):
    """inputsize = 100 000: the hint of a full chunk must cover it (the reference truncates
    inputsize >> 16 and then cannot decode its own records); ours rounds up, so both decoders
    accept what BROTLIMT_compressCCtx of this library writes."""
    data = cases.text(350000, 7)
    rv, st, _, _ = H.brotlimt_compress_via(lib, data, 100000, threads=2, level=1)
    assert rv == 0
    rv, out, _, _ = H.brotlimt_decompress_via(lib, st)
    assert rv == 0 and out == data
    if H.have_bref():
        rv, out, _, _ = H.brotlimt_decompress_via(H.bref(), st)
        assert rv == 0 and out == data
