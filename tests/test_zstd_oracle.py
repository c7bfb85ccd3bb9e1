"""zstd-mt decode oracle (oracle/zstd_oracle.c) against the reference's own output.

Golden: streams written by the reference's lib/zstd-mt_*.c + libzstd 1.4.9 (tests/golden/zstd/,
generator gen_golden_zstd.py); the oracle must decode each to the plaintext whose SHA-256 the
manifest records.  Live: when oracle/_ref is present (this container, or shipped to the GPU box),
fresh streams at several levels / chunk sizes are compared too.
"""
import ctypes as C
import json
import os

import pytest

import helpers as H
from golden import cases

ZDIR = os.path.join(H.GOLDEN_DIR, "zstd")
MAN = json.load(open(os.path.join(ZDIR, "manifest.json")))["cases"]


def _stream(name):
    ent = MAN[name]
    if "out_file" in ent:
        return open(os.path.join(ZDIR, ent["out_file"]), "rb").read()
    return None


@pytest.mark.parametrize("name", sorted(MAN))
def test_golden_decode(name):
    ent = MAN[name]
    st = _stream(name)
    if st is None:
        if not H.have_zref():
            pytest.skip("stream too large to commit; needs oracle/_ref")
        level, chunk, thunk = cases.ZCASES[name]
        rv, st, _, _ = H.zstdmt_compress_via(H.zref(), thunk(), chunk, threads=2, level=level)
        assert rv == 0
    assert H.sha256(st) == ent["out_sha256"] or "out_file" not in ent
    out = H.oracle_zstdmt_decompress(st, ent["in_len"] + 64)
    assert out is not None
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]
    # the inputs are defined by generators: the decoded bytes equal the regenerated input
    assert out == cases.ZCASES[name][2]()


def test_frame_content_size_and_xxh64():
    lib = H.oracle()
    st = _stream("z_text_100")
    assert lib.zo_zstd_frame_content_size(st[12:], len(st) - 12) == 100
    # XXH64 known answers (xxhash reference vectors)
    assert lib.zo_xxh64(b"", 0, 0) == 0xEF46DB3751D8E999
    assert lib.zo_xxh64(b"a", 1, 0) == 0xD24EC4F1A98C6E5B
    assert lib.zo_xxh64(b"abc", 3, 0) == 0x44BC2CF5AD770999
    try:
        import xxhash
        d = cases.text(100000, 3)
        for n in (0, 1, 31, 32, 33, 63, 64, 1000, 100000):
            assert lib.zo_xxh64(d[:n], n, 0) == xxhash.xxh64(d[:n], seed=0).intdigest()
            assert lib.zo_xxh64(d[:n], n, 77) == xxhash.xxh64(d[:n], seed=77).intdigest()
    except ImportError:
        pass


@pytest.mark.parametrize("name", ["z_text_64k_l1", "z_text_3000", "z_hello"])
def test_corruption_is_rejected_or_harmless(name):
    """Flip / truncate: the oracle must never crash and never return wrong-length success."""
    st = bytearray(_stream(name))
    n = MAN[name]["in_len"]
    good = H.oracle_zstdmt_decompress(bytes(st), n + 64)
    for i in range(12, len(st), max(1, len(st) // 97)):
        bad = bytearray(st)
        bad[i] ^= 0x5A
        out = H.oracle_zstdmt_decompress(bytes(bad), n + 64)
        assert out is None or len(out) <= n + 64
    for cut in (1, 2, 5, len(st) // 2):
        assert H.oracle_zstdmt_decompress(bytes(st[:-cut]), n + 64) is None
    assert good is not None


@pytest.mark.skipif(not H.have_zref(), reason="reference build not present")
@pytest.mark.parametrize("level", [1, 2, 3, 6, 12, 19])
@pytest.mark.parametrize("chunk", [0, 65536, 300000])
def test_live_against_reference(level, chunk):
    z = H.zref()
    data = cases.text(400000, 100 + level) + cases.rnd(5000, level) + bytes(20000) + cases.text(90000, 3)
    rv, st, _, _ = H.zstdmt_compress_via(z, data, chunk, threads=3, level=level)
    assert rv == 0
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data


@pytest.mark.skipif(not H.have_zref(), reason="reference build not present")
def test_checksummed_and_unknown_size_frames():
    """Frames of the zstd CLI flavour (content checksum, streaming without content size)."""
    zs = C.CDLL("/opt/conda/lib/libzstd.so.1") if os.path.exists("/opt/conda/lib/libzstd.so.1") else None
    if zs is None:
        pytest.skip("libzstd not present")
    zs.ZSTD_createCCtx.restype = C.c_void_p
    zs.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    zs.ZSTD_compress2.restype = C.c_size_t
    zs.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    zs.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    data = cases.text(300000, 21)
    lib = H.oracle()
    for level, checksum, csize in ((1, 1, 1), (3, 1, 0), (7, 0, 0)):
        cctx = zs.ZSTD_createCCtx()
        zs.ZSTD_CCtx_setParameter(cctx, 100, level)       # ZSTD_c_compressionLevel
        zs.ZSTD_CCtx_setParameter(cctx, 201, checksum)    # ZSTD_c_checksumFlag
        zs.ZSTD_CCtx_setParameter(cctx, 200, csize)       # ZSTD_c_contentSizeFlag
        dst = C.create_string_buffer(len(data) + 1024)
        n = zs.ZSTD_compress2(cctx, dst, len(dst), data, len(data))
        zs.ZSTD_freeCCtx(cctx)
        assert n < len(data)
        out = C.create_string_buffer(len(data) + 64)
        used = C.c_size_t(0)
        got = lib.zo_zstd_decompress_frame(dst.raw[:n], n, out, len(out), C.byref(used))
        assert got == len(data) and used.value == n and out.raw[:got] == data
        if checksum:
            bad = bytearray(dst.raw[:n])
            bad[-1] ^= 1
            assert lib.zo_zstd_decompress_frame(bytes(bad), n, out, len(out), None) == H.SIZE_ERR
