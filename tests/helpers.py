"""Shared test plumbing: ctypes bindings for the oracle (CPU restatement) and, when it has been
built in this container, the reference's own lz4-mt sources (oracle/_ref/liblz4mt_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/.
"""
import ctypes as C
import hashlib
import os
import struct
import fcntl
import subprocess
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "liblz4mt_ref.so")
ZREF_SO = os.path.join(ORACLE_DIR, "_ref", "libzstdmt_ref.so")
BREF_SO = os.path.join(ORACLE_DIR, "_ref", "libbrotlimt_ref.so")
BROTLI_BLOB = os.path.join(ROOT, "zstdmt_amd", "csrc", "data", "brotli_static.bin")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

_oracle = None
_ref = None
_zref = None



def locked_make(directory, target, **kw):
    """`make -C directory target` under a file lock: the test workers of `pytest -n N` build the same libraries"""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, ".make.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-C", directory, target], **kw)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def build_oracle():
    """(Re)build liboracle.so if missing or stale.  Building the checker is not using it."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("lz4_oracle.c", "zstd_oracle.c", "brotli_oracle.c", "zmt_oracle.h")]
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < max(map(os.path.getmtime, srcs)):
        locked_make(ORACLE_DIR, "liboracle.so",
                              stdout=subprocess.DEVNULL)
    return ORACLE_SO


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        sz, p8 = C.c_size_t, C.c_char_p
        lib.zo_xxh32.restype = C.c_uint32
        lib.zo_xxh32.argtypes = [p8, sz, C.c_uint32]
        lib.zo_lz4f_bound.restype = sz
        lib.zo_lz4f_bound.argtypes = [sz]
        for name in ("zo_lz4f_compress", "zo_lz4f_decompress"):
            f = getattr(lib, name)
            f.restype = sz
            f.argtypes = [p8, sz, C.c_void_p, sz]
        lib.zo_lz4mt_compress_bound.restype = sz
        lib.zo_lz4mt_compress_bound.argtypes = [sz, sz]
        lib.zo_lz4mt_compress.restype = sz
        lib.zo_lz4mt_compress.argtypes = [C.c_void_p, sz, sz, C.c_void_p, sz]
        lib.zo_lz4mt_decompress.restype = sz
        lib.zo_lz4mt_decompress.argtypes = [C.c_void_p, sz, C.c_void_p, sz]
        lib.zo_lz4mt_compress_mt.restype = sz
        lib.zo_lz4mt_compress_mt.argtypes = [C.c_void_p, sz, sz, C.c_void_p, sz, C.c_int]
        lib.zo_lz4mt_decompress_mt.restype = sz
        lib.zo_lz4mt_decompress_mt.argtypes = [C.c_void_p, sz, C.c_void_p, sz, C.c_int]
        lib.zo_lz4mt_compress_level.restype = sz
        lib.zo_lz4mt_compress_level.argtypes = [C.c_void_p, sz, sz, C.c_void_p, sz, C.c_int]
        lib.zo_xxh64.restype = C.c_uint64
        lib.zo_xxh64.argtypes = [p8, sz, C.c_uint64]
        lib.zo_zstd_frame_content_size.restype = C.c_uint64
        lib.zo_zstd_frame_content_size.argtypes = [p8, sz]
        lib.zo_zstd_decompress_frame.restype = sz
        lib.zo_zstd_decompress_frame.argtypes = [p8, sz, C.c_void_p, sz, C.POINTER(sz)]
        lib.zo_zstdmt_decompress.restype = sz
        lib.zo_zstdmt_decompress.argtypes = [p8, sz, C.c_void_p, sz]
        lib.zo_brotli_decompress.restype = C.c_long
        lib.zo_brotli_decompress.argtypes = [p8, sz, C.c_void_p, sz, p8]
        lib.zo_brotli_transform.restype = sz
        lib.zo_brotli_transform.argtypes = [p8, C.c_void_p, p8, C.c_uint32, C.c_uint32]
        lib.zo_brotlimt_decompress.restype = sz
        lib.zo_brotlimt_decompress.argtypes = [p8, sz, C.c_void_p, sz, p8]
        lib.zo_snappy_decompress.restype = sz
        lib.zo_snappy_decompress.argtypes = [p8, sz, C.c_void_p, sz]
        lib.zo_snappymt_decompress.restype = sz
        lib.zo_snappymt_decompress.argtypes = [p8, sz, C.c_void_p, sz]
        lib.zo_snappy_uncompressed_length.restype = sz
        lib.zo_snappy_uncompressed_length.argtypes = [p8, sz, C.POINTER(C.c_uint32)]
        _oracle = lib
    return _oracle


SIZE_ERR = C.c_size_t(-1).value


def oracle_compress(data: bytes, chunk: int) -> bytes:
    lib = oracle()
    cap = lib.zo_lz4mt_compress_bound(len(data), chunk)
    out = C.create_string_buffer(cap)
    n = lib.zo_lz4mt_compress(data, len(data), chunk, out, cap)
    assert n != SIZE_ERR
    return out.raw[:n]


def oracle_compress_level(data: bytes, chunk: int, level: int) -> bytes:
    """lz4-mt stream at `level`: 1-2 = LZ4 fast (lz4_oracle.c), 3-9 = LZ4 HC hash chain, 10-12 = its optimal parser (lz4hc_oracle.c)"""
    lib = oracle()
    cap = len(data) + len(data) // 64 + (len(data) // max(chunk, 1) + 2) * 64 + 1024
    out = C.create_string_buffer(cap)
    n = lib.zo_lz4mt_compress_level(data, len(data), chunk, out, cap, level)
    assert n != SIZE_ERR
    return out.raw[:n]


def oracle_decompress(stream: bytes, cap: int):
    """Returns bytes, or None on malformed input."""
    lib = oracle()
    out = C.create_string_buffer(max(cap, 1))
    n = lib.zo_lz4mt_decompress(stream, len(stream), out, cap)
    if n in (SIZE_ERR, SIZE_ERR - 1):
        return None
    return out.raw[:n]


def oracle_frame_compress(data: bytes) -> bytes:
    lib = oracle()
    cap = lib.zo_lz4f_bound(len(data))
    out = C.create_string_buffer(cap)
    n = lib.zo_lz4f_compress(data, len(data), out, cap)
    assert n != SIZE_ERR
    return out.raw[:n]


# ----------------------------------------------------------------------------------------------
# The reference itself (only where oracle/_ref has been built: this container, or shipped .so)
# ----------------------------------------------------------------------------------------------
class RefBuffer(C.Structure):  # LZ4MT_Buffer, /root/reference/lib/lz4-mt.h:67-71
    _fields_ = [("buf", C.c_void_p), ("size", C.c_size_t), ("allocated", C.c_size_t)]


RD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(RefBuffer))


class RefRdWr(C.Structure):  # LZ4MT_RdWr_t, /root/reference/lib/lz4-mt.h:84-89
    _fields_ = [("fn_read", RD_FN), ("arg_read", C.c_void_p),
                ("fn_write", RD_FN), ("arg_write", C.c_void_p)]


def have_ref():
    return os.path.exists(REF_SO)


def bind_lz4mt(lib, pfx="LZ4MT_"):
    """Attach prototypes for the LZ4MT_* (or ZSTDCB_*: same shapes, lib/zstd-mt.h:115-205) ABI,
    shared by the reference .so and our own library."""
    vp, sz = C.c_void_p, C.c_size_t
    if pfx != "LZ4MT_":
        for n, (res, args) in {
            "createCCtx": (vp, [C.c_int, C.c_int, C.c_int]), "compressCCtx": (sz, [vp, C.POINTER(RefRdWr)]),
            "freeCCtx": (None, [vp]), "createDCtx": (vp, [C.c_int, C.c_int]),
            "decompressDCtx": (sz, [vp, C.POINTER(RefRdWr)]), "freeDCtx": (None, [vp]),
            "GetFramesCCtx": (sz, [vp]), "GetInsizeCCtx": (sz, [vp]), "GetOutsizeCCtx": (sz, [vp]),
            "GetFramesDCtx": (sz, [vp]), "GetInsizeDCtx": (sz, [vp]), "GetOutsizeDCtx": (sz, [vp]),
            "isError": (C.c_uint, [sz]), "getErrorString": (C.c_char_p, [sz]),
        }.items():
            f = getattr(lib, pfx + n)
            f.restype = res
            f.argtypes = args
        return lib
    lib.LZ4MT_createCCtx.restype = vp
    lib.LZ4MT_createCCtx.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.LZ4MT_compressCCtx.restype = sz
    lib.LZ4MT_compressCCtx.argtypes = [vp, C.POINTER(RefRdWr)]
    lib.LZ4MT_freeCCtx.argtypes = [vp]
    lib.LZ4MT_freeCCtx.restype = None
    lib.LZ4MT_createDCtx.restype = vp
    lib.LZ4MT_createDCtx.argtypes = [C.c_int, C.c_int]
    lib.LZ4MT_decompressDCtx.restype = sz
    lib.LZ4MT_decompressDCtx.argtypes = [vp, C.POINTER(RefRdWr)]
    lib.LZ4MT_freeDCtx.argtypes = [vp]
    lib.LZ4MT_freeDCtx.restype = None
    for n in ("GetFramesCCtx", "GetInsizeCCtx", "GetOutsizeCCtx",
              "GetFramesDCtx", "GetInsizeDCtx", "GetOutsizeDCtx"):
        f = getattr(lib, "LZ4MT_" + n)
        f.restype = sz
        f.argtypes = [vp]
    lib.LZ4MT_isError.restype = C.c_uint
    lib.LZ4MT_isError.argtypes = [sz]
    lib.LZ4MT_getErrorString.restype = C.c_char_p
    lib.LZ4MT_getErrorString.argtypes = [sz]
    return lib


def ref():
    global _ref
    if _ref is None:
        _ref = bind_lz4mt(C.CDLL(REF_SO))
    return _ref


def have_zref():
    return os.path.exists(ZREF_SO)


def zref():
    """The reference's zstd-mt library (lib/zstd-mt_*.c + the image's libzstd 1.4.9)."""
    global _zref
    if _zref is None:
        _zref = bind_lz4mt(C.CDLL(ZREF_SO), "ZSTDCB_")
    return _zref


class MemIO:
    """In-memory fn_read / fn_write pair obeying the callback protocol of lib/README.md:19-24.
    Records every request so tests can assert the read/write pattern."""

    def __init__(self, data: bytes, fail_read_at=None, fail_write_at=None, read_rv=-1, write_rv=-1):
        self.data = data
        self.pos = 0
        self.out = []
        self.reads = []
        self.writes = []
        self.read_threads = set()    # threading.get_ident() of every fn_read / fn_write call
        self.write_threads = set()
        self.read_cpus = set()       # os.sched_getaffinity of the calling thread, as frozensets (what mt_bind_near left)
        self.write_cpus = set()
        self.lock = threading.Lock()
        self.fail_read_at = fail_read_at
        self.fail_write_at = fail_write_at
        self.read_rv = read_rv
        self.write_rv = write_rv
        self._rd = RD_FN(self._read)
        self._wr = RD_FN(self._write)
        self.rdwr = RefRdWr(self._rd, None, self._wr, None)

    def _read(self, _arg, bufp):
        b = bufp.contents
        with self.lock:
            self.read_threads.add(threading.get_ident())
            self.read_cpus.add(frozenset(os.sched_getaffinity(0)))
            if self.fail_read_at is not None and len(self.reads) >= self.fail_read_at:
                return self.read_rv
            want = b.size
            n = min(want, len(self.data) - self.pos)
            if n:
                C.memmove(b.buf, self.data[self.pos:self.pos + n], n)
            self.pos += n
            b.size = n
            self.reads.append((want, n))
        return 0

    def _write(self, _arg, bufp):
        b = bufp.contents
        with self.lock:
            self.write_threads.add(threading.get_ident())
            self.write_cpus.add(frozenset(os.sched_getaffinity(0)))
            if self.fail_write_at is not None and len(self.writes) >= self.fail_write_at:
                return self.write_rv
            self.out.append(C.string_at(b.buf, b.size))
            self.writes.append(b.size)
        return 0

    def result(self) -> bytes:
        return b"".join(self.out)


def lz4mt_compress_via(lib, data: bytes, chunk: int, threads: int = 1, level: int = 1, pfx="LZ4MT_"):
    """Run <lib>.<pfx>compressCCtx over in-memory callbacks; returns (rv, stream, io, stats)."""
    g = lambda n: getattr(lib, pfx + n)
    io = MemIO(data)
    ctx = g("createCCtx")(threads, level, chunk)
    assert ctx
    rv = g("compressCCtx")(ctx, C.byref(io.rdwr))
    stats = (g("GetFramesCCtx")(ctx), g("GetInsizeCCtx")(ctx), g("GetOutsizeCCtx")(ctx))
    g("freeCCtx")(ctx)
    return rv, io.result(), io, stats


def lz4mt_decompress_via(lib, stream: bytes, threads: int = 1, inputsize: int = 0, pfx="LZ4MT_"):
    g = lambda n: getattr(lib, pfx + n)
    io = MemIO(stream)
    ctx = g("createDCtx")(threads, inputsize)
    assert ctx
    rv = g("decompressDCtx")(ctx, C.byref(io.rdwr))
    stats = (g("GetFramesDCtx")(ctx), g("GetInsizeDCtx")(ctx), g("GetOutsizeDCtx")(ctx))
    g("freeDCtx")(ctx)
    return rv, io.result(), io, stats


def zstdmt_compress_via(lib, data, chunk, threads=1, level=1):
    return lz4mt_compress_via(lib, data, chunk, threads, level, pfx="ZSTDCB_")


def zstdmt_decompress_via(lib, stream, threads=2, inputsize=0):
    return lz4mt_decompress_via(lib, stream, threads, inputsize, pfx="ZSTDCB_")


def oracle_zstdmt_decompress(stream: bytes, cap: int):
    lib = oracle()
    out = C.create_string_buffer(max(cap, 1))
    n = lib.zo_zstdmt_decompress(stream, len(stream), out, cap)
    return None if n == SIZE_ERR else out.raw[:n]


# ----------------------------------------------------------------------------------------------
# Deterministic input generators
# brotli-mt (decompress path) ------------------------------------------------------------------
_bref = None
_blob = None
_benc = None
_bdec = None


def brotli_blob() -> bytes:
    """RFC 7932 constant data (dictionary, transforms, context lookup): tools/gen_brotli_tables.py."""
    global _blob
    if _blob is None:
        with open(BROTLI_BLOB, "rb") as f:
            _blob = f.read()
    return _blob


def have_bref():
    return os.path.exists(BREF_SO)


def bref():
    """The reference's brotli-mt library (lib/brotli-mt_*.c + the image's brotli 1.0.9)."""
    global _bref
    if _bref is None:
        _bref = bind_lz4mt(C.CDLL(BREF_SO), "BROTLIMT_")
    return _bref


def brotlimt_compress_via(lib, data, chunk, threads=1, level=3):
    return lz4mt_compress_via(lib, data, chunk, threads, level, pfx="BROTLIMT_")


def brotlimt_decompress_via(lib, stream, threads=2, inputsize=0):
    return lz4mt_decompress_via(lib, stream, threads, inputsize, pfx="BROTLIMT_")


def have_libbrotli():
    return os.path.exists("/opt/conda/lib/libbrotlienc.so.1")


def libbrotli_compress(data: bytes, quality=5, lgwin=22, mode=0) -> bytes:
    """One raw brotli stream from the image's libbrotlienc 1.0.9 (what BROTLIMT_compressCCtx calls,
    lib/brotli-mt_compress.c:269-272, with lgwin = 24)."""
    global _benc
    if _benc is None:
        _benc = C.CDLL("/opt/conda/lib/libbrotlienc.so.1")
        _benc.BrotliEncoderCompress.restype = C.c_int
        _benc.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p,
                                                C.POINTER(C.c_size_t), C.c_void_p]
    cap = len(data) + len(data) // 2 + 1024
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    assert _benc.BrotliEncoderCompress(quality, lgwin, mode, len(data), data, C.byref(n), out) == 1
    return out.raw[:n.value]


def libbrotli_decompress(stream: bytes, cap: int):
    """BrotliDecoderDecompress (lib/brotli-mt_decompress.c:344-346): bytes, or None unless SUCCESS."""
    global _bdec
    if _bdec is None:
        _bdec = C.CDLL("/opt/conda/lib/libbrotlidec.so.1")
        _bdec.BrotliDecoderDecompress.restype = C.c_int
        _bdec.BrotliDecoderDecompress.argtypes = [C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_void_p]
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t(cap)
    rv = _bdec.BrotliDecoderDecompress(len(stream), stream, C.byref(n), out)
    return out.raw[:n.value] if rv == 1 else None


def oracle_brotli_decompress(stream: bytes, cap: int):
    """zo_brotli_decompress: bytes, or the negative status."""
    out = C.create_string_buffer(max(cap, 1))
    n = oracle().zo_brotli_decompress(stream, len(stream), out, cap, brotli_blob())
    return out.raw[:n] if n >= 0 else n


def oracle_brotlimt_decompress(stream: bytes, cap: int):
    out = C.create_string_buffer(max(cap, 1))
    n = oracle().zo_brotlimt_decompress(stream, len(stream), out, cap, brotli_blob())
    return None if n == SIZE_ERR else out.raw[:n]


def brotli_record(payload: bytes, hint: int) -> bytes:
    """16-byte brotli-mt record header + payload (lib/brotli-mt_compress.c:285-304)."""
    import struct
    return struct.pack("<IIIHH", 0x184D2A50, 8, len(payload), 0x5242, hint) + payload


# ----------------------------------------------------------------------------------------------
def lcg(n: int, seed: int) -> bytes:
    """x = x*6364136223846793005 + 1442695040888963407 mod 2^64; byte = x >> 56."""
    import numpy as np
    out = np.empty(n, dtype=np.uint8)
    x = seed & 0xFFFFFFFFFFFFFFFF
    a, c, m = 6364136223846793005, 1442695040888963407, (1 << 64) - 1
    for i in range(n):
        x = (x * a + c) & m
        out[i] = x >> 56
    return out.tobytes()


def sha256(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def _cases():
    from golden import cases
    return cases


def soup(rng, n):
    """Borderline-compressible soup (runs, periods, noise, text, long-distance repeats, low-entropy
    bytes incl. values above 128): raw vs Huffman literals, raw vs compressed blocks, long matches."""
    out = bytearray()
    while len(out) < n:
        k = rng.choice([0, 1, 2, 3, 4, 5])
        m = rng.randrange(1, 9000)
        if k == 0:
            out += _cases().rnd(m, rng.randrange(1 << 30))
        elif k == 1:
            out += bytes([rng.randrange(256)]) * m
        elif k == 2:
            out += _cases().text(m, rng.randrange(1 << 30))
        elif k == 3:
            unit = _cases().rnd(rng.randrange(1, 40), rng.randrange(1 << 30))
            out += (unit * (m // len(unit) + 1))[:m]
        elif k == 4:
            back = rng.randrange(1, len(out) + 1) if out else 0
            if back:
                start = len(out) - back
                out += out[start:start + m]
        else:
            lo, span = rng.randrange(256), rng.choice([2, 4, 16, 64])
            out += bytes((lo + (b % span)) & 255 for b in _cases().rnd(m, rng.randrange(1 << 30)))
    return bytes(out[:n])




def libzstd_frame(data: bytes, level=1, checksum=0, content_size=1):
    """One zstd frame written by the image's libzstd with explicit frame parameters (test input
    generator for frame flavours zstd-mt itself never writes).  None if libzstd is not present."""
    path = "/opt/conda/lib/libzstd.so.1"
    if not os.path.exists(path):
        return None
    zs = C.CDLL(path)
    zs.ZSTD_createCCtx.restype = C.c_void_p
    zs.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    zs.ZSTD_compress2.restype = C.c_size_t
    zs.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    zs.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    cctx = zs.ZSTD_createCCtx()
    zs.ZSTD_CCtx_setParameter(cctx, 100, level)         # ZSTD_c_compressionLevel
    zs.ZSTD_CCtx_setParameter(cctx, 201, checksum)      # ZSTD_c_checksumFlag
    zs.ZSTD_CCtx_setParameter(cctx, 200, content_size)  # ZSTD_c_contentSizeFlag
    dst = C.create_string_buffer(len(data) + len(data) // 64 + 1024)
    n = zs.ZSTD_compress2(cctx, dst, len(dst), data, len(data))
    zs.ZSTD_freeCCtx(cctx)
    return dst.raw[:n]


def zstd_rle_mode_frame(nblocks=3, repeat_from=0, seq=(1000, 100, 5000), nseq=50):
    """A hand-made zstd frame (test vector generator): one raw block, then `nblocks` compressed blocks with raw literals
    whose three sequence tables are in RLE mode -- every sequence has the same LL / OF / ML code, so a sequence costs only its
    extra bits --; the blocks from index `repeat_from` on (0 = none) say Repeat_Mode for all three.  libzstd does not write such
    blocks on ordinary data.  -> (frame bytes, content bytes); RFC 8878 3.1.1.3.2."""
    cases = _cases()
    ll_base = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024,
               2048, 4096, 8192, 16384, 32768, 65536]
    ll_bits = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    ml_base = [3 + i for i in range(32)] + [35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195,
                                            16387, 32771, 65539]
    ml_bits = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]

    def block(lits, seqs, last, modes):
        acc, codes = 1, None
        for ll, ml, off in seqs:
            lc = max(i for i in range(36) if ll_base[i] <= ll)
            mc = max(i for i in range(53) if ml_base[i] <= ml)
            oc = (off + 3).bit_length() - 1
            assert codes in (None, (lc, oc, mc))
            codes = (lc, oc, mc)
            # read order of a sequence: offset, match length, literals length extra bits (the stream is read from its end)
            for v, nb in ((off + 3 - (1 << oc), oc), (ml - ml_base[mc], ml_bits[mc]), (ll - ll_base[lc], ll_bits[lc])):
                acc = (acc << nb) | v
        n = len(seqs)
        sq = (bytes([n]) if n < 128 else bytes([128 + (n >> 8), n & 255])) + bytes([modes])
        if modes == 0x54:
            sq += bytes(codes)
        sq += acc.to_bytes((acc.bit_length() + 7) // 8, "little")
        body = (0 | 3 << 2 | len(lits) << 4).to_bytes(3, "little") + lits + sq      # Raw_Literals_Block, 20-bit size
        return ((1 if last else 0) | 2 << 1 | len(body) << 3).to_bytes(3, "little") + body

    head = cases.rnd(8192, 3)
    content = bytearray(head)
    blocks = [(len(head) << 3).to_bytes(3, "little") + head]                          # Raw_Block
    for b in range(nblocks):
        lits = cases.text(nseq * seq[0] + 2000, seed=100 + b)
        seqs = [seq] * nseq
        lp = 0
        for ll, ml, off in seqs:
            content += lits[lp:lp + ll]
            lp += ll
            for _ in range(ml):
                content.append(content[-off])
        content += lits[lp:]
        blocks.append(block(lits, seqs, b == nblocks - 1, 0xFC if (repeat_from and b >= repeat_from) else 0x54))
    frame = bytes([0x28, 0xB5, 0x2F, 0xFD, 2 << 6 | 1 << 5]) + struct.pack("<I", len(content)) + b"".join(blocks)
    return frame, bytes(content)


def mt_record(frame: bytes) -> bytes:
    """12-byte skippable header + frame (lib/zstd-mt_compress.c:296-302)."""
    import struct
    return struct.pack("<III", 0x184D2A50, 4, len(frame)) + frame


def liblz4_frame(data: bytes, block_id=7, linked=1, content_size=0, checksum=1, level=1, block_checksum=0):
    """One LZ4 frame written by the image's liblz4 with explicit frame parameters (test input for the
    plain .lz4 flavours lz4-mt itself never writes: the lz4 tool's defaults are 4 MiB linked blocks, a
    content checksum and no content size).  None if liblz4 is not present."""
    path = next((p for p in ("/opt/conda/lib/liblz4.so.1", "/usr/lib/x86_64-linux-gnu/liblz4.so.1")
                 if os.path.exists(p)), None)
    if path is None:
        return None
    lz = C.CDLL(path)

    class FrameInfo(C.Structure):
        _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int),
                    ("frameType", C.c_int), ("contentSize", C.c_ulonglong), ("dictID", C.c_uint),
                    ("blockChecksumFlag", C.c_int)]

    class Prefs(C.Structure):
        _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                    ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]
    pr = Prefs()
    pr.frameInfo.blockSizeID = block_id
    pr.frameInfo.blockMode = 0 if linked else 1          # LZ4F_blockLinked = 0
    pr.frameInfo.contentChecksumFlag = checksum
    pr.frameInfo.contentSize = len(data) if content_size else 0
    pr.frameInfo.blockChecksumFlag = block_checksum
    pr.compressionLevel = level
    lz.LZ4F_compressFrameBound.restype = C.c_size_t
    lz.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(Prefs)]
    lz.LZ4F_compressFrame.restype = C.c_size_t
    lz.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Prefs)]
    cap = lz.LZ4F_compressFrameBound(len(data), C.byref(pr))
    dst = C.create_string_buffer(cap)
    n = lz.LZ4F_compressFrame(dst, cap, data, len(data), C.byref(pr))
    assert n <= cap
    return dst.raw[:n]


def dense_sequences(n, lits=b"aaaabbcd"):
    """7-byte words from a small dictionary with one literal between them: the densest sequence
    lists the device zstd encoder writes; lits=None makes the literals incompressible (raw sections)."""
    import random
    rng = random.Random(77)
    words = [bytes(rng.randrange(256) for _ in range(7)) for _ in range(48)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) + bytes([rng.choice(lits) if lits else rng.randrange(256)])
    return bytes(out[:n])


def zstd_walk_blocks(frame: bytes):
    """Blocks of one zstd frame (RFC 8878 3.1.1): list of dicts with type (0 raw, 1 RLE, 2 compressed),
    size, and for compressed blocks the literals type, the number of sequences and the
    Symbol_Compression_Modes byte (None without sequences).  Test helper, no validation."""
    assert frame[:4] == bytes([0x28, 0xB5, 0x2F, 0xFD])
    fhd = frame[4]
    ss, fcs_flag, did = (fhd >> 5) & 1, fhd >> 6, fhd & 3
    at = 5 + (0 if ss else 1) + (0, 1, 2, 4)[did] + ((1 if ss else 0), 2, 4, 8)[fcs_flag]
    out = []
    while True:
        bh = int.from_bytes(frame[at:at + 3], "little")
        last, typ, size = bh & 1, (bh >> 1) & 3, bh >> 3
        ent = {"type": typ, "size": size}
        body = frame[at + 3:at + 3 + (1 if typ == 1 else size)]
        if typ == 2:
            b0 = body[0]
            lt, sf = b0 & 3, (b0 >> 2) & 3
            if lt < 2:      # raw / RLE literals
                hl = (1, 2, 1, 3)[sf]
                regen = int.from_bytes(body[:hl], "little") >> (3 if hl == 1 else 4)
                lsz = hl + (regen if lt == 0 else 1)
            else:
                hl = (3, 3, 4, 5)[sf]
                v = int.from_bytes(body[:hl], "little")
                bits = (10, 10, 14, 18)[sf]
                lsz = hl + ((v >> (4 + bits)) & ((1 << bits) - 1))
            sq = body[lsz:]
            n, p = sq[0], 1
            if n == 255:
                n, p = int.from_bytes(sq[1:3], "little") + 0x7F00, 3
            elif n >= 128:
                n, p = ((n - 128) << 8) + sq[1], 2
            ent.update(lit_type=lt, nseq=n, modes=sq[p] if n else None)
        out.append(ent)
        at += 3 + (1 if typ == 1 else size)
        if last:
            return out


# snappy-mt ------------------------------------------------------------------------------------
LIBSNAPPY = "/opt/conda/lib/libsnappy.so.1"
_snappy = None


def have_libsnappy():
    return os.path.exists(LIBSNAPPY)


def libsnappy():
    """The image's libsnappy 1.1.8 through its C API (snappy-c.h).  Not the reference's library (that
    is a C port the zstdmt repository vendors outside /root/reference) but an independent
    implementation of the same format: the pin of oracle/snappy_oracle.c."""
    global _snappy
    if _snappy is None:
        L = C.CDLL(LIBSNAPPY)
        L.snappy_compress.restype = C.c_int
        L.snappy_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        L.snappy_uncompress.restype = C.c_int
        L.snappy_uncompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        L.snappy_max_compressed_length.restype = C.c_size_t
        L.snappy_max_compressed_length.argtypes = [C.c_size_t]
        _snappy = L
    return _snappy


def libsnappy_compress(data: bytes) -> bytes:
    L = libsnappy()
    cap = L.snappy_max_compressed_length(len(data))
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    assert L.snappy_compress(data, len(data), out, C.byref(n)) == 0
    return out.raw[:n.value]


def libsnappy_decompress(stream: bytes, cap: int):
    """snappy_uncompress: bytes, or None unless SNAPPY_OK."""
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t(cap)
    rv = libsnappy().snappy_uncompress(stream, len(stream), out, C.byref(n))
    return out.raw[:n.value] if rv == 0 else None


def oracle_snappy_decompress(stream: bytes, cap: int):
    out = C.create_string_buffer(max(cap, 1))
    n = oracle().zo_snappy_decompress(stream, len(stream), out, cap)
    return None if n == SIZE_ERR else out.raw[:n]


def oracle_snappymt_decompress(stream: bytes, cap: int):
    out = C.create_string_buffer(max(cap, 1))
    n = oracle().zo_snappymt_decompress(stream, len(stream), out, cap)
    return None if n == SIZE_ERR else out.raw[:n]


def snappy_record(payload: bytes, hint: int) -> bytes:
    """One snappy-mt record (lib/snappy-mt_compress.c:280-300): skippable magic, 8, size, "SP", hint."""
    return struct.pack("<IIIHH", 0x184D2A50, 8, len(payload), 0x5053, hint) + payload


def snappymt_stream(data: bytes, chunk: int) -> bytes:
    """What SNAPPYMT_compressCCtx writes for `data` (framing of the reference, payloads by libsnappy)."""
    out = []
    n = max(1, -(-len(data) // chunk))
    for i in range(n):
        part = data[i * chunk:(i + 1) * chunk]
        hint = (len(part) >> 16) + 1 if len(part) < chunk else chunk >> 16
        out.append(snappy_record(libsnappy_compress(part), hint))
    return b"".join(out)


def snappymt_compress_via(lib, data, chunk, threads=1, level=1):
    return lz4mt_compress_via(lib, data, chunk, threads, level, pfx="SNAPPYMT_")


def snappymt_decompress_via(lib, stream, threads=2, inputsize=0):
    return lz4mt_decompress_via(lib, stream, threads, inputsize, pfx="SNAPPYMT_")
