"""Drive the fiber-emulated kernels (tests/emu/libzmt_emu.so) with the same call shapes as the
gpumt_* device API, on numpy host buffers.  TEST HARNESS ONLY."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

import helpers as H

EMU_DIR = os.path.join(H.ROOT, "tests", "emu")
_lib = None


def lib():
    global _lib
    if _lib is None:
        # EMU_STRICT=1: the build that checks every wave-uniformity claim (tests/test_emu_strict.py)
        name = "libzmt_emu_strict.so" if os.environ.get("EMU_STRICT") == "1" else "libzmt_emu.so"
        H.locked_make(EMU_DIR, name, stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(EMU_DIR, name))
        L.emu_lz4_slot_stride.restype = C.c_size_t
        L.emu_lz4_slot_stride.argtypes = [C.c_size_t]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def xxh32_batch(buf: np.ndarray, off: np.ndarray, length: np.ndarray) -> np.ndarray:
    n = len(off)
    out = np.zeros(n, np.uint32)
    lib().emu_xxh32_batch(_p(buf), _p(off.astype(np.uint64)), _p(length.astype(np.uint32)),
                          C.c_uint32(n), _p(out))
    return out


def compress(data: bytes, chunk: int, level: int = 1):
    """-> (stream bytes, rec_off[n+1], rec_len[n])"""
    L = lib()
    n = len(data)
    nrec = max(1, (n + chunk - 1) // chunk)
    stride = L.emu_lz4_slot_stride(chunk)
    inp = np.frombuffer(data + b"\0" * 64, np.uint8).copy()   # the device contract: 64 readable bytes past the input
    slots = np.full(nrec * stride, 0xEE, np.uint8)
    rec_len = np.zeros(nrec, np.uint32)
    if level >= 3:
        L.emu_lz4hc_compress_batch(_p(inp), C.c_uint64(n), C.c_uint32(chunk), _p(slots),
                                   C.c_uint64(stride), _p(rec_len), C.c_int(level))
    else:
        L.emu_lz4_compress_batch(_p(inp), C.c_uint64(n), C.c_uint32(chunk), _p(slots),
                                 C.c_uint64(stride), _p(rec_len))
    rec_off = np.zeros(nrec + 1, np.uint64)
    stream = np.full(int(rec_len.sum()) + 16, 0xDD, np.uint8)
    L.emu_lz4_compact(_p(slots), C.c_uint64(stride), _p(rec_len), C.c_uint32(nrec), _p(stream),
                      _p(rec_off))
    total = int(rec_off[nrec])
    return stream[:total].tobytes(), rec_off, rec_len


def walk_records(stream: bytes):
    """Host-side header walk (what pt_read does): -> rec_off[n], rec_len[n] or None if malformed"""
    import struct
    offs, lens = [], []
    i = 0
    while i < len(stream):
        if len(stream) - i < 12:
            return None
        c = struct.unpack_from("<I", stream, i + 8)[0]
        offs.append(i)
        lens.append(min(12 + c, len(stream) - i))
        i += 12 + c
    return np.array(offs, np.uint64), np.array(lens, np.uint32)


def decompress(stream: bytes, variant: int = 0, rec=None):
    """-> (content bytes, status[n]); variant 0 | ring << 4 = frames + parse4 + copy3 kernels (ring 12 if omitted), 1 = serial decoder"""
    L = lib()
    ro, rl = rec if rec is not None else walk_records(stream)
    nrec = len(ro)
    sbuf = np.frombuffer(stream + b"\0" * 512, np.uint8).copy()
    out_len = np.zeros(nrec, np.uint32)
    out_off = np.zeros(nrec + 1, np.uint64)
    L.emu_lz4_probe_sizes(_p(sbuf), _p(ro), _p(rl), C.c_uint32(nrec), _p(out_len), _p(out_off))
    total = int(out_off[nrec])
    out = np.full(total + 64, 0xCC, np.uint8)
    status = np.full(nrec, 99, np.uint32)
    L.emu_lz4_decompress_batch(C.c_int(variant), _p(sbuf), C.c_uint64(len(stream)), _p(ro), _p(rl),
                               C.c_uint32(nrec), _p(out), C.c_uint64(total), _p(out_off), _p(out_len),
                               _p(status))
    assert (out[total:] == 0xCC).all(), "decoder wrote past the end of its output"
    return out[:total].tobytes(), status


def zstd_decompress(stream: bytes, rec=None):
    """zstd-mt stream -> (content bytes, status[n]) through the emulated probe + decode kernels"""
    L = lib()
    ro, rl = rec if rec is not None else walk_records(stream)
    nrec = len(ro)
    pad = b"\xEE" * 300          # the device contract: 256 readable bytes past the stream
    sbuf = np.frombuffer(stream + pad, np.uint8).copy()
    out_len = np.zeros(nrec, np.uint32)
    status = np.full(nrec, 99, np.uint32)
    L.emu_zstd_probe(_p(sbuf), _p(ro), _p(rl), C.c_uint32(nrec), _p(out_len), _p(status))
    out_off = np.zeros(nrec + 1, np.uint64)
    out_off[1:] = np.cumsum(out_len.astype(np.uint64))
    total = int(out_off[nrec])
    out = np.full(total + 64, 0xCC, np.uint8)
    L.emu_zstd_decompress_batch(_p(sbuf), C.c_uint64(len(stream)), _p(ro), _p(rl), C.c_uint32(nrec),
                                _p(out), _p(out_off), _p(out_len), _p(status))
    assert (out[total:] == 0xCC).all(), "decoder wrote past the end of its output"
    return out[:total].tobytes(), status


def zstd_last_marks():
    """blocks whose sequences the pre-pass (zstd_dec_seq.hip) decoded ahead in the last zstd_decompress call"""
    L = lib()
    L.emu_zstd_last_marks.restype = C.c_uint32
    return int(L.emu_zstd_last_marks())


def walk_brotli_records(stream: bytes):
    """16-byte brotli-mt headers -> (payload offsets u64[n], payload sizes u32[n], capacities u32[n]);
    what the host engine does while reading (lib/brotli-mt_decompress.c:187-284)"""
    ro, rl, cap = [], [], []
    ip = 0
    while ip < len(stream):
        assert len(stream) - ip >= 16
        magic, eight, csize, br, hint = struct.unpack_from("<IIIHH", stream, ip)
        assert magic == 0x184D2A50 and eight == 8 and br == 0x5242
        ro.append(ip + 16)
        rl.append(csize)
        cap.append(hint << 16)
        ip += 16 + csize
    assert ip == len(stream)
    return np.array(ro, np.uint64), np.array(rl, np.uint32), np.array(cap, np.uint32)


def brotli_decompress(stream: bytes, grid: int = 2, rec=None):
    """brotli-mt stream -> (list of decoded records, status[n]) through the emulated decoder"""
    L = lib()
    ro, rl, cap = rec if rec is not None else walk_brotli_records(stream)
    nrec = len(ro)
    with open(H.BROTLI_BLOB, "rb") as f:
        blob = np.frombuffer(f.read(), np.uint8).copy()
    sbuf = np.frombuffer(stream + b"\xEE" * 300, np.uint8).copy()
    out_off = np.zeros(nrec + 1, np.uint64)
    out_off[1:] = np.cumsum(cap.astype(np.uint64))
    total = int(out_off[nrec])
    out = np.full(total + 64, 0xCC, np.uint8)
    out_len = np.full(nrec, 0xFFFFFFFF, np.uint32)
    status = np.full(nrec, 99, np.uint32)
    L.emu_brotli_decompress_batch(_p(sbuf), _p(ro), _p(rl), C.c_uint32(nrec), _p(out), _p(out_off), _p(cap),
                                  _p(out_len), _p(status), _p(blob), C.c_uint32(grid))
    assert (out[total:] == 0xCC).all(), "decoder wrote past the end of its output"
    recs = []
    for i in range(nrec):
        o = int(out_off[i])
        n = int(out_len[i]) if status[i] == 0 else 0
        assert n <= int(cap[i])
        assert (out[o + n:o + int(cap[i])] == 0xCC).all() or status[i] != 0, "wrote past the decoded size"
        recs.append(out[o:o + n].tobytes())
    return recs, status


def brotli_compress(data: bytes, chunk: int, grid: int = 3, level: int = 1):
    """-> brotli-mt stream bytes through the emulated block encoder (tier of quality `level`) + assemble + compact"""
    L = lib()
    L.emu_zstd_slot_stride.restype = C.c_size_t
    L.emu_zstd_slot_stride.argtypes = [C.c_size_t]
    n = len(data)
    nrec = max(1, (n + chunk - 1) // chunk)
    stride = L.emu_zstd_slot_stride(chunk)
    inp = np.frombuffer(data + b"\xEE" * 64, np.uint8).copy()
    slots = np.full(nrec * stride, 0xEE, np.uint8)
    rec_len = np.zeros(nrec, np.uint32)
    L.emu_brotli_compress_batch_level(_p(inp), C.c_uint64(n), C.c_uint32(chunk), _p(slots), C.c_uint64(stride),
                                      _p(rec_len), C.c_uint32(grid), C.c_int(level))
    rec_off = np.zeros(nrec + 1, np.uint64)
    stream = np.full(int(rec_len.sum()) + 16, 0xDD, np.uint8)
    L.emu_lz4_compact(_p(slots), C.c_uint64(stride), _p(rec_len), C.c_uint32(nrec), _p(stream), _p(rec_off))
    return stream[:int(rec_off[nrec])].tobytes()


def zstd_compress(data: bytes, chunk: int, grid: int = 3, level: int = 1):
    """-> zstd-mt stream bytes through the emulated block encoder (tier of `level`) + assemble + compact kernels"""
    L = lib()
    L.emu_zstd_slot_stride.restype = C.c_size_t
    L.emu_zstd_slot_stride.argtypes = [C.c_size_t]
    n = len(data)
    nrec = max(1, (n + chunk - 1) // chunk)
    stride = L.emu_zstd_slot_stride(chunk)
    inp = np.frombuffer(data + b"\xEE" * 64, np.uint8).copy()
    slots = np.full(nrec * stride, 0xEE, np.uint8)
    rec_len = np.zeros(nrec, np.uint32)
    L.emu_zstd_compress_batch_level(_p(inp), C.c_uint64(n), C.c_uint32(chunk), _p(slots), C.c_uint64(stride),
                                    _p(rec_len), C.c_uint32(grid), C.c_int(level))
    rec_off = np.zeros(nrec + 1, np.uint64)
    stream = np.full(int(rec_len.sum()) + 16, 0xDD, np.uint8)
    L.emu_lz4_compact(_p(slots), C.c_uint64(stride), _p(rec_len), C.c_uint32(nrec), _p(stream), _p(rec_off))
    return stream[:int(rec_off[nrec])].tobytes()


# snappy-mt ------------------------------------------------------------------------------------
def walk_snappy_records(stream: bytes):
    """(payload offsets, payload lengths, hints) of a snappy-mt stream, or None (host-side walk)."""
    ro, rl, hints = [], [], []
    at = 0
    while at < len(stream):
        if len(stream) - at < 16:
            return None
        magic, eight, csz, sp, hint = struct.unpack_from("<IIIHH", stream, at)
        if magic != 0x184D2A50 or eight != 8 or sp != 0x5053 or csz > len(stream) - at - 16:
            return None
        ro.append(at + 16)
        rl.append(csz)
        hints.append(hint)
        at += 16 + csz
    return np.array(ro, np.uint64), np.array(rl, np.uint32), hints


def snappy_preamble(payload: bytes):
    v = 0
    for i, b in enumerate(payload[:5]):
        v |= (b & 127) << (7 * i)
        if b < 128:
            return v if v < 1 << 32 else None
    return None


def snappy_decompress(stream: bytes, grid: int = 2, rec=None, caps=None):
    """Emulated gpumt_snappy_decompress_batch over the records of `stream` (capacity of a record = its
    preamble, as the engine sizes it; `caps` overrides).  -> (bytes of the accepted records, status[])"""
    L = lib()
    ro, rl = rec if rec is not None else walk_snappy_records(stream)[:2]
    n = len(ro)
    if caps is None:
        caps = []
        for o, ln in zip(ro.tolist(), rl.tolist()):
            v = snappy_preamble(stream[o:o + ln])
            caps.append(v if v is not None else 0)
    cap = np.array(caps, np.uint32)
    out_off = np.zeros(n + 1, np.uint64)
    out_off[1:] = np.cumsum(cap.astype(np.uint64))
    buf = np.frombuffer(stream + b"\0" * 256, np.uint8).copy()
    out = np.full(int(out_off[n]) + 64, 0xCC, np.uint8)
    out_len = np.zeros(n, np.uint32)
    status = np.full(n, 0xA5, np.uint32)
    L.emu_snappy_decompress_batch(_p(buf), _p(ro), _p(rl), C.c_uint32(n), _p(out), _p(out_off), _p(cap),
                                  _p(out_len), _p(status), C.c_uint32(grid))
    parts = [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(n) if status[i] == 0]
    return b"".join(parts), status


def snappy_compress(data: bytes, chunk: int, grid: int = 3):
    """Emulated gpumt_snappy_compress_batch + compact -> the snappy-mt stream."""
    L = lib()
    L.emu_snappy_slot_stride.restype = C.c_size_t
    L.emu_snappy_slot_stride.argtypes = [C.c_size_t]
    n = len(data)
    nrec = max(1, (n + chunk - 1) // chunk)
    stride = L.emu_snappy_slot_stride(chunk)
    inp = np.frombuffer(data + b"\0" * 64, np.uint8).copy()
    slots = np.full(nrec * stride, 0xEE, np.uint8)
    rec_len = np.full(nrec, 0xA5A5A5A5, np.uint32)
    L.emu_snappy_compress_batch(_p(inp), C.c_uint64(n), C.c_uint32(chunk), _p(slots), C.c_uint64(stride),
                                _p(rec_len), C.c_uint32(grid))
    return b"".join(slots[i * stride:i * stride + int(rec_len[i])].tobytes() for i in range(nrec))
