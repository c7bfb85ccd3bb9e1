"""Thin object wrapper over the gpumt_* C ABI (include/gpumt.h) for tests and bench.py.

Device buffers are plain integers (device pointers) owned by the Engine; numpy arrays are only
used to stage host data.  Every call that fails raises NativeError with the HIP error text.
"""
import ctypes as C

import numpy as np

from ._native import NativeError, lib

ST_NAMES = {0: "ok", 1: "bad_record", 2: "bad_frame", 3: "bad_block", 4: "size_mismatch",
            5: "bad_checksum", 6: "trailing", 7: "unsupported"}


class DevBuf:
    __slots__ = ("ptr", "nbytes", "eng")

    def __init__(self, eng, nbytes):
        self.eng = eng
        self.nbytes = int(nbytes)
        self.ptr = lib().gpumt_malloc(eng.h, self.nbytes)
        if not self.ptr:
            raise NativeError(f"gpumt_malloc({nbytes}) failed")

    def free(self):
        if self.ptr:
            lib().gpumt_device_sync(self.eng.h)   # gpumt_free wants an idle buffer (it may be cached and reused)
            lib().gpumt_free(self.eng.h, self.ptr)
            self.ptr = None

    def at(self, byte_off):
        return C.c_void_p(self.ptr + int(byte_off))


class Engine:
    """One HIP device. `Engine(0)` raises NativeError when no gfx950 GPU is usable."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.gpumt_open(device, C.byref(h))
        if rc != 0:
            raise NativeError(f"gpumt_open({device}) failed with {rc} (no gfx950 device / HIP runtime)")
        self.h = h
        self.name = self.L.gpumt_device_name(h).decode()

    def close(self):
        if self.h:
            self.L.gpumt_close(self.h)
            self.h = None

    # ---- helpers ---------------------------------------------------------------------------
    def _ck(self, rc, what):
        if rc != 0:
            raise NativeError(f"{what} -> {rc}: {self.L.gpumt_last_error(self.h).decode()}")

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def upload(self, arr, stream=0, slack=256):
        """numpy array / bytes -> new device buffer (+slack: 8-byte hash reads of the encoder and the
        128-byte line fetches of the decoder's parse kernel stay inside the allocation)"""
        a = np.frombuffer(arr, np.uint8) if isinstance(arr, (bytes, bytearray)) else arr
        a = np.ascontiguousarray(a)
        d = self.alloc(a.nbytes + slack)
        if a.nbytes:
            self._ck(self.L.gpumt_memcpy_h2d(self.h, d.ptr, a.ctypes.data, a.nbytes, stream), "h2d")
            self.sync(stream)
        return d

    def download(self, d, nbytes, dtype=np.uint8, offset=0, stream=0):
        out = np.empty(int(nbytes), np.uint8)
        if nbytes:
            self._ck(self.L.gpumt_memcpy_d2h(self.h, out.ctypes.data, d.ptr + int(offset), int(nbytes),
                                             stream), "d2h")
            self.sync(stream)
        return out.view(dtype)

    def equal(self, d_a, d_b, nbytes, piece=128 << 20):
        """byte-for-byte comparison of two device buffers: both are copied to pinned host memory in
        pieces and compared there (bench.py's round-trip verification)"""
        nbytes = int(nbytes)
        if nbytes == 0:
            return True
        piece = min(piece, nbytes)
        self.L.gpumt_host_alloc.restype = C.c_void_p
        ha = self.L.gpumt_host_alloc(self.h, piece)
        hb = self.L.gpumt_host_alloc(self.h, piece)
        if not ha or not hb:
            raise NativeError("gpumt_host_alloc failed")
        libc = C.CDLL(None)
        libc.memcmp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        same = True
        try:
            for off in range(0, nbytes, piece):
                m = min(piece, nbytes - off)
                self._ck(self.L.gpumt_memcpy_d2h(self.h, C.c_void_p(ha), d_a.ptr + off, m, 2), "d2h")
                self._ck(self.L.gpumt_memcpy_d2h(self.h, C.c_void_p(hb), d_b.ptr + off, m, 3), "d2h")
                self.sync(2)
                self.sync(3)
                if libc.memcmp(ha, hb, m) != 0:
                    same = False
                    break
        finally:
            self.L.gpumt_host_free(self.h, C.c_void_p(ha))
            self.L.gpumt_host_free(self.h, C.c_void_p(hb))
        return same

    def replicas_equal(self, d_buf, base_n, reps, piece=1 << 20, stream=0):
        """every one of `reps` back-to-back replicas of `base_n` bytes in d_buf equals replica 0: XXH32 of each `piece`
        on the device (gpumt_xxh32_batch, one pass at HBM speed), the hash rows compared on the host -- bench.py's check
        of the replicated decode legs beyond the first replica, which is compared with the text byte for byte"""
        base_n, reps = int(base_n), int(reps)
        if reps <= 1 or base_n == 0:
            return True
        per = (base_n + piece - 1) // piece
        off1 = np.arange(per, dtype=np.uint64) * np.uint64(piece)
        len1 = np.full(per, piece, np.uint32)
        len1[-1] = base_n - (per - 1) * piece
        off = np.concatenate([off1 + np.uint64(r * base_n) for r in range(reps)])
        d_off, d_len = self.upload(off, stream), self.upload(np.tile(len1, reps), stream)
        d_hash = self.alloc(per * reps * 4)
        try:
            self._ck(self.L.gpumt_xxh32_batch(self.h, d_buf.ptr, d_off.ptr, d_len.ptr, per * reps, d_hash.ptr, stream),
                     "xxh32_batch")
            hh = self.download(d_hash, per * reps * 4, np.uint32, stream=stream).reshape(reps, per)
        finally:
            for b in (d_off, d_len, d_hash):
                b.free()
        return bool((hh == hh[0]).all())

    def sync(self, stream=None):
        if stream is None:
            self._ck(self.L.gpumt_device_sync(self.h), "device_sync")
        else:
            self._ck(self.L.gpumt_stream_sync(self.h, stream), "stream_sync")

    def timer_start(self, slot, stream=0):
        self._ck(self.L.gpumt_timer_start(self.h, slot, stream), "timer_start")

    def timer_stop(self, slot, stream=0):
        self._ck(self.L.gpumt_timer_stop(self.h, slot, stream), "timer_stop")

    def timer_ms(self, slot):
        ms = C.c_float()
        self._ck(self.L.gpumt_timer_ms(self.h, slot, C.byref(ms)), "timer_ms")
        return ms.value

    def set_variant(self, what, v):
        return self.L.gpumt_set_variant(self.h, what.encode(), v)

    # ---- LZ4, device-resident ----------------------------------------------------------------
    def slot_stride(self, chunk):
        return self.L.gpumt_lz4_slot_stride(chunk)

    def record_count(self, n, chunk):
        return self.L.gpumt_lz4_record_count(n, chunk)

    def lz4_compress(self, d_in, n, chunk, d_slots, stride, d_rec_len, stream=0, level=1):
        self._ck(self.L.gpumt_lz4_compress_batch_level(self.h, d_in.ptr, n, chunk, d_slots.ptr, stride,
                                                       d_rec_len.ptr, level, stream),
                 "lz4_compress_batch_level")

    def lz4_compact(self, d_slots, stride, d_rec_len, nrec, d_stream, d_rec_off, stream=0):
        self._ck(self.L.gpumt_lz4_compact(self.h, d_slots.ptr, stride, d_rec_len.ptr, nrec,
                                          d_stream.ptr, d_rec_off.ptr, stream), "lz4_compact")

    def lz4_probe(self, d_stream, d_rec_off, d_rec_len, nrec, d_out_len, d_out_off, stream=0):
        self._ck(self.L.gpumt_lz4_probe_sizes(self.h, d_stream.ptr, d_rec_off.ptr, d_rec_len.ptr, nrec,
                                              d_out_len.ptr, d_out_off.ptr, stream), "lz4_probe_sizes")

    def lz4_decompress(self, d_stream, stream_bytes, d_rec_off, d_rec_len, nrec, d_out, out_bytes,
                       d_out_off, d_out_len, d_status, stream=0):
        self._ck(self.L.gpumt_lz4_decompress_batch(self.h, d_stream.ptr, int(stream_bytes),
                                                   d_rec_off.ptr, d_rec_len.ptr, nrec, d_out.ptr,
                                                   int(out_bytes), d_out_off.ptr, d_out_len.ptr,
                                                   d_status.ptr, stream), "lz4_decompress_batch")

    def zstd_slot_stride(self, chunk):
        return int(self.L.gpumt_zstd_slot_stride(chunk))

    def zstd_compress(self, d_in, n, chunk, d_slots, stride, d_rec_len, stream=0, level=1):
        self._ck(self.L.gpumt_zstd_compress_batch_level(self.h, d_in.ptr, int(n), int(chunk), d_slots.ptr,
                                                        int(stride), d_rec_len.ptr, int(level), stream),
                 "zstd_compress_batch_level")

    def zstd_probe(self, d_stream, d_rec_off, d_rec_len, nrec, d_out_len, d_out_off, d_status, stream=0):
        self._ck(self.L.gpumt_zstd_probe_sizes(self.h, d_stream.ptr, d_rec_off.ptr, d_rec_len.ptr, nrec,
                                               d_out_len.ptr, d_out_off.ptr, d_status.ptr, stream),
                 "zstd_probe_sizes")

    def zstd_decompress(self, d_stream, stream_bytes, d_rec_off, d_rec_len, nrec, d_out, out_bytes,
                        d_out_off, d_out_len, d_status, stream=0):
        self._ck(self.L.gpumt_zstd_decompress_batch(self.h, d_stream.ptr, int(stream_bytes),
                                                    d_rec_off.ptr, d_rec_len.ptr, nrec, d_out.ptr,
                                                    int(out_bytes), d_out_off.ptr, d_out_len.ptr,
                                                    d_status.ptr, stream), "zstd_decompress_batch")

    def brotli_compress(self, d_in, n, chunk, d_slots, stride, d_rec_len, stream=0, level=1):
        self._ck(self.L.gpumt_brotli_compress_batch_level(self.h, d_in.ptr, int(n), int(chunk), d_slots.ptr,
                                                          int(stride), d_rec_len.ptr, int(level), stream),
                 "brotli_compress_batch_level")

    def brotli_decompress(self, d_stream, d_rec_off, d_rec_len, nrec, d_out, d_out_off, d_out_cap, d_out_len,
                          d_status, stream=0):
        self._ck(self.L.gpumt_brotli_decompress_batch(self.h, d_stream.ptr, d_rec_off.ptr, d_rec_len.ptr, nrec,
                                                      d_out.ptr, d_out_off.ptr, d_out_cap.ptr, d_out_len.ptr,
                                                      d_status.ptr, stream), "brotli_decompress_batch")

    # snappy-mt (include/gpumt.h gpumt_snappy_*): same call shapes as brotli
    def snappy_slot_stride(self, chunk):
        return int(self.L.gpumt_snappy_slot_stride(int(chunk)))

    def snappy_compress(self, d_in, n, chunk, d_slots, stride, d_rec_len, stream=0):
        self._ck(self.L.gpumt_snappy_compress_batch(self.h, d_in.ptr, int(n), int(chunk), d_slots.ptr,
                                                    int(stride), d_rec_len.ptr, stream), "snappy_compress_batch")

    def snappy_decompress(self, d_stream, d_rec_off, d_rec_len, nrec, d_out, d_out_off, d_out_cap, d_out_len,
                          d_status, stream=0):
        self._ck(self.L.gpumt_snappy_decompress_batch(self.h, d_stream.ptr, d_rec_off.ptr, d_rec_len.ptr, nrec,
                                                      d_out.ptr, d_out_off.ptr, d_out_cap.ptr, d_out_len.ptr,
                                                      d_status.ptr, stream), "snappy_decompress_batch")

    def brotli_decompress_bytes(self, stream: bytes, rec_off, rec_len, cap):
        """Records of a brotli-mt stream (payload offsets / sizes / capacities as the host engine
        parses them) -> (list of decoded records, status[n])"""
        nrec = len(rec_off)
        out_off = np.zeros(nrec + 1, np.uint64)
        out_off[1:] = np.cumsum(np.asarray(cap, np.uint64))
        total = int(out_off[nrec])
        d_stream = self.upload(stream + b"\0" * 256)
        d_ro, d_rl = self.upload(np.asarray(rec_off, np.uint64).copy()), self.upload(np.asarray(rec_len, np.uint32).copy())
        d_oo, d_oc = self.upload(out_off), self.upload(np.asarray(cap, np.uint32).copy())
        d_ol, d_st = self.alloc(nrec * 4), self.alloc(nrec * 4)
        d_out = self.upload(np.full(total + 64, 0xCC, np.uint8))
        try:
            self.brotli_decompress(d_stream, d_ro, d_rl, nrec, d_out, d_oo, d_oc, d_ol, d_st)
            status = self.download(d_st, nrec * 4, np.uint32)
            out_len = self.download(d_ol, nrec * 4, np.uint32)
            raw = self.download(d_out, total + 64)
        finally:
            for b in (d_stream, d_ro, d_rl, d_oo, d_oc, d_ol, d_st, d_out):
                b.free()
        assert (raw[total:] == 0xCC).all(), "decoder wrote past the end of its output"
        recs = []
        for i in range(nrec):
            o, n = int(out_off[i]), int(out_len[i])
            assert n <= int(cap[i])
            if status[i] == 0:
                assert (raw[o + n:o + int(cap[i])] == 0xCC).all(), "wrote past the decoded size"
            recs.append(raw[o:o + n].tobytes())
        return recs, status

    # ---- convenience round trips on host bytes (tests) ----------------------------------------
    def compress_bytes(self, data: bytes, chunk: int, codec="lz4", level=1):
        """-> (stream bytes, rec_off[n+1] u64, rec_len[n] u32)"""
        n = len(data)
        nrec = self.record_count(n, chunk)
        stride = self.zstd_slot_stride(chunk) if codec in ("zstd", "brotli") else self.slot_stride(chunk)
        d_in = self.upload(data)
        d_slots = self.alloc(nrec * stride)
        d_len = self.alloc(nrec * 4)
        d_off = self.alloc((nrec + 1) * 8)
        try:
            if codec == "zstd":
                self.zstd_compress(d_in, n, chunk, d_slots, stride, d_len, level=level)
            elif codec == "brotli":
                self.brotli_compress(d_in, n, chunk, d_slots, stride, d_len, level=level)
            else:
                self.lz4_compress(d_in, n, chunk, d_slots, stride, d_len, level=level)
            rec_len = self.download(d_len, nrec * 4, np.uint32)
            total = int(rec_len.astype(np.uint64).sum())
            d_stream = self.alloc(total + 64)
            try:
                self.lz4_compact(d_slots, stride, d_len, nrec, d_stream, d_off)
                rec_off = self.download(d_off, (nrec + 1) * 8, np.uint64)
                assert int(rec_off[nrec]) == total
                stream = self.download(d_stream, total).tobytes()
            finally:
                d_stream.free()
        finally:
            for b in (d_in, d_slots, d_len, d_off):
                b.free()
        return stream, rec_off, rec_len

    def decompress_bytes(self, stream: bytes, rec_off, rec_len, codec="lz4"):
        """-> (content bytes, status[n])"""
        nrec = len(rec_len)
        d_stream = self.upload(stream)
        d_ro = self.upload(np.asarray(rec_off, np.uint64)[:nrec].copy())
        d_rl = self.upload(np.asarray(rec_len, np.uint32).copy())
        d_ol = self.alloc(nrec * 4)
        d_oo = self.alloc((nrec + 1) * 8)
        d_st = self.alloc(nrec * 4)
        try:
            if codec == "zstd":
                self.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
            else:
                self.lz4_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo)
            out_off = self.download(d_oo, (nrec + 1) * 8, np.uint64)
            total = int(out_off[nrec])
            d_out = self.alloc(total + 64)
            try:
                lib().gpumt_memset(self.h, d_out.ptr, 0xCC, total + 64, 0)
                if codec == "zstd":
                    self.zstd_decompress(d_stream, len(stream), d_ro, d_rl, nrec, d_out, total, d_oo, d_ol, d_st)
                else:
                    self.lz4_decompress(d_stream, len(stream), d_ro, d_rl, nrec, d_out, total, d_oo, d_ol, d_st)
                status = self.download(d_st, nrec * 4, np.uint32)
                raw = self.download(d_out, total + 64)
                assert (raw[total:] == 0xCC).all(), "decoder wrote past the end of its output"
                out = raw[:total].tobytes()
            finally:
                d_out.free()
        finally:
            for b in (d_stream, d_ro, d_rl, d_ol, d_oo, d_st):
                b.free()
        return out, status
