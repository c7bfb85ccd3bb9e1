"""Device plumbing the host engines rely on (include/gpumt.h): the push kernel that carries results to
pinned host memory, the process-wide buffer caches, and several contexts at work at the same time."""
import ctypes as C
import threading

import numpy as np
import pytest

import helpers as H
from cases import rnd, text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import zstdmt_amd as z
    e = z.Engine(0)
    yield e
    e.close()


def _host(eng, n):
    eng.L.gpumt_host_alloc.restype = C.c_void_p
    p = eng.L.gpumt_host_alloc(eng.h, n)
    assert p
    return p


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 4096, 1 << 20, (5 << 20) + 7])
def test_push_host_fixed_and_device_counted(eng, n):
    """gpumt_push_host: n bytes, or min(n, *d_n) with the count read on the device"""
    data = rnd(max(n, 1), 3)[:n]
    d = eng.upload(data + bytes(64))
    h = _host(eng, n + 64)
    C.memset(h, 0xEE, n + 64)
    eng._ck(eng.L.gpumt_push_host(eng.h, C.c_void_p(h), d.ptr, n, None, 5), "push")
    eng.sync(5)
    got = C.string_at(h, n + 16)
    assert got[:n] == data and got[n:] == b"\xEE" * 16
    if n >= 16:
        cnt = eng.upload(np.array([n - 9], np.uint64))
        C.memset(h, 0xEE, n + 64)
        eng._ck(eng.L.gpumt_push_host(eng.h, C.c_void_p(h), d.ptr, n, cnt.ptr, 6), "push")
        eng.sync(6)
        got = C.string_at(h, n)
        assert got[:n - 9] == data[:n - 9] and got[n - 9:] == b"\xEE" * 9
        cnt.free()
    # misaligned addresses are refused, not copied slowly
    assert eng.L.gpumt_push_host(eng.h, C.c_void_p(h + 4), d.ptr, 16, None, 5) != 0
    eng.L.gpumt_host_free(eng.h, C.c_void_p(h))
    d.free()


def test_freed_buffers_are_reused(eng):
    """device and pinned buffers come back from the process-wide caches (same address for the same size)"""
    a = eng.alloc(37 << 20)
    pa = a.ptr
    a.free()
    b = eng.alloc(37 << 20)
    assert b.ptr == pa
    b.free()
    h1 = _host(eng, 21 << 20)
    eng.L.gpumt_host_free(eng.h, C.c_void_p(h1))
    h2 = _host(eng, 21 << 20)
    assert h2 == h1
    eng.L.gpumt_host_free(eng.h, C.c_void_p(h2))


def test_trim_caches_gives_idle_buffers_back(eng):
    """gpumt_trim_caches: the cached (idle) device / pinned buffers are released and counted; a later allocation works"""
    eng.L.gpumt_trim_caches.restype = C.c_size_t
    eng.L.gpumt_trim_caches.argtypes = [C.c_void_p]
    eng.L.gpumt_trim_caches(eng.h)                     # whatever earlier tests left behind
    a = eng.alloc(41 << 20)
    a.free()
    h1 = _host(eng, 9 << 20)
    eng.L.gpumt_host_free(eng.h, C.c_void_p(h1))
    freed = eng.L.gpumt_trim_caches(eng.h)
    assert freed >= (41 << 20) + (9 << 20)
    assert eng.L.gpumt_trim_caches(eng.h) == 0         # nothing idle is left
    b = eng.alloc(41 << 20)
    b.free()


def test_host_register_pins_caller_memory(eng):
    """gpumt_host_register: memory the caller owns becomes a valid target of the push kernel (what the ranks of
    bench.py --gather d2h do with their shared mapping)"""
    import mmap
    n = 3 << 20
    m = mmap.mmap(-1, n)
    buf = (C.c_ubyte * n).from_buffer(m)
    eng.L.gpumt_host_register.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    eng.L.gpumt_host_unregister.argtypes = [C.c_void_p, C.c_void_p]
    assert eng.L.gpumt_host_register(eng.h, C.addressof(buf), n) == 0
    try:
        data = np.frombuffer(rnd(n, 77), np.uint8)
        d = eng.upload(data)
        eng.L.gpumt_push_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        assert eng.L.gpumt_push_host(eng.h, C.addressof(buf), C.c_void_p(d.ptr), n, None, 0) == 0
        eng.sync()
        assert bytes(buf) == data.tobytes()
        d.free()
    finally:
        assert eng.L.gpumt_host_unregister(eng.h, C.addressof(buf)) == 0
        del buf
        m.close()


def test_contexts_on_several_threads():
    """four callers, each with its own contexts, at the same time: the caches and the pipelines keep
    them apart (every stream equals the oracle's, every round trip its input)"""
    from zstdmt_amd._native import lib_path
    lib = H.bind_lz4mt(C.CDLL(lib_path()))
    datas = [text(3_000_000 + 70_001 * i, seed=40 + i) + rnd(50_000, i) for i in range(4)]
    errs = []

    def work(i):
        try:
            for rep in range(3):
                rv, stream, _, _ = H.lz4mt_compress_via(lib, datas[i], 131072, threads=2 + i, level=1)
                assert rv == 0 and stream == H.oracle_compress(datas[i], 131072)
                rv, out, _, _ = H.lz4mt_decompress_via(lib, stream, threads=2)
                assert rv == 0 and out == datas[i]
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_debug_free_guard_counts_busy_frees(eng):
    """GPUMT_DEBUG_FREE / gpumt_set_variant("debug_free", 1): a buffer freed while the context's streams still have work
    queued is counted and the streams are drained first (include/gpumt.h: "free a buffer only when no queued work uses
    it" -- a violation is a cross-context use-after-free through the caches); an idle free counts nothing"""
    eng.L.gpumt_debug_free_busy()
    prev = eng.set_variant("debug_free", 1)
    try:
        n = 256 << 20
        data = np.frombuffer(text(1 << 20, 3), np.uint8)
        d_in = eng.upload(np.tile(data, n >> 20))
        stride = eng.slot_stride(131072)
        nrec = n // 131072
        d_slots, d_rl = eng.alloc(nrec * stride), eng.alloc(nrec * 4)
        victim = eng.alloc(1 << 20)
        eng.sync()
        eng.lz4_compress(d_in, n, 131072, d_slots, stride, d_rl)      # ~10 ms of kernels queued
        # (DevBuf.free() waits for the device first, as the contract wants: the raw call is what breaks it)
        eng.L.gpumt_free(eng.h, C.c_void_p(victim.ptr))                # busy: counted, drained
        victim.ptr = None
        assert eng.L.gpumt_debug_free_busy() == 1
        other = eng.alloc(1 << 20)
        eng.sync()
        eng.L.gpumt_free(eng.h, C.c_void_p(other.ptr))                 # idle: not counted
        other.ptr = None
        assert eng.L.gpumt_debug_free_busy() == 0
        for b in (d_in, d_slots, d_rl):
            b.free()
    finally:
        eng.set_variant("debug_free", prev)


def test_host_engines_never_free_busy_buffers(tmp_path):
    """the drop-in engines under the guard: a round trip through LZ4MT_* / ZSTDCB_* with GPUMT_DEBUG_FREE=1 reports no
    busy free on stderr"""
    import os
    import subprocess
    exe = os.path.join(H.ROOT, "zstdmt_amd", "bin", "api_bench")
    lib = os.path.join(H.ROOT, "zstdmt_amd", "lib", "libzstdmt_amd.so")
    for codec, chunk in (("lz4", 131072), ("zstd", 1 << 20)):
        p = subprocess.run([exe, codec, str(256 << 20), str(chunk), lib, "1"], capture_output=True, timeout=300,
                           env=dict(os.environ, GPUMT_DEBUG_FREE="1"))
        assert p.returncode == 0, p.stderr[-300:]
        assert b"free guard on" in p.stderr, "the guard was not enabled through the environment: " + repr(p.stderr[-300:])
        assert b"with work queued" not in p.stderr, p.stderr[-500:]
