"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU compute here)."""
import ctypes as C
import os
import re

import pytest

import helpers as H


def _declared(header):
    txt = open(os.path.join(H.ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*#\s*define[^\n]*$", "", txt, flags=re.M)   # function-like macros are not symbols
    return sorted(set(re.findall(r"\b((?:gpumt|LZ4MT|ZSTDCB|BROTLIMT|SNAPPYMT)_[A-Za-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def native():
    from zstdmt_amd._native import lib_path
    if not os.path.exists(lib_path()):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(lib_path())


@pytest.mark.parametrize("header", [h for h in ("gpumt.h", "lz4-mt.h", "zstd-mt.h", "brotli-mt.h", "snappy-mt.h")
                                    if os.path.exists(os.path.join(H.ROOT, "include", h))])
def test_exports(native, header):
    names = _declared(header)
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(native, n)]
    assert not missing, missing


def test_no_libzstd_symbol_collision(native):
    """libzstd exports ZSTDMT_* itself (its own multithreading API): this library must not."""
    out = __import__("subprocess").check_output(["nm", "-D", "--defined-only", native._name], text=True)
    assert not re.findall(r"\bZSTDMT_\w+", out)
    # the opt-in header names are macros over the exported ZSTDCB_* entry points
    txt = open(os.path.join(H.ROOT, "include", "zstd-mt.h")).read()
    for m in re.findall(r"#define (ZSTDMT_[A-Za-z]+) (ZSTDCB_[A-Za-z]+)", txt):
        assert m[0][7:] == m[1][7:] and (m[1].endswith("_MAX") or m[1].endswith("_MIN") or hasattr(native, m[1])), m


def test_binding_table_matches_header(native):
    from zstdmt_amd._native import GPUMT_SYMBOLS
    assert sorted(GPUMT_SYMBOLS) == _declared("gpumt.h")


def test_pure_helpers(native):
    native.gpumt_lz4_slot_stride.restype = C.c_size_t
    native.gpumt_lz4_slot_stride.argtypes = [C.c_size_t]
    native.gpumt_lz4_record_count.restype = C.c_size_t
    native.gpumt_lz4_record_count.argtypes = [C.c_size_t, C.c_size_t]
    # 12 + LZ4F_compressFrameBound(chunk), rounded to 256
    assert native.gpumt_lz4_slot_stride(131072) == (131119 + 255) // 256 * 256
    assert native.gpumt_lz4_slot_stride(4 << 20) == (4194599 + 255) // 256 * 256
    assert native.gpumt_lz4_record_count(0, 131072) == 1
    assert native.gpumt_lz4_record_count(131072, 131072) == 1
    assert native.gpumt_lz4_record_count(131073, 131072) == 2


def test_no_device_fails_loudly(native):
    """Without a GPU the engine must refuse to open -- never fall back to a CPU path."""
    native.gpumt_device_count.restype = C.c_int
    if native.gpumt_device_count() > 0:
        pytest.skip("a GPU is present")
    import zstdmt_amd as z
    with pytest.raises(z.NativeError):
        z.Engine(0)


def test_bind_to_node_only_narrows_the_threads_mask(native):
    """mt_pipe.c's mt_bind_to_node (the reader / writer threads of the host engines move next to the device's pinned memory):
    never widens the calling thread's CPU mask, does nothing for an unknown node or with GPUMT_NUMA=0"""
    import os, threading
    out = {}

    def run():
        before = os.sched_getaffinity(0)
        native.mt_bind_to_node.restype = C.c_int
        native.mt_bind_to_node.argtypes = [C.c_int]
        out["none"] = native.mt_bind_to_node(-1), native.mt_bind_to_node(4095)
        os.environ["GPUMT_NUMA"] = "0"
        out["off"] = native.mt_bind_to_node(0)
        del os.environ["GPUMT_NUMA"]
        out["rv"] = native.mt_bind_to_node(0)
        out["before"], out["after"] = before, os.sched_getaffinity(0)

    t = threading.Thread(target=run)
    t.start()
    t.join()
    assert out["none"] == (0, 0) and out["off"] == 0
    assert out["after"] <= out["before"] and len(out["after"]) > 0
    assert (out["rv"] == 1) == (out["after"] != out["before"])
