"""ctypes loader for libzstdmt_amd.so.  Fails loudly when the library is missing."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class NativeError(RuntimeError):
    pass


def lib_path():
    # ZMT_LIB: developer override (A/B builds of the kernels, tools/variant_build.sh)
    return os.environ.get("ZMT_LIB") or os.path.join(HERE, "lib", "libzstdmt_amd.so")


# name -> (restype, argtypes)   -- every symbol include/gpumt.h declares
_vp, _sz, _i, _u32p, _u64p = C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p
GPUMT_SYMBOLS = {
    "gpumt_device_count": (_i, []),
    "gpumt_open": (_i, [_i, C.POINTER(_vp)]),
    "gpumt_close": (None, [_vp]),
    "gpumt_last_error": (C.c_char_p, [_vp]),
    "gpumt_device_name": (C.c_char_p, [_vp]),
    "gpumt_host_node": (C.c_int, [_vp]),
    "gpumt_malloc": (_vp, [_vp, _sz]),
    "gpumt_free": (None, [_vp, _vp]),
    "gpumt_host_alloc": (_vp, [_vp, _sz]),
    "gpumt_host_free": (None, [_vp, _vp]),
    "gpumt_host_register": (_i, [_vp, _vp, _sz]),
    "gpumt_host_unregister": (_i, [_vp, _vp]),
    "gpumt_trim_caches": (_sz, [_vp]),
    "gpumt_debug_free_busy": (C.c_ulong, []),
    "gpumt_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz, _i]),
    "gpumt_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz, _i]),
    "gpumt_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz, _i]),
    "gpumt_push_host": (_i, [_vp, _vp, _vp, _sz, _vp, _i]),
    "gpumt_memset": (_i, [_vp, _vp, _i, _sz, _i]),
    "gpumt_stream_sync": (_i, [_vp, _i]),
    "gpumt_device_sync": (_i, [_vp]),
    "gpumt_stream_wait": (_i, [_vp, _i, _i]),
    "gpumt_mark": (_i, [_vp, _i, _i]),
    "gpumt_mark_sync": (_i, [_vp, _i]),
    "gpumt_stream_handle": (_vp, [_vp, _i]),
    "gpumt_timer_start": (_i, [_vp, _i, _i]),
    "gpumt_timer_stop": (_i, [_vp, _i, _i]),
    "gpumt_timer_ms": (_i, [_vp, _i, C.POINTER(C.c_float)]),
    "gpumt_lz4_slot_stride": (_sz, [_sz]),
    "gpumt_lz4_record_count": (_sz, [_sz, _sz]),
    "gpumt_lz4_compress_batch": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _u32p, _i]),
    "gpumt_lz4_compress_batch_level": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _u32p, _i, _i]),
    "gpumt_lz4_level_supported": (_i, [_i]),
    "gpumt_lz4_compact": (_i, [_vp, _vp, _sz, _u32p, _sz, _vp, _u64p, _i]),
    "gpumt_lz4_probe_sizes": (_i, [_vp, _vp, _u64p, _u32p, _sz, _u32p, _u64p, _i]),
    "gpumt_lz4_decompress_batch": (_i, [_vp, _vp, _sz, _u64p, _u32p, _sz, _vp, _sz, _u64p, _u32p, _u32p, _i]),
    "gpumt_zstd_slot_stride": (_sz, [_sz]),
    "gpumt_zstd_compress_batch": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _u32p, _i]),
    "gpumt_zstd_compress_batch_level": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _u32p, _i, _i]),
    "gpumt_zstd_level_tier": (_i, [_i]),
    "gpumt_zstd_probe_sizes": (_i, [_vp, _vp, _u64p, _u32p, _sz, _u32p, _u64p, _u32p, _i]),
    "gpumt_zstd_decompress_batch": (_i, [_vp, _vp, _sz, _u64p, _u32p, _sz, _vp, _sz, _u64p, _u32p, _u32p, _i]),
    "gpumt_brotli_compress_batch": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _u32p, _i]),
    "gpumt_brotli_compress_batch_level": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _u32p, _i, _i]),
    "gpumt_brotli_level_tier": (_i, [_i]),
    "gpumt_brotli_decompress_batch": (_i, [_vp, _vp, _u64p, _u32p, _sz, _vp, _u64p, _u32p, _u32p, _u32p, _i]),
    "gpumt_snappy_slot_stride": (_sz, [_sz]),
    "gpumt_snappy_compress_batch": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _u32p, _i]),
    "gpumt_snappy_decompress_batch": (_i, [_vp, _vp, _u64p, _u32p, _sz, _vp, _u64p, _u32p, _u32p, _u32p, _i]),
    "gpumt_xxh32_batch": (_i, [_vp, _vp, _u64p, _u32p, _sz, _u32p, _i]),
    "gpumt_set_variant": (_i, [_vp, C.c_char_p, _i]),
    "gpumt_debug_counters": (_i, [_vp, _vp, _i]),
}


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise NativeError(
                f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            L = C.CDLL(p)
        except OSError as e:
            raise NativeError(f"cannot load {p}: {e}") from e
        for name, (res, args) in GPUMT_SYMBOLS.items():
            f = getattr(L, name)  # AttributeError = ABI drift, let it surface
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB
