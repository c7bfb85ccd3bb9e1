"""zstdmt_amd -- MI355X-native block-parallel (de)compression engine behind the zstdmt (lz4-mt) API.

The product is the native library zstdmt_amd/lib/libzstdmt_amd.so (HIP kernels for gfx950 + the
C ABI of include/gpumt.h and include/lz4-mt.h).  This package is only the thin ctypes binding used
by the tests and bench.py; it never computes anything itself and has no CPU fallback.
"""
from ._native import NativeError, lib, lib_path  # noqa: F401
from .device import Engine  # noqa: F401
