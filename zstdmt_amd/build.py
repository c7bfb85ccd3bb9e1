"""Build the in-tree native libraries.

  libzstdmt_amd.so  hipcc --offload-arch=gfx950: HIP kernels + C-ABI shim (include/gpumt.h) and the
                    plain-C host engine behind the reference API (include/lz4-mt.h)
  libzmt_tools.so   gcc: synthetic-input generators used by bench.py and the tests

hipcc cross-compiles without a GPU; the .so files are git-ignored but travel with gpurun.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")

HIP_SRCS = ["xxh32.hip", "snappy.hip", "lz4_enc3.hip", "lz4_enc5.hip", "lz4_enc_hc.hip", "lz4_dec.hip", "lz4_dec_split.hip", "lz4_dec_parse4.hip", "lz4_dec_copy3.hip", "zstd_dec.hip", "zstd_dec_seq.hip", "zstd_enc.hip", "brotli_dec.hip", "brotli_dec4.hip", "brotli_enc.hip", "pack.hip", "gpumt.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
HIPFLAGS += os.environ.get("ZMT_HIPFLAGS", "").split()   # developer: -D overrides for A/B builds
CFLAGS = ["-O2", "-g", "-std=gnu11", "-fPIC", "-pthread", "-Wall", "-Wextra",
          "-I" + os.path.join(ROOT, "include")]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipdir = os.path.join(CSRC, "hip")
    hostdir = os.path.join(CSRC, "host")
    headers = [os.path.join(hipdir, h) for h in os.listdir(hipdir) if h.endswith(".h")]
    headers += [os.path.join(ROOT, "include", h) for h in os.listdir(os.path.join(ROOT, "include"))]
    objs = []
    for src in HIP_SRCS:
        s = os.path.join(hipdir, src)
        o = os.path.join(OBJDIR, src + ".o")
        if force or _stale(o, [s] + headers):
            _run([HIPCC] + HIPFLAGS + ["-c", s, "-o", o])
        objs.append(o)
    if os.path.isdir(hostdir):
        for src in sorted(os.listdir(hostdir)):
            if not src.endswith(".c"):
                continue
            s = os.path.join(hostdir, src)
            o = os.path.join(OBJDIR, src + ".o")
            hh = [os.path.join(hostdir, h) for h in os.listdir(hostdir) if h.endswith((".h", ".inc"))]
            datadir = os.path.join(CSRC, "data")
            hh += [os.path.join(datadir, d) for d in os.listdir(datadir)]
            if force or _stale(o, [s] + headers + hh):
                _run(["gcc"] + CFLAGS + ["-Wa,-I" + datadir, "-c", s, "-o", o])
            objs.append(o)
    lib = os.path.join(LIBDIR, "libzstdmt_amd.so")
    if force or _stale(lib, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"])
    tools = os.path.join(LIBDIR, "libzmt_tools.so")
    tsrc = os.path.join(CSRC, "tools", "gen_text.c")
    if force or _stale(tools, [tsrc]):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-pthread", tsrc, "-o", tools, "-lm"])
    # command line front ends (programs/zmt_cli.c): lz4-mt, zstd-mt, brotli-mt, snappy-mt + their un* / *cat personalities
    bindir = os.path.join(HERE, "bin")
    os.makedirs(bindir, exist_ok=True)
    cli = os.path.join(ROOT, "programs", "zmt_cli.c")
    for name, flags, links in (("lz4-mt", [], ("unlz4-mt", "lz4cat-mt")),
                               ("zstd-mt", ["-DZMT_ZSTD"], ("unzstd-mt", "zstdcat-mt")),
                               ("brotli-mt", ["-DZMT_BROTLI"], ("unbrotli-mt", "brotlicat-mt")),
                               ("snappy-mt", ["-DZMT_SNAPPY"], ("unsnappy-mt", "snappycat-mt"))):
        exe = os.path.join(bindir, name)
        if force or _stale(exe, [cli, lib] + headers):
            _run(["gcc", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include")] + flags +
                 [cli, "-o", exe, "-L" + LIBDIR, "-lzstdmt_amd", "-Wl,-rpath,$ORIGIN/../lib"])
        for ln in links:
            dst = os.path.join(bindir, ln)
            if not os.path.lexists(dst):
                os.symlink(name, dst)
    # PCIe-inclusive API benchmark (tools/api_bench.c): bench.py's "drop-in API" leg
    ab = os.path.join(bindir, "api_bench")
    absrc = os.path.join(ROOT, "tools", "api_bench.c")
    if force or _stale(ab, [absrc, tsrc]):
        _run(["gcc", "-O2", "-pthread", absrc, tsrc, "-o", ab, "-ldl", "-lm"])
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
