"""Boundary check, build container only: the REFERENCE's own callers -- programs/lz4-mt.c, zstd-mt.c and
brotli-mt.c, each of which `#include`s programs/main.c, plus programs/platform.c -- compiled IN PLACE from
/root/reference against this repo's include/*.h and linked to the emulated host library
(tests/emu/libzstdmt_emu_host.so: the product's host engines over the fiber-emulated device boundary).

"Compiles unchanged" is the drop-in claim of SURVEY 8b / INTEGRATION.md section 1: the binding macros of
/root/reference/programs/lz4-mt.c:13-43 must find every type, constant and function they name in our headers,
and main.c (do_compress :207, do_decompress :255, ReadData / WriteData :172-200) must run against the library.
Nothing of the reference is copied into the repo or shipped to the GPU box: the test is skipped where
/root/reference does not exist."""
import os
import subprocess

import pytest

import helpers as H
from golden import cases

REF = "/root/reference/programs"
EMU_DIR = os.path.join(H.ROOT, "tests", "emu")
ENV = dict(os.environ, GPUMT_BATCH_KB="256")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main.c")),
                                reason="reference tree not present (GPU box)")


@pytest.fixture(scope="module")
def tools(tmp_path_factory):
    H.locked_make(EMU_DIR, "libzstdmt_emu_host.so", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = tmp_path_factory.mktemp("refcli")
    exe = {}
    for codec in ("lz4", "zstd", "brotli"):
        out = str(d / f"{codec}-mt")
        # the reference's link line (programs/Makefile:201-225) with our headers and library in place of its own
        cmd = ["gcc", "-O1", "-Wall", "-DVERSION=\"reference-caller\"", "-I" + os.path.join(H.ROOT, "include"), "-I" + REF,
               os.path.join(REF, f"{codec}-mt.c"), os.path.join(REF, "platform.c"), "-o", out,
               "-L" + EMU_DIR, "-l:libzstdmt_emu_host.so", "-Wl,-rpath," + EMU_DIR, "-lpthread"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, f"{codec}-mt.c does not compile against include/: {p.stderr}"
        assert "implicit declaration" not in p.stderr and "incompatible" not in p.stderr, p.stderr
        exe[codec] = out
    return exe


def _run(args, data=None):
    p = subprocess.run(args, input=data, capture_output=True, timeout=600, env=ENV)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


@pytest.mark.parametrize("codec", ["lz4", "zstd", "brotli"])
def test_reference_caller_round_trips(tools, codec, tmp_path):
    data = cases.text(300000, 11) + cases.rnd(3000, 2) + bytes(4000)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    comp = _run([tools[codec], "-1", "-T2", "-c", str(f)])
    assert 0 < len(comp) < len(data)
    g = tmp_path / "c.bin"
    g.write_bytes(comp)
    assert _run([tools[codec], "-d", "-c", str(g)]) == data
    # stdin / stdout, the shape of the reference's own test loop (programs/Makefile:252-260)
    assert _run([tools[codec], "-d"], _run([tools[codec], "-z"], data)) == data


def test_reference_lz4_caller_writes_the_reference_stream(tools, tmp_path):
    """bit-exact: what the reference's main.c writes through our library is the oracle's stream (level 1, the
    CLI's 4 MiB default chunk does not bind at this size: one record per read of `-b 1` MiB)"""
    data = cases.text(1500000, 5)
    f = tmp_path / "in.bin"
    f.write_bytes(data)
    comp = _run([tools["lz4"], "-1", "-T3", "-b", "1", "-c", str(f)])
    assert comp == H.oracle_compress(data, 1 << 20)
