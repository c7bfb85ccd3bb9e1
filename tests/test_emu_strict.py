"""The kernels once more under the emulator's strict build (-DZMT_EMU_STRICT): wv_readfirst exchanges for
real there, so a value claimed wave-uniform that is not, or a cross-lane operation under a divergent
branch -- both silent on the fast build, the second one a hang on hardware at best -- aborts the run.
A representative subset of the emulator tests of every kernel family, in a process of its own (the
emulator library is chosen at load time)."""
import os
import subprocess
import sys

import helpers as H

SUBSET = [
    ("tests/test_emu_kernels.py", "golden or hc_compress"),
    ("tests/test_emu_zstd.py", "identical or checksummed or unit"),
    ("tests/test_emu_brotli.py", "golden or identical"),
    ("tests/test_emu_snappy.py", "identical or golden or hand_built or rejects"),
]


import pytest


@pytest.mark.parametrize("order", ["2"])   # "1" (plain reverse) passes too; one of the two keeps the suite short
def test_kernels_with_lanes_in_another_order(order):
    """EMU_REVERSE=1: between two synchronisation points the lanes run from the highest down, so a lane sees
    the stores of the lanes above it instead of those below; EMU_REVERSE=2: in a fresh pseudo-random order
    every round.  Code that is right every way does not depend on which neighbour happened to run first (on
    hardware: on what the memory system happens to have done)."""
    env = dict(os.environ, EMU_REVERSE=order)
    for path, expr in SUBSET:
        cmd = [sys.executable, "-m", "pytest", os.path.join(H.ROOT, path), "-q", "-x", "-p", "no:cacheprovider"]
        if expr:
            cmd += ["-k", expr]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=H.ROOT)
        assert p.returncode == 0, f"{path} under EMU_REVERSE={order}:\n{(p.stdout + p.stderr)[-1500:]}"


def test_kernels_under_the_strict_emulator():
    env = dict(os.environ, EMU_STRICT="1")
    for path, expr in SUBSET:
        cmd = [sys.executable, "-m", "pytest", os.path.join(H.ROOT, path), "-q", "-x", "-p", "no:cacheprovider"]
        if expr:
            cmd += ["-k", expr]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=H.ROOT)
        tail = (p.stdout + p.stderr)[-1500:]
        assert p.returncode == 0, f"{path} under EMU_STRICT=1:\n{tail}"
