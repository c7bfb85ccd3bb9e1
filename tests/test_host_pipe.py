"""The host engines' three-role batch pipeline (zstdmt_amd/csrc/host/mt_pipe.c) on the CPU, with synthetic
roles (tests/host/pipe_harness.c): batches leave strictly in order, a slot is never refilled before it
was drained, every slot count works, the inline (threads == 1) variant does the same on one thread, and an
error in any role at any batch stops the pipeline and is what the call returns."""
import os
import subprocess

import pytest

import helpers as H

HOST = os.path.join(H.ROOT, "zstdmt_amd", "csrc", "host")
EXE = os.path.join(H.ROOT, "tests", "host", "pipe_harness")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["gcc", "-O1", "-g", "-pthread", "-Wall", "-Wextra", "-I" + HOST,
                           os.path.join(H.ROOT, "tests", "host", "pipe_harness.c"), os.path.join(HOST, "mt_pipe.c"),
                           "-o", EXE])
    return EXE


def run(exe, nbatch, nslot, inline=0, fail_role=0, fail_batch=0, jitter=1, env=None):
    out = subprocess.check_output([exe, str(nbatch), str(nslot), str(inline), str(fail_role), str(fail_batch),
                                   str(jitter)], timeout=120, env=env)
    rv, drained, violation, slots_for = (int(x) for x in out.split())
    return rv, drained, violation, slots_for


@pytest.mark.parametrize("nslot", [2, 3, 4, 8, 18, 32])
@pytest.mark.parametrize("nbatch", [0, 1, 2, 5, 40])
def test_order_and_slot_reuse(exe, nbatch, nslot):
    rv, drained, violation, _ = run(exe, nbatch, nslot)
    assert (rv, drained, violation) == (0, nbatch, 0)


@pytest.mark.parametrize("nbatch", [0, 1, 7])
def test_inline_variant(exe, nbatch):
    rv, drained, violation, _ = run(exe, nbatch, 2, inline=1)
    assert (rv, drained, violation) == (0, nbatch, 0)


@pytest.mark.parametrize("role,code", [(1, -5), (2, -6), (3, -7), (4, -8)])
@pytest.mark.parametrize("at", [0, 3, 11])
def test_first_error_stops_everything(exe, role, code, at):
    for nslot in (2, 4, 9):
        rv, drained, violation, _ = run(exe, 12, nslot, fail_role=role, fail_batch=at)
        assert rv == code and violation == 0
        assert drained <= at   # nothing behind the failing batch was written


def test_slot_defaults(exe):
    env = {k: v for k, v in os.environ.items() if k != "GPUMT_SLOTS"}
    assert run(exe, 0, 1, env=env)[3] == 4            # one device: 4 slots
    assert run(exe, 0, 8, env=env)[3] == 18           # 8 devices: 2 per device + 2
    assert run(exe, 0, 16, env=env)[3] == 32          # capped at MT_NSLOT
    assert run(exe, 0, 1, env=dict(env, GPUMT_SLOTS="32"))[3] == 12   # 12 kernel streams / marks per device
    assert run(exe, 0, 1, env=dict(env, GPUMT_SLOTS="1"))[3] == 2
