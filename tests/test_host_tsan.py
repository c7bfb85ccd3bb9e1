"""The host engines of lz4-mt and of the 16-byte-header codecs (mt16_engine.inc, as snappy-mt) and the batch
pipeline (mt_pipe.c) under
ThreadSanitizer: reader, caller and writer threads over the slots, the counters and the callbacks, with a
plain-C stand-in for the device (tests/host/tsan_harness.c; the fiber emulator cannot run under TSan).
The reference has no race detection; its worker threads share the same kind of state
behind two mutexes."""
import os
import subprocess

import pytest

import helpers as H

HOST = os.path.join(H.ROOT, "zstdmt_amd", "csrc", "host")
EXE = os.path.join(H.ROOT, "tests", "host", "tsan_harness")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["gcc", "-O1", "-g", "-pthread", "-fsanitize=thread", "-I" + os.path.join(H.ROOT, "include"),
                           "-I" + HOST, os.path.join(H.ROOT, "tests", "host", "tsan_harness.c"),
                           os.path.join(HOST, "snappymt_engine.c"), os.path.join(HOST, "lz4mt_engine.c"),
                           os.path.join(HOST, "mt_pipe.c"), os.path.join(H.ROOT, "oracle", "snappy_oracle.c"),
                           os.path.join(H.ROOT, "oracle", "lz4_oracle.c"), "-o", EXE])
    return EXE


@pytest.mark.parametrize("nbytes,chunk,threads,slots", [(1 << 20, 4096, 4, 4), (1 << 20, 4096, 1, 4), (3000000, 65536, 3, 2),
                                                        (0, 4096, 2, 4), (5000000, 4096, 8, 12), (70000, 4096, 2, 32)])
@pytest.mark.parametrize("codec", ["snappy", "lz4"])
def test_engine_threads_are_race_free(exe, nbytes, chunk, threads, slots, codec):
    env = dict(os.environ, GPUMT_BATCH_KB="64", TSAN_OPTIONS="halt_on_error=0")
    p = subprocess.run([exe, str(nbytes), str(chunk), str(threads), str(slots), codec], capture_output=True, text=True,
                       timeout=300, env=env)
    if "FATAL: ThreadSanitizer" in p.stderr:
        pytest.skip("ThreadSanitizer cannot start in this environment: " + p.stderr.strip().splitlines()[0])
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[:3000]
    rv_c, rv_d, frames, ok = (int(x) for x in p.stdout.split())
    assert (rv_c, rv_d, ok) == (0, 0, 1) and frames == max(1, -(-nbytes // chunk))
