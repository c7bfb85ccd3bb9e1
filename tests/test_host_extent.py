"""Host-side frame splitting of the plain .lz4 / .zst paths (lz4_frame_extent, zstd_frame_extent in the
engines) on the CPU: frame lengths equal what liblz4 / libzstd wrote, the capacity bound covers the
content, and no proper prefix of a frame is mistaken for a complete one (the incremental reader relies
on that to wait for more input)."""
import os
import subprocess

import pytest

import helpers as H
from cases import rnd, text

HOST = os.path.join(H.ROOT, "tests", "host")


def build(kind):
    exe = os.path.join(HOST, "extent_harness_" + kind)
    subprocess.check_call(["gcc", "-O1", "-g", "-pthread", "-w", "-I" + os.path.join(H.ROOT, "include"),
                           "-I" + os.path.join(H.ROOT, "zstdmt_amd", "csrc", "host")]
                          + (["-DHARNESS_ZSTD"] if kind == "zstd" else [])
                          + [os.path.join(HOST, "extent_harness.c"),
                             os.path.join(H.ROOT, "zstdmt_amd", "csrc", "host", "mt_pipe.c"),
                             "-Wl,--unresolved-symbols=ignore-all", "-o", exe])
    return exe


def frames(kind):
    parts = [text(300_000, 5), b"", rnd(70_000, 2), text(17, 3), bytes(500_000), text(1_100_000, 9)]
    out = []
    for i, p in enumerate(parts):
        if kind == "lz4":
            out.append(H.liblz4_frame(p, content_size=i & 1, checksum=(i >> 1) & 1, block_checksum=(i >> 2) & 1,
                                      block_id=4 + i % 4))
        else:
            out.append(H.libzstd_frame(p, content_size=i & 1, checksum=(i >> 1) & 1, level=1 + 3 * (i % 3)))
    return parts, out


@pytest.mark.parametrize("kind", ["lz4", "zstd"])
def test_frame_extents(kind, tmp_path):
    if (kind == "lz4" and H.liblz4_frame(b"x") is None) or (kind == "zstd" and H.libzstd_frame(b"x") is None):
        pytest.skip("codec library not on this box")
    exe = build(kind)
    parts, fr = frames(kind)
    f = tmp_path / "stream.bin"
    f.write_bytes(b"".join(fr))
    rows = [tuple(int(x) for x in ln.split()) for ln in subprocess.check_output([exe, str(f)]).decode().splitlines()]
    assert len(rows) == len(fr)
    off = 0
    for (o, ln, bound, flag), frame, part in zip(rows, fr, parts):
        assert (o, ln) == (off, len(frame))
        assert bound >= len(part)
        off += len(frame)
    # no proper prefix of a frame passes for a whole one
    for frame in (fr[0], fr[2], fr[3]):
        f.write_bytes(frame)
        full, wrong = (int(x) for x in subprocess.check_output([exe, str(f), "cut"]).split())
        assert (full, wrong) == (len(frame), 0)


@pytest.mark.parametrize("kind", ["lz4", "zstd"])
def test_damaged_frame_starts_are_rejected_at_once(kind, tmp_path):
    """a frame start that can never become a frame is `invalid`, not `wait`: the incremental reader of the plain
    .lz4 / .zst paths then reports the error instead of buffering the rest of the input"""
    if (kind == "lz4" and H.liblz4_frame(b"x") is None) or (kind == "zstd" and H.libzstd_frame(b"x") is None):
        pytest.skip("codec library not on this box")
    exe = build(kind)
    f = tmp_path / "f.bin"

    def verdict(b):
        f.write_bytes(b)
        return subprocess.check_output([exe, str(f), "verdict"]).decode().strip()

    if kind == "lz4":
        good = H.liblz4_frame(text(200_000, 4), content_size=1, checksum=1, block_checksum=0, block_id=4)
        assert verdict(good) == "frame" and verdict(good[:1000]) == "wait"
        bad = bytearray(good[:1000]); bad[4] ^= 0x80                 # version bits
        assert verdict(bytes(bad)) == "invalid"
        bad = bytearray(good[:1000]); bad[5] = 0x30                  # block size id 3
        assert verdict(bytes(bad)) == "invalid"
        bad = bytearray(good[:1000]); bad[15 + 2] = 0x7F             # first block header: 8 MiB in a 64 KiB frame
        assert verdict(bytes(bad)) == "invalid"
    else:
        good = H.libzstd_frame(text(400_000, 4), content_size=1, checksum=1, level=1)
        assert verdict(good) == "frame" and verdict(good[:1000]) == "wait"
        bad = bytearray(good[:1000]); bad[4] |= 8                    # reserved bit of the frame header descriptor
        assert verdict(bytes(bad)) == "invalid"
        hdr = 6 + (1 << (good[4] >> 6)) - 1 if (good[4] >> 6) else 6
        bad = bytearray(good[:1000])
        # first block header -> reserved block type 3
        i = 5 + (0 if (good[4] >> 5) & 1 else 1) + [0, 1, 2, 4][good[4] & 3] + ([1, 2, 4, 8][good[4] >> 6] if (good[4] >> 6) or ((good[4] >> 5) & 1) else 0)
        bad[i] |= 6
        assert verdict(bytes(bad)) == "invalid"
