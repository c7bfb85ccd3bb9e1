"""The command line front ends on the CPU: programs/zmt_cli.c linked against the emulated host library
(tests/emu/bin, `make -C tests/emu cli`) -- option letters, personalities by argv[0], file handling, -B
report lines, -l list mode and exit codes of the reference CLI (programs/main.c), on files small
enough for the emulator.  The same checks run at full size against the device in tests/test_gpu_cli.py."""
import json
import os
import re
import subprocess
import zlib

import pytest

import helpers as H
from golden import cases

EMU_DIR = os.path.join(H.ROOT, "tests", "emu")
BIN = os.path.join(EMU_DIR, "bin")
LZ4, ZSTD, BROTLI, SNAPPY = (os.path.join(BIN, n) for n in ("lz4-mt", "zstd-mt", "brotli-mt", "snappy-mt"))
ENV = dict(os.environ, GPUMT_BATCH_KB="256")


@pytest.fixture(scope="module", autouse=True)
def built():
    H.locked_make(EMU_DIR, "cli", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def run(args, data=None, check=True):
    p = subprocess.run(args, input=data, capture_output=True, timeout=300, env=ENV)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


def test_lz4_files_in_place_and_report_lines(tmp_path):
    data = cases.text(90000, 7) + cases.rnd(5000, 1)
    f = tmp_path / "blob"
    f.write_bytes(data)
    p = run([LZ4, "-1", "-T4", "-B", str(f)])
    out = (tmp_path / "blob.lz4").read_bytes()
    assert not f.exists()                                   # replaced, like gzip
    assert out == H.oracle_compress(data, 4 << 20)          # default chunk 4 MiB: one record
    lines = p.stderr.decode().splitlines()
    assert lines[0] == "Level;Threads;InSize;OutSize;Frames"
    assert lines[1] == f"1;4;{len(data)};{len(out)};1"
    assert lines[2] == "Real;User;Sys;MaxMem" and re.fullmatch(r"\d+\.\d+;\d+\.\d+;\d+\.\d+;\d+", lines[3])
    run([LZ4, "-d", "-T4", str(tmp_path / "blob.lz4")])
    assert f.read_bytes() == data and not (tmp_path / "blob.lz4").exists()


def test_lz4_pipes_test_mode_and_levels():
    data = cases.text(60000, 2)
    p = run([LZ4, "-1", "-T2", "-b", "1", "-c"], data)
    assert p.stdout == H.oracle_compress(data, 1 << 20)
    assert run([LZ4, "-d", "-c"], p.stdout).stdout == data
    assert run([os.path.join(BIN, "lz4cat-mt")], p.stdout).stdout == data
    assert run([LZ4, "-t"], p.stdout).returncode == 0
    bad = bytearray(p.stdout)
    bad[len(bad) // 2] ^= 0xFF
    r = run([LZ4, "-t"], bytes(bad), check=False)
    assert r.returncode == 1 and b"lz4-mt" in r.stderr
    # no level option: the reference's default, level 3 = LZ4HC (programs/lz4-mt.c:19)
    assert run([LZ4, "-c"], data).stdout == H.oracle_compress_level(data, 4 << 20, 3)
    assert run([LZ4, "-9", "-c"], data[:20000]).stdout == H.oracle_compress_level(data[:20000], 4 << 20, 9)
    assert run([LZ4, "-12", "-c"], data[:6000]).stdout == H.oracle_compress_level(data[:6000], 4 << 20, 12)
    assert run([LZ4, "-13", "-c"], b"abc" * 100, check=False).returncode != 0


def test_zstd_keep_force_output_suffix(tmp_path):
    data = cases.text(80000, 12)
    f = tmp_path / "book.txt"
    f.write_bytes(data)
    p = run([ZSTD, "-1", "-T8", "-k", "-B", str(f)])
    z = tmp_path / "book.txt.zst"
    assert f.exists() and z.exists()
    st = z.read_bytes()
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data
    assert p.stderr.decode().splitlines()[1] == f"1;8;{len(data)};{len(st)};1"
    r = run([ZSTD, "-1", "-k", str(f)], check=False)       # refuses to overwrite without -f
    assert r.returncode == 1 and b"already exists" in r.stderr
    run([ZSTD, "-1", "-k", "-f", str(f)])
    f.unlink()
    run([ZSTD, "-d", str(z)])
    assert f.read_bytes() == data and not z.exists()
    run([ZSTD, "-1", "-o", str(tmp_path / "x.bin"), str(f)])
    assert f.exists()                                       # -o keeps the input
    assert run([os.path.join(BIN, "zstdcat-mt"), str(tmp_path / "x.bin")]).stdout == data
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), (tmp_path / "x.bin").read_bytes(), threads=2)
        assert rv == 0 and back == data


def test_bad_arguments_and_usage():
    assert run([LZ4, "-T", "500", "-c"], b"x", check=False).returncode == 1
    assert run([ZSTD, "-23", "-c"], b"x", check=False).returncode == 1
    assert run([SNAPPY, "-2", "-c"], b"x", check=False).returncode == 1
    assert run([BROTLI, "-12", "-c"], b"x", check=False).returncode == 1
    assert b"Usage" in run([ZSTD, "-h"]).stdout
    assert run([LZ4, "-d", "-c"], b"not a stream at all", check=False).returncode == 1


def test_brotli_personalities_on_a_reference_stream(tmp_path):
    bdir = os.path.join(H.GOLDEN_DIR, "brotli")
    with open(os.path.join(bdir, "manifest.json")) as f:
        ent = json.load(f)["cases"]["b_hello"]
    st = open(os.path.join(bdir, ent["out_file"]), "rb").read()
    f = tmp_path / "page.html.brot"
    f.write_bytes(st)
    run([BROTLI, "-d", "-k", "-T4", str(f)])
    out = (tmp_path / "page.html").read_bytes()
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]
    assert run([os.path.join(BIN, "brotlicat-mt")], st).stdout == out
    assert run([BROTLI, "-t"], st).returncode == 0
    data = cases.english(30000, 3)
    z = run([BROTLI, "-1", "-c"], data).stdout
    assert len(z) < len(data) and H.oracle_brotlimt_decompress(z, len(data) + 65536) == data
    assert run([os.path.join(BIN, "unbrotli-mt"), "-c"], z).stdout == data
    bad = bytearray(z)
    bad[40] ^= 0xFF
    bad[41] ^= 0xFF
    r = run([BROTLI, "-t"], bytes(bad), check=False)
    assert r.returncode == 1


def test_snappy_tool(tmp_path):
    data = cases.text(150000, 64) + bytes(20000)
    f = tmp_path / "blob"
    f.write_bytes(data)
    p = run([SNAPPY, "-T4", "-B", str(f)])
    out = (tmp_path / "blob.snp").read_bytes()
    assert not f.exists() and H.oracle_snappymt_decompress(out, len(data) + 64) == data
    frames = -(-len(data) // 65536)
    assert p.stderr.decode().splitlines()[1] == f"0;4;{len(data)};{len(out)};{frames}"
    assert run([os.path.join(BIN, "snappycat-mt"), str(tmp_path / "blob.snp")]).stdout == data
    run([os.path.join(BIN, "unsnappy-mt"), str(tmp_path / "blob.snp")])
    assert f.read_bytes() == data
    if H.have_libsnappy():
        assert run([SNAPPY, "-d", "-c"], H.snappymt_stream(data, 65536)).stdout == data


def test_list_mode(tmp_path):
    """-l: sizes and ratio in the reference's layout (programs/main.c:383-418); -lv adds method, crc32 of the
    content and the file date, -C switches the crc off; the file stays."""
    data = cases.text(70000, 5)
    f = tmp_path / "t.txt"
    f.write_bytes(data)
    run([LZ4, "-1", "-k", str(f)])
    z = tmp_path / "t.txt.lz4"
    csz = z.stat().st_size
    lines = run([LZ4, "-l", str(z)]).stdout.decode().splitlines()
    assert lines[0].split() == ["compressed", "uncompressed", "ratio", "uncompressed_name"]
    cols = lines[1].split()
    assert int(cols[0]) == csz and int(cols[1]) == len(data) and cols[3] == str(z)
    assert abs(float(cols[2].rstrip("%")) - (100 - csz * 100 / len(data))) < 0.01
    assert z.exists()
    lines = run([LZ4, "-l", "-v", str(z)]).stdout.decode().splitlines()
    assert lines[0].split()[:2] == ["method", "crc32"]
    cols = lines[1].split()
    assert cols[0] == "lz4" and int(cols[1], 16) == (zlib.crc32(data) & 0xFFFFFFFF)
    assert run([LZ4, "-l", "-v", "-C", str(z)]).stdout.decode().splitlines()[1].split()[1] == "00000000"
    bad = bytearray(z.read_bytes())
    bad[40] ^= 0xFF
    zb = tmp_path / "bad.lz4"
    zb.write_bytes(bytes(bad))
    p = run([LZ4, "-l", str(zb)], check=False)
    assert p.returncode != 0 and p.stdout.decode().splitlines()[1].split()[:3] == ["-", "-", "-"]


@pytest.mark.skipif(H.liblz4_frame(b"x") is None, reason="liblz4 not on this box")
def test_plain_streams_through_the_cat_tools():
    """Plain .lz4 / .zst streams (no skippable records) in several rounds of the incremental reader
    (GPUMT_BATCH_KB=256: each round holds about one frame)."""
    parts = [cases.text(60000 + 7000 * i, seed=70 + i) for i in range(5)] + [cases.rnd(30000, 5), b"", cases.text(1234, seed=3)]
    data = b"".join(parts)
    lz = b"".join(H.liblz4_frame(p, content_size=i & 1, block_id=4) for i, p in enumerate(parts))
    assert run([os.path.join(BIN, "lz4cat-mt")], lz).stdout == data
    r = run([os.path.join(BIN, "lz4cat-mt")], lz[:-3], check=False)
    assert r.returncode != 0 and data.startswith(r.stdout)
    if H.libzstd_frame(b"x") is not None:
        zs = b"".join(H.libzstd_frame(p, content_size=i & 1, checksum=(i >> 1) & 1) for i, p in enumerate(parts))
        assert run([os.path.join(BIN, "zstdcat-mt")], zs).stdout == data
