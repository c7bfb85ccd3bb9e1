"""The command line front ends (programs/zmt_cli.c -> zstdmt_amd/bin/{lz4-mt,zstd-mt}): option
letters, file handling and -B report lines of the reference CLI (programs/main.c), typed the way
BASELINE.json writes its configurations."""
import json
import os
import re
import subprocess

import pytest

import helpers as H
from golden import cases

pytestmark = pytest.mark.gpu

BIN = os.path.join(H.ROOT, "zstdmt_amd", "bin")
LZ4 = os.path.join(BIN, "lz4-mt")
ZSTD = os.path.join(BIN, "zstd-mt")
with open(os.path.join(H.GOLDEN_DIR, "manifest.json")) as _f:
    MAN = json.load(_f)["cases"]


def run(args, data=None, check=True):
    p = subprocess.run(args, input=data, capture_output=True, timeout=300)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


def test_lz4_config1_style_invocation(tmp_path):
    """BASELINE configs[0] as typed (`lz4-mt -1 -T4` on PRNG bytes, default 4 MiB chunks; 16 MiB here
    -> 4 frames): byte-identical stream, -B statistics lines, in-place file handling."""
    data = cases.rnd(16 << 20, 7)
    f = tmp_path / "blob"
    f.write_bytes(data)
    p = run([LZ4, "-1", "-T4", "-B", str(f)])
    out = (tmp_path / "blob.lz4").read_bytes()
    assert not f.exists()                                   # replaced, like gzip
    assert out == H.oracle_compress(data, 4 << 20)          # bit-exact with the reference path
    lines = p.stderr.decode().splitlines()
    assert lines[0] == "Level;Threads;InSize;OutSize;Frames"
    assert lines[1] == f"1;4;{len(data)};{len(out)};4"
    assert lines[2] == "Real;User;Sys;MaxMem" and re.fullmatch(r"\d+\.\d+;\d+\.\d+;\d+\.\d+;\d+", lines[3])
    run([LZ4, "-d", "-T4", str(tmp_path / "blob.lz4")])
    assert f.read_bytes() == data and not (tmp_path / "blob.lz4").exists()


def test_lz4_pipes_chunk_option_and_golden():
    name = "text_3x128k_p100"
    chunk, thunk = cases.CASES[name]
    assert chunk == 131072
    data = thunk()
    # -b takes MiB; 128 KiB chunks need the library default -> use a 1 MiB case for -b, stdin/stdout here
    p = run([LZ4, "-1", "-T2", "-b", "1", "-c"], data)
    assert p.stdout == H.oracle_compress(data, 1 << 20)
    back = run([LZ4, "-d", "-c"], p.stdout)
    assert back.stdout == data
    assert run([LZ4, "-t"], p.stdout).returncode == 0
    bad = bytearray(p.stdout)
    bad[len(bad) // 2] ^= 0xFF
    r = run([LZ4, "-t"], bytes(bad), check=False)
    assert r.returncode == 1 and b"lz4-mt" in r.stderr


def test_lz4_default_level_is_3_and_every_level_is_served():
    data = cases.text(300000)
    # no level option: the reference's default, level 3 = LZ4HC (programs/lz4-mt.c:19), 4 MiB chunks
    p = run([LZ4, "-c"], data)
    assert p.stdout == H.oracle_compress_level(data, 4 << 20, 3)
    assert run([LZ4, "-d", "-c"], p.stdout).stdout == data
    p9 = run([LZ4, "-9", "-c"], data)
    assert p9.stdout == H.oracle_compress_level(data, 4 << 20, 9)
    p12 = run([LZ4, "-12", "-c"], data)
    assert p12.stdout == H.oracle_compress_level(data, 4 << 20, 12)
    r = run([LZ4, "-13", "-c"], b"abc" * 1000, check=False)
    assert r.returncode != 0


def test_zstd_round_trip_keep_force_suffix(tmp_path):
    data = cases.text(3 * (1 << 20) + 4321, 12)
    f = tmp_path / "book.txt"
    f.write_bytes(data)
    p = run([ZSTD, "-1", "-T8", "-k", "-B", str(f)])
    z = tmp_path / "book.txt.zst"
    assert f.exists() and z.exists()
    st = z.read_bytes()
    assert H.oracle_zstdmt_decompress(st, len(data) + 64) == data     # decompress-identical
    assert p.stderr.decode().splitlines()[1] == f"1;8;{len(data)};{len(st)};4"
    # refuses to overwrite without -f
    r = run([ZSTD, "-1", "-k", str(f)], check=False)
    assert r.returncode == 1 and b"already exists" in r.stderr
    run([ZSTD, "-1", "-k", "-f", str(f)])
    f.unlink()
    run([ZSTD, "-d", str(z)])
    assert f.read_bytes() == data and not z.exists()
    # -o and -S
    run([ZSTD, "-1", "-o", str(tmp_path / "x.bin"), str(f)])
    assert f.exists()                                       # -o keeps the input
    assert run([ZSTD, "-d", "-c", str(tmp_path / "x.bin")]).stdout == data
    if H.have_zref():
        rv, back, _, _ = H.zstdmt_decompress_via(H.zref(), (tmp_path / "x.bin").read_bytes(), threads=2)
        assert rv == 0 and back == data


def test_bad_arguments():
    assert run([LZ4, "-T", "500", "-c"], b"x", check=False).returncode == 1
    assert run([ZSTD, "-23", "-c"], b"x", check=False).returncode == 1
    assert b"Usage" in run([ZSTD, "-h"]).stdout


def test_brotli_decompress_personalities(tmp_path):
    """brotli-mt: decompression of a stream the reference wrote (committed fixture), through -d, the
    un*/cat personalities and -t; compression through -1 -c (decompress-identical)."""
    bdir = os.path.join(H.GOLDEN_DIR, "brotli")
    with open(os.path.join(bdir, "manifest.json")) as f:
        ent = json.load(f)["cases"]["b_english_chunks"]
    st = open(os.path.join(bdir, ent["out_file"]), "rb").read()
    brotli = os.path.join(BIN, "brotli-mt")
    f = tmp_path / "page.html.brot"
    f.write_bytes(st)
    run([brotli, "-d", "-k", "-T4", str(f)])
    out = (tmp_path / "page.html").read_bytes()
    assert len(out) == ent["in_len"] and H.sha256(out) == ent["in_sha256"]
    assert run([os.path.join(BIN, "brotlicat-mt")], st).stdout == out
    assert run([brotli, "-t"], st).returncode == 0
    bad = bytearray(st)
    bad[100] ^= 0xFF
    bad[101] ^= 0xFF
    r = run([brotli, "-t"], bytes(bad), check=False)
    assert r.returncode == 1 and b"Could not decompress frame at once" in r.stderr
    # compression round trip through the command line
    data = cases.english(200000, 3)
    z = run([brotli, "-1", "-c"], data).stdout
    assert len(z) < len(data) and H.oracle_brotlimt_decompress(z, len(data) + 65536) == data
    assert run([brotli, "-d", "-c"], z).stdout == data


def test_list_mode(tmp_path):
    """-l: sizes and ratio in the reference's layout (programs/main.c:383-418); -lv adds method, crc32
    of the content and the file date, -C switches the crc off; the file stays."""
    import zlib
    data = cases.text(300000, 5)
    f = tmp_path / "t.txt"
    f.write_bytes(data)
    run([LZ4, "-1", "-k", str(f)])
    z = tmp_path / "t.txt.lz4"
    csz = z.stat().st_size
    p = run([LZ4, "-l", str(z)])
    lines = p.stdout.decode().splitlines()
    assert lines[0].split() == ["compressed", "uncompressed", "ratio", "uncompressed_name"]
    cols = lines[1].split()
    assert int(cols[0]) == csz and int(cols[1]) == len(data) and cols[3] == str(z)
    assert abs(float(cols[2].rstrip("%")) - (100 - csz * 100 / len(data))) < 0.01
    assert z.exists()
    p = run([LZ4, "-l", "-v", str(z)])
    lines = p.stdout.decode().splitlines()
    assert lines[0].split()[:2] == ["method", "crc32"]
    cols = lines[1].split()
    assert cols[0] == "lz4" and int(cols[1], 16) == (zlib.crc32(data) & 0xFFFFFFFF)
    p = run([LZ4, "-l", "-v", "-C", str(z)])
    assert p.stdout.decode().splitlines()[1].split()[1] == "00000000"
    # a damaged file lists as dashes and fails
    bad = bytearray(z.read_bytes())
    bad[40] ^= 0xFF
    zb = tmp_path / "bad.lz4"
    zb.write_bytes(bytes(bad))
    p = run([LZ4, "-l", str(zb)], check=False)
    assert p.returncode != 0 and p.stdout.decode().splitlines()[1].split()[:3] == ["-", "-", "-"]


@pytest.mark.skipif(H.liblz4_frame(b"x") is None, reason="liblz4 not on this box")
def test_plain_streams_are_decoded_incrementally():
    """A plain .lz4 / .zst stream (frames of the lz4 / zstd tools, no skippable records) larger than
    several device batches: the engines read ahead about one batch, decode the complete frames and keep
    the tail (GPUMT_BATCH_MB=16 makes a 90 MiB stream many rounds; one frame is larger than a batch)."""
    import subprocess
    parts = [cases.text(5_000_000 + 300_000 * i, seed=70 + i) for i in range(14)] + [cases.text(40 << 20, seed=99)]
    parts += [cases.rnd(3_000_000, 5), b"", cases.text(1_234_567, seed=3)]
    data = b"".join(parts)
    env = dict(os.environ, GPUMT_BATCH_MB="16")
    lz = b"".join(H.liblz4_frame(p, content_size=i & 1) for i, p in enumerate(parts))
    r = subprocess.run([os.path.join(BIN, "lz4cat-mt")], input=lz, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-300:]
    assert r.stdout == data
    zs = b"".join(H.libzstd_frame(p, content_size=i & 1, checksum=(i >> 1) & 1) for i, p in enumerate(parts))
    r = subprocess.run([os.path.join(BIN, "zstdcat-mt")], input=zs, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-300:]
    assert r.stdout == data
    # a stream cut inside its last frame is an error, after everything before it was written
    r = subprocess.run([os.path.join(BIN, "lz4cat-mt")], input=lz[:-3], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=600)
    assert r.returncode != 0 and data.startswith(r.stdout)


@pytest.mark.parametrize("env", [
    {"GPUMT_SLOTS": "2", "GPUMT_BATCH_MB": "16"},
    {"GPUMT_SLOTS": "8", "GPUMT_BATCH_MB": "32", "GPU_MAX_HW_QUEUES": "4"},
    {"GPUMT_PINNED_CACHE_MB": "0", "GPUMT_DEVICE_CACHE_MB": "0", "GPUMT_TRACE": "1"},
    # slots dealt out to several device contexts (here: the same GPU opened twice / three times, and "all")
    {"GPUMT_DEVICES": "0,0", "GPUMT_BATCH_MB": "16"},
    {"GPUMT_DEVICES": "0,0,0", "GPUMT_SLOTS": "5", "GPUMT_BATCH_MB": "16"},
    {"GPUMT_DEVICES": "all"},
])
def test_pipeline_knobs_round_trip(env):
    """the pipeline's environment knobs (slots, batch size, caches off, trace, several devices) change
    timing, never bytes"""
    import subprocess
    data = cases.text(70 << 20, seed=17) + cases.rnd(2 << 20, 4)
    e = dict(os.environ, **env)
    want = H.oracle_compress(data, 1 << 20)
    for tool, cat in ((LZ4, "lz4cat-mt"), (ZSTD, "zstdcat-mt"), (os.path.join(BIN, "brotli-mt"), "brotlicat-mt")):
        c = subprocess.run([tool, "-1", "-b", "1", "-c"], input=data, capture_output=True, env=e, timeout=300)
        assert c.returncode == 0, c.stderr[-300:]
        if tool == LZ4:
            assert c.stdout == want
        d = subprocess.run([os.path.join(BIN, cat)], input=c.stdout, capture_output=True, env=e, timeout=300)
        assert d.returncode == 0 and d.stdout == data, (tool, d.stderr[-300:])


def test_batches_are_dealt_over_device_contexts_in_order():
    """Multi-device readiness (the in-order multi-worker writer of lib/lz4-mt_compress.c:178-205): with
    GPUMT_DEVICES=0,0 one LZ4MT_compressCCtx / decompressDCtx drives TWO device contexts.  Batch b goes to slot
    b % nslot, slot s to device context s % 2 -- asserted from the engine's own launch trace -- and the frames leave
    in input order: the stream is the reference's, byte for byte.  (Same GPU opened twice: the box has one.)"""
    import re
    import subprocess
    data = cases.text(160 << 20, seed=23)
    e = dict(os.environ, GPUMT_DEVICES="0,0", GPUMT_BATCH_MB="16", GPUMT_SLOTS="4", GPUMT_TRACE="2")
    c = subprocess.run([LZ4, "-1", "-b", "1", "-c"], input=data, capture_output=True, env=e, timeout=300)
    assert c.returncode == 0, c.stderr[-300:]
    assert c.stdout == H.oracle_compress(data, 1 << 20)            # order preserved across the two contexts
    tr = re.findall(r"\[lz4mt compress\] launch slot (\d+) -> device context (\d+) of (\d+), stream (\d+), (\d+) records",
                    c.stderr.decode())
    assert len(tr) >= 8, c.stderr[-500:]                            # 16 MiB batches: ten of them
    for b, (slot, dev, ndev, stream, nrec) in enumerate(tr):
        assert int(ndev) == 2 and int(slot) == b % 4 and int(dev) == int(slot) % 2, (b, slot, dev)
    assert {int(t[1]) for t in tr} == {0, 1}                        # both contexts worked
    assert sum(int(t[4]) for t in tr) == 160                        # every record exactly once
    d = subprocess.run([os.path.join(BIN, "lz4cat-mt")], input=c.stdout, capture_output=True, env=e, timeout=300)
    assert d.returncode == 0 and d.stdout == data
    td = re.findall(r"\[lz4mt decompress\] launch slot (\d+) -> device context (\d+) of 2", d.stderr.decode())
    assert len(td) >= 2 and all(int(s_) % 2 == int(dv) for s_, dv in td) and {int(t[1]) for t in td} == {0, 1}
