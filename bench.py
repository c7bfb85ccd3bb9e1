#!/usr/bin/env python3
"""bench.py -- lz4-mt hot path on MI355X: compress + decompress of the synthetic enwik-style buffer.

Workload (BASELINE.json configs[1]): lz4-mt level 1, 8 GiB enwik-style synthetic text per GPU,
128 KiB chunks (65 536 chunks -> 65 536 records).  One "step" = one full pass of the hot path over
that buffer, inputs already resident in HBM:

    compress_batch (XXH32 of every chunk + bit-exact LZ4 frame encode into per-chunk slots)
    -> compact (scan of record sizes + ordered pack into the MT stream)
    -> probe_sizes (content-size fields + scan)
    -> decompress_batch (frame decode + XXH32 content-checksum verification)

value = uncompressed MB (1e6 B) round-tripped per second, whole job (all ranks).  Per-direction
rates, per-kernel HIP-event times and the HBM roofline of the kernels are carried alongside.

Multi-GPU: chunks are independent, so each rank takes its own 8 GiB shard (weak scaling); the only
exchange is the all-gather of per-rank stream sizes that gives every rank its offset in the final
stream (frame reassembly); `--gather` additionally times the RCCL gather of the compressed
segments to rank 0 (reported separately).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12        # B/s, MI355X spec (MI355X_MICROARCH.md)
HBM_COPY = 6.29e12       # measured float4 copy ceiling, same guide
SEED = 20260926


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=8.0, help="uncompressed GiB per GPU")
    ap.add_argument("--codec", choices=("lz4", "zstd", "brotli"), default="lz4",
                    help="lz4 = BASELINE configs[1] (the metric's config); zstd = configs[3], zstd-mt level 1; "
                         "brotli = configs[4], brotli-mt decompress of level-1 streams at 1 MiB chunks")
    ap.add_argument("--chunk", type=int, default=0, help="0 = 128 KiB for lz4 (configs[1]), 1 MiB for zstd (level-1 default)")
    ap.add_argument("--dec-variant", type=int, default=0)
    ap.add_argument("--enc-variant", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-mib", type=int, default=2048, help="cpu_baseline sample size")
    ap.add_argument("--gather", action="store_true", help="also time the RCCL gather of segments")
    ap.add_argument("--verify", action="store_true", help="download and compare the round trip")
    ap.add_argument("--no-encoder", action="store_true",
                    help="--codec brotli: skip the device-encoder leg (profiling runs of the decoder alone)")
    return ap.parse_args()


def tools():
    t = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
    t.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
    return t


def cpu_baseline(args):
    """oracle/_ref (reference sources + liblz4) if present, else the oracle port; bounded sample."""
    exe = os.path.join(ROOT, "oracle", "cpu_bench")
    zstd = args.codec == "zstd"
    brotli = args.codec == "brotli"
    ref = os.path.join(ROOT, "oracle", "_ref", "libbrotlimt_ref.so" if brotli else
                       "libzstdmt_ref.so" if zstd else "liblz4mt_ref.so")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_bench"],
                              stdout=subprocess.DEVNULL)
    cores = os.cpu_count() or 1
    threads = min(cores, 128)   # LZ4MT_THREAD_MAX, lib/lz4-mt.h:28
    kind = "reference" if os.path.exists(ref) else "port"
    if (zstd or brotli) and kind == "port":
        return {"value": None, "unit": "MB/s", "cores": threads, "kind": "reference",
                "error": f"{os.path.basename(ref)} not present (the {args.codec} oracle has no compressor)"}
    n = args.cpu_mib << 20
    try:
        out = subprocess.check_output([exe, "reference-brotli" if brotli else "reference-zstd" if zstd else kind,
                                       ref if kind == "reference" else "-", str(n),
                                       str(args.chunk), str(threads), str(SEED)], timeout=600)
        r = json.loads(out)
    except Exception as e:  # report, never hide
        return {"value": None, "unit": "MB/s", "cores": threads, "kind": kind, "error": repr(e)}
    if brotli:
        return {"value": r["decompress_MBps"], "unit": "MB/s", "cores": threads, "kind": kind,
                "compress_MBps": r["compress_MBps"], "decompress_MBps": r["decompress_MBps"], "host_cpus": cores,
                "sample": f"{args.cpu_mib} MiB of the same synthetic text, {args.chunk}-byte chunks, level 1: "
                          f"BROTLIMT_decompressDCtx of the stream BROTLIMT_compressCCtx wrote, memcpy "
                          f"callbacks, T={threads}"}
    return {"value": r["roundtrip_MBps"], "unit": "MB/s", "cores": threads, "kind": kind,
            "compress_MBps": r["compress_MBps"], "decompress_MBps": r["decompress_MBps"],
            "host_cpus": cores,
            "sample": f"{args.cpu_mib} MiB of the same synthetic text, {args.chunk}-byte chunks, "
                      f"{'ZSTDCB' if zstd else 'LZ4MT'}_compressCCtx+decompressDCtx (level 1) with memcpy "
                      f"callbacks, T={threads}"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "RANK" in os.environ:
        # launched by torch.distributed.run: one process per GPU, RCCL ("nccl") process group.
        # torch must be imported BEFORE the native library so both share one HIP runtime.
        import torch
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import zstdmt_amd as z
    eng = z.Engine(local)
    L, h = eng.L, eng.h
    eng.set_variant("lz4_dec", args.dec_variant)
    eng.set_variant("lz4_enc", args.enc_variant)
    eng.set_variant("profile", 1)

    zstd = args.codec == "zstd"
    if args.codec == "brotli":
        return bench_brotli(args, eng, rank, world, dist)
    if not args.chunk:
        args.chunk = (1 << 20) if zstd else 131072
    n = int(args.gib * (1 << 30)) // args.chunk * args.chunk
    chunk = args.chunk
    nrec = eng.record_count(n, chunk)
    stride = eng.zstd_slot_stride(chunk) if zstd else eng.slot_stride(chunk)

    # ---- synthetic input, generated on the host in 256 MiB pieces, uploaded once ----
    d_in = eng.alloc(n + 64)
    piece = 256 << 20
    hbuf = np.empty(min(piece, n), np.uint8)
    T = tools()
    gen_threads = max(1, (os.cpu_count() or 1) // max(1, world))
    t0 = time.time()
    for off in range(0, n, piece):
        m = min(piece, n - off)
        # each rank generates a different part of the (conceptually world*n byte) corpus
        T.zmt_gen_text(hbuf.ctypes.data, m, SEED, rank * n + off, gen_threads)
        eng._ck(L.gpumt_memcpy_h2d(h, d_in.ptr + off, hbuf.ctypes.data, m, 0), "h2d")
        eng.sync(0)
    gen_s = time.time() - t0

    d_slots = eng.alloc(nrec * stride)
    d_rl = eng.alloc(nrec * 4)
    d_ro = eng.alloc((nrec + 1) * 8)
    d_stream = eng.alloc(nrec * stride)       # worst case
    d_ol = eng.alloc(nrec * 4)
    d_oo = eng.alloc((nrec + 1) * 8)
    d_st = eng.alloc(nrec * 4)
    d_out = eng.alloc(n + 64)

    def step():
        eng.timer_start(1)
        if zstd:
            eng.zstd_compress(d_in, n, chunk, d_slots, stride, d_rl)
        else:
            eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl)
        eng.timer_stop(1)
        eng.timer_start(2)
        eng.lz4_compact(d_slots, stride, d_rl, nrec, d_stream, d_ro)
        eng.timer_stop(2)
        eng.timer_start(3)
        if zstd:
            eng.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
            eng.zstd_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
        else:
            eng.lz4_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo)
            eng.lz4_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
        eng.timer_stop(3)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()
        eng.sync()

    for _ in range(args.warmup):
        step()
    barrier()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        eng.sync(0)
        # slots 1-3: API-level legs; 8..13: individual kernels (profile mode)
        slots = (("compress", 1), ("compact", 2), ("decompress", 3), ("k_lz4_enc", 9),
                 ("k_scan_compact", 10), ("k_lz4_dec", 11))
        if not zstd:
            slots += (("k_xxh32_c", 8), ("k_xxh32_d", 12), ("k_dec_frames", 13), ("k_dec_parse", 14),
                      ("k_dec_copy", 15))
        for name, slot in slots:
            acc[name] = acc.get(name, 0.0) + eng.timer_ms(slot)
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        tw = torch.tensor([wall], device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    ms = {k: v / args.steps for k, v in acc.items()}
    total_c = int(eng.download(d_ro, 8, np.uint64, offset=nrec * 8)[0])
    status = eng.download(d_st, nrec * 4, np.uint32)
    bad = int((status != 0).sum())

    # ---- frame reassembly across ranks: sizes all-gather (-> offsets), optional bulk gather ----
    gather_ms = None
    seg_off = 0
    if dist is not None:
        from zstdmt_amd.shard import exchange_segment_sizes
        sizes, seg_off = exchange_segment_sizes(total_c, device="cuda")
        if args.gather:
            gather_ms = rccl_gather(eng, dist, d_stream, sizes, rank, world)

    ok = True
    if args.verify:
        a = eng.download(d_in, n)
        b = eng.download(d_out, n)
        ok = bool((a == b).all())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    U = float(n)
    Cb = float(total_c)
    alg = U + Cb                              # algorithmic bytes either direction
    t_c = (ms["compress"] + ms["compact"]) * 1e-3
    t_d = ms["decompress"] * 1e-3
    if zstd:
        return report_zstd(args, eng, world, wall, ms, U, Cb, nrec, chunk, bad, ok, gen_s, seg_off, gather_ms, dist)
    kern = {}
    split = args.dec_variant == 0
    ntok_bytes = 0.0   # token list written by the parse kernel and read by the copy kernel
    for k, byt in (("k_xxh32_c", U), ("k_lz4_enc", U + Cb), ("k_scan_compact", 2 * Cb),
                   ("k_lz4_dec", U + Cb), ("k_xxh32_d", U), ("k_dec_frames", 0.0),
                   ("k_dec_parse", Cb), ("k_dec_copy", U + Cb)):
        if k.startswith("k_dec_") and not split:
            continue
        t = ms[k] * 1e-3
        kern[k] = {"ms": round(ms[k], 4), "alg_bytes": byt,
                   "GBps": round(byt / t / 1e9, 2) if t > 0 else None}
    dom = max(("k_lz4_enc", "k_lz4_dec", "k_xxh32_c", "k_xxh32_d", "k_scan_compact"),
              key=lambda k: ms[k])
    traffic = {}
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        with open(tf) as f:
            traffic = json.load(f).get("per_launch_bytes_8gib", {})

    def roof(k):
        t = ms[k] * 1e-3
        a = kern[k]["alg_bytes"] / t / 1e9
        kname = {"k_lz4_enc": {0: "zmt_lz4_enc3_p17_kernel" if 65536 < chunk <= 131072 else
                                   ("zmt_lz4_enc3_u16_kernel" if chunk <= 65536 else "zmt_lz4_enc3_u32_kernel"),
                               1: "zmt_lz4_enc_v1_kernel", 2: "zmt_lz4_enc_kernel"}[args.enc_variant],
                 "k_lz4_dec": ("zmt_dec_frames+parse+copy_kernel" if split else "zmt_lz4_dec_*"),
                 "k_xxh32_c": "zmt_xxh32_kernel", "k_xxh32_d": "zmt_xxh32_kernel",
                 "k_scan_compact": "zmt_compact_kernel", "k_dec_copy": "zmt_dec_copy_kernel",
                 "k_dec_parse": "zmt_dec_parse_kernel"}[k]
        return {"kernel": kname,
                "bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(a * 1e9 / HBM_PEAK, 5), "frac_of_copy_ceiling": round(a * 1e9 / HBM_COPY, 5),
                "alg_bytes_per_launch": kern[k]["alg_bytes"], "avg_launch_ms": round(ms[k], 4),
                "traffic": (traffic.get(kname) if abs(args.gib - 8.0) < 1e-9 else None)}

    step_s = wall / args.steps
    res = {
        "metric": "MB/s compress+decompress, 8 GiB synthetic, lz4-mt; % HBM roofline",
        "value": round(world * U / 1e6 / step_s, 1),
        "unit": "MB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"lz4-mt -1, {args.gib:g} GiB enwik-style synthetic per GPU, "
                               f"{chunk // 1024} KiB chunks, device-resident compress+decompress",
                   "chunk": chunk, "records_per_gpu": nrec, "level": 1,
                   "ratio": round(U / Cb, 4), "dec_variant": args.dec_variant,
                   "parallelism": f"chunk-sharded x{world}"},
        "compress_MBps": round(world * U / 1e6 / t_c, 1),
        "decompress_MBps": round(world * U / 1e6 / t_d, 1),
        "roofline": roof(dom),
        "roofline_decompress": roof("k_lz4_dec"),
        "roofline_decompress_copy_kernel": roof("k_dec_copy") if split else None,
        "roofline_decompress_path": {
            "what": "decode + XXH32 verify kernels together", "achieved": round(alg / t_d / 1e9, 2),
            "unit": "GB/s", "frac": round(alg / t_d / HBM_PEAK, 5)},
        "kernels": kern,
        "decode_errors": bad, "roundtrip_verified": ok if args.verify else None,
        "gen_s": round(gen_s, 2), "device": eng.name,
        "segment_offset_rank0": seg_off, "gather_ms": gather_ms,
    }
    if not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def report_zstd(args, eng, world, wall, ms, U, Cb, nrec, chunk, bad, ok, gen_s, seg_off, gather_ms, dist):
    """BASELINE configs[3]: zstd-mt level 1.  Kernels: zmt_zstd_enc_kernel + zmt_zstd_assemble_kernel
    (timer slot 9), zmt_zstd_dec_kernel (slot 11); algorithmic bytes U + C either direction."""
    alg = U + Cb
    t_c = (ms["compress"] + ms["compact"]) * 1e-3
    t_d = ms["decompress"] * 1e-3
    step_s = wall / args.steps

    traffic = {}
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf) and abs(args.gib - 8.0) < 1e-9 and chunk == 1 << 20:
        with open(tf) as f:
            traffic = json.load(f).get("per_launch_bytes_8gib", {})

    def roof(name, t_ms, pmc):
        a = alg / (t_ms * 1e-3) / 1e9
        tr = [traffic.get(k) for k in pmc if traffic.get(k) is not None]
        return {"kernel": name, "bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": round(a * 1e9 / HBM_PEAK, 5),
                "frac_of_copy_ceiling": round(a * 1e9 / HBM_COPY, 5), "alg_bytes_per_launch": alg,
                "avg_launch_ms": round(t_ms, 4), "traffic": sum(tr) if tr else None}

    enc = roof("zmt_zstd_enc_kernel(+assemble)", ms["k_lz4_enc"], ("zmt_zstd_enc_kernel", "zmt_zstd_assemble_kernel"))
    dec = roof("zmt_zstd_dec_small_kernel(+zmt_zstd_dec_kernel for frames with full-size tables)", ms["k_lz4_dec"],
               ("zmt_zstd_dec_small_kernel", "zmt_zstd_dec_kernel"))
    res = {
        "metric": "MB/s compress+decompress, 8 GiB synthetic, zstd-mt level 1; % HBM roofline",
        "value": round(world * U / 1e6 / step_s, 1), "unit": "MB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"zstd-mt -1, {args.gib:g} GiB enwik-style synthetic per GPU, "
                               f"{chunk // 1024} KiB chunks, device-resident compress+decompress",
                   "chunk": chunk, "records_per_gpu": nrec, "level": 1, "ratio": round(U / Cb, 4),
                   "parity": "decompress-identical (SURVEY 8a C4)", "parallelism": f"chunk-sharded x{world}"},
        "compress_MBps": round(world * U / 1e6 / t_c, 1),
        "decompress_MBps": round(world * U / 1e6 / t_d, 1),
        "roofline": enc if ms["k_lz4_enc"] >= ms["k_lz4_dec"] else dec,
        "roofline_compress": enc, "roofline_decompress": dec,
        "kernels": {"k_zstd_enc": {"ms": round(ms["k_lz4_enc"], 4)}, "k_scan_compact": {"ms": round(ms["k_scan_compact"], 4)},
                    "k_zstd_dec": {"ms": round(ms["k_lz4_dec"], 4)}},
        "decode_errors": bad, "roundtrip_verified": ok if args.verify else None,
        "gen_s": round(gen_s, 2), "device": eng.name,
        "segment_offset_rank0": seg_off, "gather_ms": gather_ms,
    }
    if not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def reference_brotli_stream(data, chunk, level, threads):
    """The workload's input: `data` compressed by the REFERENCE's brotli-mt (oracle/_ref, SURVEY 8d
    "cfg5: the same text compressed by own/oracle brotli at 1 MiB chunks").  The device has no brotli
    encoder yet, so the compressed input can only come from the reference build; this is input
    preparation on the host, outside the timed region."""
    so = os.path.join(ROOT, "oracle", "_ref", "libbrotlimt_ref.so")
    lib = C.CDLL(so)

    class Buf(C.Structure):
        _fields_ = [("buf", C.c_void_p), ("size", C.c_size_t), ("allocated", C.c_size_t)]
    FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Buf))

    class RdWr(C.Structure):
        _fields_ = [("fn_read", FN), ("arg_read", C.c_void_p), ("fn_write", FN), ("arg_write", C.c_void_p)]
    lib.BROTLIMT_createCCtx.restype = C.c_void_p
    lib.BROTLIMT_createCCtx.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.BROTLIMT_compressCCtx.restype = C.c_size_t
    lib.BROTLIMT_compressCCtx.argtypes = [C.c_void_p, C.POINTER(RdWr)]
    lib.BROTLIMT_freeCCtx.argtypes = [C.c_void_p]
    lib.BROTLIMT_isError.argtypes = [C.c_size_t]
    src = data.ctypes.data
    n = data.nbytes
    state = {"pos": 0}
    out = []

    def rd(_a, bp):
        b = bp.contents
        m = min(b.size, n - state["pos"])
        if m:
            C.memmove(b.buf, src + state["pos"], m)
        state["pos"] += m
        b.size = m
        return 0

    def wr(_a, bp):
        b = bp.contents
        out.append(C.string_at(b.buf, b.size))
        return 0
    frd, fwr = FN(rd), FN(wr)
    io = RdWr(frd, None, fwr, None)
    ctx = lib.BROTLIMT_createCCtx(threads, level, chunk)
    rv = lib.BROTLIMT_compressCCtx(ctx, C.byref(io))
    lib.BROTLIMT_freeCCtx(ctx)
    if lib.BROTLIMT_isError(rv):
        raise RuntimeError("reference brotli-mt compress failed")
    return b"".join(out)


def bench_brotli(args, eng, rank, world, dist):
    """BASELINE configs[4]: brotli-mt decompress.  Input: level-1 streams of the synthetic text at
    1 MiB chunks (the reference's default chunk for level 1, lib/brotli-mt_compress.c:105-109), written
    by the reference build; one step = gpumt_brotli_decompress_batch over all records of the rank,
    input and output resident in HBM.  value = uncompressed MB/s (decompress only: there is no
    compression on this path)."""
    import struct
    L, h = eng.L, eng.h
    chunk = args.chunk or (1 << 20)
    args.chunk = chunk
    n = int(args.gib * (1 << 30)) // chunk * chunk
    base_n = min(n, 1 << 30) // chunk * chunk          # compressed once on the host, replicated in HBM
    reps = (n + base_n - 1) // base_n
    n = base_n * reps
    T = tools()
    threads = min(os.cpu_count() or 1, 128) // max(1, world) or 1
    t0 = time.time()
    text = np.empty(base_n, np.uint8)
    T.zmt_gen_text(text.ctypes.data, base_n, SEED, rank * base_n, threads)
    stream = reference_brotli_stream(text, chunk, 1, threads)
    # record table (what the host engine parses while reading, lib/brotli-mt_decompress.c:187-284)
    ro, rl, cap = [], [], []
    ip = 0
    while ip < len(stream):
        magic, eight, csize, br, hint = struct.unpack_from("<IIIHH", stream, ip)
        assert magic == 0x184D2A50 and eight == 8 and br == 0x5242
        ro.append(ip + 16)
        rl.append(csize)
        cap.append(hint << 16)
        ip += 16 + csize
    gen_s = time.time() - t0
    nb = len(ro)
    nrec = nb * reps
    seg = (len(stream) + 511) & ~255                     # replica stride in HBM
    rec_off = np.concatenate([np.asarray(ro, np.uint64) + np.uint64(r * seg) for r in range(reps)])
    rec_len = np.tile(np.asarray(rl, np.uint32), reps)
    out_cap = np.tile(np.asarray(cap, np.uint32), reps)
    out_off = np.zeros(nrec + 1, np.uint64)
    out_off[1:] = np.cumsum(out_cap.astype(np.uint64))
    total_cap = int(out_off[nrec])
    d_stream = eng.alloc(seg * reps + 512)
    hs = np.frombuffer(stream, np.uint8)
    for r in range(reps):
        eng._ck(L.gpumt_memcpy_h2d(h, d_stream.ptr + r * seg, hs.ctypes.data, hs.nbytes, 0), "h2d")
    eng.sync(0)
    d_ro, d_rl, d_oo, d_oc = eng.upload(rec_off), eng.upload(rec_len), eng.upload(out_off), eng.upload(out_cap)
    d_ol, d_st = eng.alloc(nrec * 4), eng.alloc(nrec * 4)
    d_out = eng.alloc(total_cap + 64)

    def step():
        eng.timer_start(3)
        eng.brotli_decompress(d_stream, d_ro, d_rl, nrec, d_out, d_oo, d_oc, d_ol, d_st)
        eng.timer_stop(3)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()
        eng.sync()

    for _ in range(args.warmup):
        step()
    barrier()
    acc = {"decompress": 0.0, "k_brotli_dec": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        eng.sync(0)
        acc["decompress"] += eng.timer_ms(3)
        acc["k_brotli_dec"] += eng.timer_ms(12)
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        tw = torch.tensor([wall], device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    ms = {k: v / args.steps for k, v in acc.items()}
    status = eng.download(d_st, nrec * 4, np.uint32)
    olen = eng.download(d_ol, nrec * 4, np.uint32)
    bad = int((status != 0).sum())
    U = float(olen.astype(np.uint64).sum())
    # content check of the first replica (always: it is 1 GiB at most)
    ok = bad == 0 and int(U) == n
    if ok:
        got = eng.download(d_out, base_n)              # capacities == chunk sizes for full chunks
        ok = bool((got == text).all()) if int(out_off[nb]) == base_n else None
    # ---- the device encoder on the same text (not part of `value`: configs[4] is decompress) ----
    own = None
    try:
        if args.no_encoder:
            raise RuntimeError("skipped (--no-encoder)")
        d_text = eng.alloc(n + 64)
        for r in range(reps):
            eng._ck(L.gpumt_memcpy_h2d(h, d_text.ptr + r * base_n, text.ctypes.data, base_n, 0), "h2d")
        eng.sync(0)
        stride = eng.zstd_slot_stride(chunk)
        nrec_c = n // chunk
        d_slots = eng.alloc(nrec_c * stride)
        d_cl, d_co = eng.alloc(nrec_c * 4), eng.alloc((nrec_c + 1) * 8)
        d_cs = eng.alloc(nrec_c * stride)
        for it in range(2):
            eng.timer_start(1)
            eng.brotli_compress(d_text, n, chunk, d_slots, stride, d_cl)
            eng.lz4_compact(d_slots, stride, d_cl, nrec_c, d_cs, d_co)
            eng.timer_stop(1)
            eng.sync(0)
        t_enc = eng.timer_ms(1)
        c_own = int(eng.download(d_co, 8, np.uint64, offset=nrec_c * 8)[0])
        # decode the device encoder's streams
        crl = eng.download(d_cl, nrec_c * 4, np.uint32)
        cro = eng.download(d_co, nrec_c * 8, np.uint64) + np.uint64(16)
        d_ro2, d_rl2 = eng.upload(cro), eng.upload((crl - 16).astype(np.uint32))
        cap2 = np.full(nrec_c, chunk, np.uint32)
        oo2 = np.arange(nrec_c + 1, dtype=np.uint64) * np.uint64(chunk)
        d_oo2, d_oc2 = eng.upload(oo2), eng.upload(cap2)
        for it in range(2):
            eng.timer_start(2)
            eng.brotli_decompress(d_cs, d_ro2, d_rl2, nrec_c, d_out, d_oo2, d_oc2, d_ol, d_st)
            eng.timer_stop(2)
            eng.sync(0)
        t_dec = eng.timer_ms(2)
        st2 = eng.download(d_st, nrec_c * 4, np.uint32)
        ok2 = bool((st2 == 0).all()) and bool((eng.download(d_out, base_n) == text).all())
        own = {"what": "zmt_brotli_enc_kernel(+assemble+compact) on the same text, and the decode of its streams",
               "compress_ms": round(t_enc, 3), "compress_MBps": round(world * n / 1e6 / (t_enc * 1e-3), 1),
               "ratio": round(n / c_own, 4), "decompress_ms": round(t_dec, 3),
               "decompress_MBps": round(world * n / 1e6 / (t_dec * 1e-3), 1), "roundtrip_verified": ok2}
    except Exception as e:  # report, never hide
        own = {"error": repr(e)}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    Cb = float(sum(rl) + 16 * nb) * reps
    alg = U + Cb
    step_s = wall / args.steps
    t_k = ms["k_brotli_dec"] * 1e-3
    a = alg / t_k / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf) and abs(args.gib - 8.0) < 1e-9 and chunk == 1 << 20:
        with open(tf) as f:
            traffic = json.load(f).get("per_launch_bytes_8gib", {}).get("zmt_brotli_dec_kernel")
    res = {
        "metric": "MB/s decompress, 8 GiB synthetic, brotli-mt (level-1 streams); % HBM roofline",
        "value": round(world * U / 1e6 / step_s, 1), "unit": "MB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"brotli-mt decompress, {n / (1 << 30):g} GiB enwik-style synthetic per GPU, "
                               f"{chunk // 1024} KiB chunks compressed at level 1 by the reference build "
                               f"({base_n >> 20} MiB compressed on the host, replicated x{reps} in HBM), "
                               f"device-resident decode",
                   "chunk": chunk, "records_per_gpu": nrec, "level": 1, "ratio": round(U / Cb, 4),
                   "parallelism": f"record-sharded x{world}"},
        "decompress_MBps": round(world * U / 1e6 / (ms["decompress"] * 1e-3), 1),
        "roofline": {"kernel": "zmt_brotli_dec_kernel", "bound": "hbm", "achieved": round(a, 2),
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(a * 1e9 / HBM_PEAK, 5),
                     "frac_of_copy_ceiling": round(a * 1e9 / HBM_COPY, 5), "alg_bytes_per_launch": alg,
                     "avg_launch_ms": round(ms["k_brotli_dec"], 4), "traffic": traffic},
        "kernels": {"k_brotli_dec": {"ms": round(ms["k_brotli_dec"], 4)}},
        "device_encoder": own,
        "decode_errors": bad, "roundtrip_verified": ok,
        "gen_s": round(gen_s, 2), "device": eng.name,
    }
    if not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def rccl_gather(eng, dist, d_stream, sizes, rank, world):
    """gatherv of the per-rank compressed segments to rank 0 over RCCL (zstdmt_amd.shard), timed."""
    import torch
    from zstdmt_amd.shard import gather_segments
    mine = int(sizes[rank])
    # RCCL wants torch tensors: stage the segment into a torch-owned buffer (D2D copy, untimed)
    seg = torch.empty(mine, dtype=torch.uint8, device="cuda")
    eng._ck(eng.L.gpumt_memcpy_d2d(eng.h, seg.data_ptr(), d_stream.ptr, mine, 0), "d2d")
    eng.sync(0)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gather_segments(seg, sizes, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    return round((time.perf_counter() - t0) * 1e3, 3)


if __name__ == "__main__":
    main()
