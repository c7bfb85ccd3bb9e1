#!/usr/bin/env python3
"""bench.py -- the lz4-mt / zstd-mt / brotli-mt hot path on MI355X, device-resident, with roofline.

Default run (what the driver times): BASELINE.json configs[1] -- lz4-mt level 1, 8 GiB enwik-style
synthetic text, 128 KiB chunks (65 536 records).  One "step" = one full pass of the hot path over the
buffer, inputs already resident in HBM:

    compress_batch (XXH32 of every chunk + bit-exact LZ4 frame encode into per-chunk slots)
    -> compact (scan of record sizes + ordered pack into the MT stream)
    -> probe_sizes (content-size fields + scan)
    -> decompress_batch (frame decode + XXH32 content-checksum verification)

`value` = uncompressed MB (1e6 B) round-tripped per second, whole job (all ranks).  Per-direction
rates, per-kernel HIP-event times and the HBM roofline of the kernels ride along, and -- on the
default single-GPU run -- short legs of the other BASELINE configs under "configs": zstd-mt level 1
(configs[3]), brotli-mt decompress (configs[4]) and the PCIe-inclusive drop-in APIs (LZ4MT_* /
ZSTDCB_* / BROTLIMT_* with memcpy callbacks), each with its own roofline / cpu_baseline.

Multi-GPU (`--gpus N`): bench.py starts N ranks itself (torch.distributed.run, one process per GPU,
RCCL) unless it already runs under a launcher (RANK in the environment, which is how the driver
starts it).  Chunks are independent, so rank r takes the contiguous chunk range shard_range(...) of
ONE `--gib` buffer (`--scaling strong`, the default: BASELINE's metric is one 8 GiB job at 1/2/4/8
GPUs) or its own `--gib` buffer (`--scaling weak`).  The only exchange on the path is the all-gather
of the per-rank segment sizes (-> byte offsets of the segments in the final stream, included in the
timed region); `--gather rccl|d2h` additionally times the reassembly of the
segments (grouped send/recv gatherv to rank 0, or every GPU's own copy into one shared pinned host buffer) and reports
it as gather_ms / value_with_gather, never as `value`; `per_rank_ms` carries every rank's kernel times.
`--mode decompress` = configs[2] (decompress only: the records are written untimed, then timed).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12        # B/s, MI355X spec (MI355X_MICROARCH.md)
HBM_COPY = 6.29e12       # measured float4 copy ceiling, same guide
SEED = 20260926
TRAFFIC_FILE = os.path.join("profiles", "pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=8.0,
                    help="uncompressed GiB: of the whole job (--scaling strong) or per GPU (weak)")
    ap.add_argument("--codec", choices=("lz4", "zstd", "brotli", "snappy"), default="lz4",
                    help="lz4 = BASELINE configs[1] (the metric's config); zstd = configs[3], zstd-mt level 1; "
                         "brotli = configs[4], brotli-mt decompress of level-1 streams at 1 MiB chunks; "
                         "snappy = snappy-mt round trip at its default 64 KiB chunks (no BASELINE config, one GPU)")
    ap.add_argument("--mode", choices=("roundtrip", "decompress"), default="roundtrip",
                    help="decompress = BASELINE configs[2]: only the decompress leg is timed")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--chunk", type=int, default=0, help="0 = 128 KiB for lz4 (configs[1]), 1 MiB for zstd / brotli")
    ap.add_argument("--dec-variant", type=int, default=0,
                    help="lz4 decoder: 0 = frames + parse4 + copy3 (default), 1 = frame-serial")
    ap.add_argument("--lz4-ring", type=int, default=12, help="log2 of copy3's LDS ring per wave (12..14)")
    ap.add_argument("--snappy-dec", type=int, default=0,
                    help="--codec snappy: 1 = the batched decoder (zmt_snappy_dec2_kernel), 0 = element by element")
    ap.add_argument("--zstd-level", type=int, default=1,
                    help="--codec zstd: level handed to the device encoder (tiers 1-2 / 3-9 / 10-22); BASELINE configs[3] is 1")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-mib", type=int, default=2048, help="cpu_baseline sample size")
    ap.add_argument("--gather", nargs="?", const="rccl", default="none", choices=("none", "rccl", "d2h"),
                    help="also time the reassembly of the per-rank segments (never part of `value`): rccl = grouped "
                         "send/recv gatherv to rank 0 over xGMI; d2h = every GPU copies its segment to its offset of ONE "
                         "pinned host buffer shared by the ranks, no inter-GPU traffic (SURVEY 8e)")
    ap.add_argument("--no-verify", dest="verify", action="store_false",
                    help="skip the byte-for-byte comparison of the round trip (on by default)")
    ap.add_argument("--only", action="store_true", help="main leg only: no zstd / brotli / api legs under 'configs'")
    ap.add_argument("--extra-gib", type=float, default=8.0,
                    help="size of the extra legs (zstd / brotli / snappy) of the default run: BASELINE's 8 GiB")
    ap.add_argument("--api-gib", type=float, default=8.0, help="input size of the drop-in API legs of the default run")
    ap.add_argument("--no-encoder", action="store_true",
                    help="--codec brotli: skip the device-encoder leg (profiling runs of the decoder alone)")
    ap.add_argument("--zref-only", action="store_true",
                    help="developer: only the 'zstd-mt decompress of reference-written streams' leg, printed as JSON")
    ap.add_argument("--zstd-seq", type=int, default=0, help="1 = no sequence pre-pass in front of the zstd frame decoder")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, report who runs where (gloo, no GPU touched) and exit")
    return ap.parse_args()


def tools():
    t = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
    t.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
    return t


def spawn(args):
    """--gpus N without a launcher: start N ranks of this script on this node (one per GPU)."""
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(codec, chunk, cpu_mib):
    """oracle/_ref (reference sources + the image's codec library) if present, else the oracle port;
    bounded sample, rank 0 at N=1 only."""
    exe = os.path.join(ROOT, "oracle", "cpu_bench")
    zstd = codec == "zstd"
    brotli = codec == "brotli"
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
                "error": f"{os.path.basename(ref)} not present (the {codec} oracle has no compressor)"}
    n = cpu_mib << 20
    try:
        out = subprocess.check_output([exe, "reference-brotli" if brotli else "reference-zstd" if zstd else kind,
                                       ref if kind == "reference" else "-", str(n),
                                       str(chunk), str(threads), str(SEED)], timeout=600)
        r = json.loads(out)
    except Exception as e:  # report, never hide
        return {"value": None, "unit": "MB/s", "cores": threads, "kind": kind, "error": repr(e)}
    if brotli:
        return {"value": r["decompress_MBps"], "unit": "MB/s", "cores": threads, "kind": kind,
                "compress_MBps": r["compress_MBps"], "decompress_MBps": r["decompress_MBps"], "host_cpus": cores,
                "sample": f"{cpu_mib} MiB of the same synthetic text, {chunk}-byte chunks, level 1: "
                          f"BROTLIMT_decompressDCtx of the stream BROTLIMT_compressCCtx wrote, memcpy "
                          f"callbacks, T={threads}"}
    # the reference at T = 1 and 4 as well (configs[0] is `-T4`), on a small sample
    by_threads = {}
    if kind == "reference" and not zstd:
        for t, mib in ((1, 128), (4, 256)):
            try:
                o = json.loads(subprocess.check_output([exe, kind, ref, str(mib << 20), str(chunk), str(t), str(SEED)],
                                                       timeout=120))
                by_threads[str(t)] = {"compress_MBps": o["compress_MBps"], "decompress_MBps": o["decompress_MBps"],
                                      "sample_MiB": mib}
            except Exception as e:  # report, never hide
                by_threads[str(t)] = {"error": repr(e)}
    return {"value": r["roundtrip_MBps"], "unit": "MB/s", "cores": threads, "kind": kind,
            "compress_MBps": r["compress_MBps"], "decompress_MBps": r["decompress_MBps"],
            "host_cpus": cores, "by_threads": by_threads,
            "sample": f"{cpu_mib} MiB of the same synthetic text, {chunk}-byte chunks, "
                      f"{'ZSTDCB' if zstd else 'LZ4MT'}_compressCCtx+decompressDCtx (level 1) with memcpy "
                      f"callbacks, T={threads}"}


class Ctx:
    """what every leg needs: engine, rank layout, process group"""

    def __init__(self, args, eng, rank, world, dist):
        self.args, self.eng, self.rank, self.world, self.dist = args, eng, rank, world, dist

    def barrier(self):
        self.eng.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.eng.sync()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_floats(self, vals):
        """-> one list of floats per rank (on every rank)"""
        if self.dist is None:
            return [list(vals)]
        import torch
        mine = torch.tensor(vals, dtype=torch.float64, device="cuda")
        out = torch.zeros(self.world * len(vals), dtype=torch.float64, device="cuda")
        self.dist.all_gather_into_tensor(out, mine)
        return out.view(self.world, len(vals)).tolist()

    def all_true(self, ok):
        return self.sum_over_ranks(0.0 if ok else 1.0) == 0.0


def traffic_table(gib, chunk, want_chunk):
    """PMC HBM traffic per launch from the committed summary of separate rocprofv3 --pmc passes
    (tools/profile_round.sh): valid for the 8 GiB single-GPU shape only."""
    tf = os.path.join(ROOT, TRAFFIC_FILE)
    if os.path.exists(tf) and abs(gib - 8.0) < 1e-9 and chunk == want_chunk:
        with open(tf) as f:
            return json.load(f).get("per_launch_bytes_8gib", {})
    return {}


def shard(args, world, rank, chunk):
    """-> (bytes of this rank, offset of this rank in the synthetic corpus, total bytes of the job)"""
    total = int(args.gib * (1 << 30)) // chunk * chunk
    if world == 1:
        return total, 0, total
    if args.scaling == "weak":
        return total, rank * total, total * world
    from zstdmt_amd.shard import shard_range
    lo, hi = shard_range(total // chunk, rank, world)
    return (hi - lo) * chunk, lo * chunk, total


def generate(ctx, n, corpus_off):
    """synthetic text, generated on the host in 256 MiB pieces, uploaded once"""
    eng, L = ctx.eng, ctx.eng.L
    d_in = eng.alloc(n + 64)
    piece = 256 << 20
    hbuf = np.empty(min(piece, max(n, 1)), np.uint8)
    T = tools()
    gen_threads = max(1, (os.cpu_count() or 1) // max(1, ctx.world))
    t0 = time.time()
    for off in range(0, n, piece):
        m = min(piece, n - off)
        T.zmt_gen_text(hbuf.ctypes.data, m, SEED, corpus_off + off, gen_threads)
        eng._ck(L.gpumt_memcpy_h2d(eng.h, d_in.ptr + off, hbuf.ctypes.data, m, 0), "h2d")
        eng.sync(0)
    return d_in, time.time() - t0


def bench_lz4_zstd(ctx, codec, gib_args=None, steps=None, warmup=None, cpu=True, main=True, dec_only=None):
    """configs[1] (lz4), configs[2] (lz4 --mode decompress), configs[3] (zstd)"""
    args, eng, rank, world, dist = ctx.args, ctx.eng, ctx.rank, ctx.world, ctx.dist
    zstd = codec == "zstd"
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    chunk = args.chunk if (args.chunk and main) else ((1 << 20) if zstd else 131072)
    saved_gib = args.gib
    if gib_args is not None:
        args.gib = gib_args
    n, corpus_off, total_n = shard(args, world, rank, chunk)
    gib = args.gib
    args.gib = saved_gib
    dec_only = (args.mode == "decompress" and main) if dec_only is None else dec_only
    nrec = eng.record_count(n, chunk)
    stride = eng.zstd_slot_stride(chunk) if zstd else eng.slot_stride(chunk)

    d_in, gen_s = generate(ctx, n, corpus_off)
    d_slots = eng.alloc(nrec * stride)
    d_rl = eng.alloc(nrec * 4)
    d_ro = eng.alloc((nrec + 1) * 8)
    d_stream = eng.alloc(nrec * stride)       # worst case
    d_ol = eng.alloc(nrec * 4)
    d_oo = eng.alloc((nrec + 1) * 8)
    d_st = eng.alloc(nrec * 4)
    d_out = eng.alloc(n + 64)
    bufs = [d_in, d_slots, d_rl, d_ro, d_stream, d_ol, d_oo, d_st, d_out]

    def compress():
        eng.timer_start(1)
        if zstd:
            eng.zstd_compress(d_in, n, chunk, d_slots, stride, d_rl, level=args.zstd_level if main else 1)
        else:
            eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl)
        eng.timer_stop(1)
        eng.timer_start(2)
        eng.lz4_compact(d_slots, stride, d_rl, nrec, d_stream, d_ro)
        eng.timer_stop(2)

    def decompress():
        eng.timer_start(3)
        if zstd:
            eng.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
            eng.zstd_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
        else:
            eng.lz4_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo)
            eng.lz4_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
        eng.timer_stop(3)

    seg_off = [0]

    def step():
        if not dec_only:
            compress()
        decompress()
        if dist is not None and not dec_only:
            # frame reassembly: every rank learns the byte offset of its segment in the final stream
            from zstdmt_amd.shard import exchange_segment_sizes
            eng.sync(0)
            total_c = int(eng.download(d_ro, 8, np.uint64, offset=nrec * 8)[0])
            _sizes, seg_off[0] = exchange_segment_sizes(total_c, device="cuda")

    if dec_only:
        compress()                     # the records to decode: written once, untimed
        eng.sync()
    for _ in range(warmup):
        step()
    ctx.barrier()
    acc = {}
    slots = (("compress", 1), ("compact", 2), ("decompress", 3), ("k_lz4_enc", 9),
             ("k_scan_compact", 10), ("k_lz4_dec", 11))
    if not zstd:
        slots += (("k_xxh32_c", 8), ("k_xxh32_d", 12), ("k_dec_frames", 13), ("k_dec_parse", 14),
                  ("k_dec_copy", 15))
    if dec_only:
        slots = tuple(s for s in slots if s[0] not in ("compress", "compact", "k_lz4_enc", "k_scan_compact", "k_xxh32_c"))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        eng.sync(0)
        # slots 1-3: API-level legs; 8..15: individual kernels (HIP events on the launching stream)
        for name, slot in slots:
            acc[name] = acc.get(name, 0.0) + eng.timer_ms(slot)
    ctx.barrier()
    wall = ctx.max_over_ranks(time.perf_counter() - t0)

    ms = {k: v / steps for k, v in acc.items()}
    total_c = int(eng.download(d_ro, 8, np.uint64, offset=nrec * 8)[0])
    status = eng.download(d_st, nrec * 4, np.uint32)
    bad = int(ctx.sum_over_ranks(float((status != 0).sum())))
    ok = None
    if args.verify:
        ok = ctx.all_true(eng.equal(d_in, d_out, n))

    gather_ms = None
    if dist is not None and args.gather != "none" and main:
        from zstdmt_amd.shard import exchange_segment_sizes
        sizes, _ = exchange_segment_sizes(n if dec_only else total_c, device="cuda")
        fn = rccl_gather if args.gather == "rccl" else d2h_gather
        gather_ms = fn(eng, dist, d_out if dec_only else d_stream, sizes, rank, world)
    Cb_all = ctx.sum_over_ranks(float(total_c))
    # per-rank kernel times (ms per step): what each GPU spent, next to the max-over-ranks wall clock
    by_rank = ctx.gather_floats([ms.get("k_lz4_enc", 0.0), ms.get("k_lz4_dec", 0.0), ms.get("compress", 0.0) +
                                 ms.get("compact", 0.0), ms.get("decompress", 0.0)])
    for b in bufs:
        b.free()
    if rank != 0:
        return None

    U = float(n)                       # this rank's bytes: kernel times below are this rank's
    Cb = float(total_c)
    alg = U + Cb                       # algorithmic bytes either direction
    U_all = float(total_n)
    step_s = wall / steps
    t_d = ms["decompress"] * 1e-3
    t_c = (ms["compress"] + ms["compact"]) * 1e-3 if not dec_only else None
    want_chunk = (1 << 20) if zstd else 131072
    traffic = traffic_table(gib if world == 1 else -1, chunk, want_chunk)
    split = args.dec_variant == 0

    def roof(kname, t_ms, alg_bytes, pmc_names):
        t = t_ms * 1e-3
        a = alg_bytes / t / 1e9 if t > 0 else 0.0
        tr = [traffic.get(k) for k in pmc_names if traffic.get(k) is not None]
        return {"kernel": kname, "bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(a * 1e9 / HBM_PEAK, 5), "frac_of_copy_ceiling": round(a * 1e9 / HBM_COPY, 5),
                "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": round(t_ms, 4),
                "traffic": sum(tr) if tr else None,
                "traffic_source": (TRAFFIC_FILE + " (separate rocprofv3 --pmc passes of this command, not this run)")
                if tr else None}

    kern = {k: {"ms": round(v, 4)} for k, v in ms.items() if k.startswith("k_")}
    if zstd:
        zl = args.zstd_level if main else 1
        name, what = f"zstd-mt level {zl}", f"zstd-mt -{zl}"
        zk = ("zmt_zstd_enc_kernel", "zmt_zstd_enc_t2_kernel", "zmt_zstd_enc_t3_kernel")[eng.L.gpumt_zstd_level_tier(zl)]
        r_enc = None if dec_only else roof(zk + "(+assemble)", ms["k_lz4_enc"], alg, (zk, "zmt_zstd_assemble_kernel"))
        r_dec = roof("zmt_zstd_dec_small_kernel(+zmt_zstd_dec_kernel for frames with full-size tables)",
                     ms["k_lz4_dec"], alg, ("zmt_zstd_seq_kernel", "zmt_zstd_dec_small_kernel", "zmt_zstd_dec_kernel"))
    else:
        name, what = "lz4-mt", "lz4-mt -1"
        # (GPUMT_LZ4_ENC=3 selects round 5's probe-batch encoder, lz4_enc3.hip; the default is the window encoder, lz4_enc5.hip)
        ev = "3" if os.environ.get("GPUMT_LZ4_ENC", "") == "3" else "5"
        ek = (f"zmt_lz4_enc{ev}_p17_kernel" if 65536 < chunk <= 131072 else
              (f"zmt_lz4_enc{ev}_u16_kernel" if chunk <= 65536 else f"zmt_lz4_enc{ev}_u32_kernel"))
        r_enc = None if dec_only else roof(ek, ms["k_lz4_enc"], alg, (ek,))
        k_parse = "zmt_dec_parse4_kernel"
        k_copy = "zmt_dec_copy3_w%d_kernel" % (1 << (args.lz4_ring - 10))
        r_dec = roof(f"zmt_dec_frames_kernel + {k_parse} + {k_copy}" if split else "zmt_lz4_dec_serial", ms["k_lz4_dec"], alg,
                     ("zmt_dec_frames_kernel", k_parse, k_copy) if split else ())
    dom = r_dec if (dec_only or ms["k_lz4_dec"] >= ms.get("k_lz4_enc", 0.0)) else r_enc
    res = {
        "metric": (f"MB/s decompress, {U_all / (1 << 30):g} GiB synthetic, {name}; % HBM roofline" if dec_only else
                   f"MB/s compress+decompress, {U_all / (1 << 30):g} GiB synthetic, {name}; % HBM roofline"),
        "value": round(U_all / 1e6 / step_s, 1),
        "unit": "MB/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(step_s * 1e3, 3),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{what}{' decompress-only' if dec_only else ''}, {U_all / (1 << 30):g} GiB enwik-style "
                               f"synthetic ({'whole job' if args.scaling == 'strong' or world == 1 else 'per GPU x ' + str(world)}), "
                               f"{chunk // 1024} KiB chunks, device-resident",
                   "chunk": chunk, "records_per_gpu": nrec, "records": int(total_n // chunk),
                   "level": (args.zstd_level if main else 1) if zstd else 1,
                   "ratio": round(ctx_ratio(U_all, Cb_all), 4), "dec_variant": args.dec_variant,
                   "lz4_ring": args.lz4_ring if args.dec_variant == 0 else None,
                   "parallelism": f"chunk-sharded x{world}"},
        "decompress_MBps": round(U / 1e6 / t_d * world, 1),
        "roofline": dom,
        "roofline_decompress": r_dec,
        "roofline_decompress_path": {
            "what": "probe + decode + checksum-verify kernels together (HIP events around the leg)",
            "achieved": round(alg / t_d / 1e9, 2), "unit": "GB/s", "frac": round(alg / t_d / HBM_PEAK, 5)},
        "kernels": kern,
        "decode_errors": bad, "roundtrip_verified": ok,
        "gen_s": round(gen_s, 2), "device": eng.name,
        "segment_offset_rank0": seg_off[0], "gather": args.gather, "gather_ms": gather_ms,
        "per_rank_ms": [{"rank": r, "k_enc": round(v[0], 3), "k_dec": round(v[1], 3), "compress_leg": round(v[2], 3),
                         "decompress_leg": round(v[3], 3)} for r, v in enumerate(by_rank)],
    }
    if not dec_only:
        res["compress_MBps"] = round(U / 1e6 / t_c * world, 1)
        res["roofline_compress"] = r_enc
    if zstd:
        res["config"]["parity"] = "decompress-identical (SURVEY 8a C4)"
    elif split:
        res["roofline_decompress_copy_kernel"] = roof(k_copy, ms["k_dec_copy"], alg, (k_copy,))
        res["roofline_decompress_parse_kernel"] = roof(k_parse, ms["k_dec_parse"], Cb, (k_parse,))
    if gather_ms is not None:
        res["value_with_gather"] = round(U_all / 1e6 / (step_s + gather_ms * 1e-3), 1)
    if cpu and not args.no_cpu and world == 1:
        res["cpu_baseline"] = cpu_baseline(codec, chunk, args.cpu_mib)
    return res


def ctx_ratio(u, c):
    return u / c if c else 0.0


def reference_stream(data, chunk, level, threads, so="libbrotlimt_ref.so", pfx="BROTLIMT"):
    """The workload's input: `data` compressed by the REFERENCE's brotli-mt / zstd-mt (oracle/_ref, SURVEY 8d
    "cfg5: the same text compressed by own/oracle brotli at 1 MiB chunks").  Input preparation on the
    host, outside the timed region; the device encoder's streams are timed as well (device_encoder)."""
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", so))

    class Buf(C.Structure):
        _fields_ = [("buf", C.c_void_p), ("size", C.c_size_t), ("allocated", C.c_size_t)]
    FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Buf))

    class RdWr(C.Structure):
        _fields_ = [("fn_read", FN), ("arg_read", C.c_void_p), ("fn_write", FN), ("arg_write", C.c_void_p)]
    create, compress = getattr(lib, pfx + "_createCCtx"), getattr(lib, pfx + "_compressCCtx")
    free, is_error = getattr(lib, pfx + "_freeCCtx"), getattr(lib, pfx + "_isError")
    create.restype = C.c_void_p
    create.argtypes = [C.c_int, C.c_int, C.c_int]
    compress.restype = C.c_size_t
    compress.argtypes = [C.c_void_p, C.POINTER(RdWr)]
    free.argtypes = [C.c_void_p]
    is_error.argtypes = [C.c_size_t]
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
    ctx = create(threads, level, chunk)
    rv = compress(ctx, C.byref(io))
    free(ctx)
    if is_error(rv):
        raise RuntimeError("reference %s compress failed" % pfx)
    return b"".join(out)


def reference_brotli_stream(data, chunk, level, threads):
    return reference_stream(data, chunk, level, threads)


def bench_zstd_ref(ctx, gib, steps, warmup):
    """zstd-mt decompress of streams the REFERENCE wrote (lib/zstd-mt_compress.c + libzstd at level 1, 1 MiB
    chunks): what a drop-in decoder meets in the wild.  1 GiB is compressed on the host by
    oracle/_ref/libzstdmt_ref.so, outside the timed region, and replicated in HBM; one step = zstd_probe +
    gpumt_zstd_decompress_batch over all records; one GPU."""
    import struct
    eng = ctx.eng
    L, h = eng.L, eng.h
    chunk = 1 << 20
    n_all = int(gib * (1 << 30)) // chunk * chunk
    base_n = min(n_all, 1 << 30)
    reps = max(1, n_all // base_n)
    n = base_n * reps
    threads = min(os.cpu_count() or 1, 128)
    t0 = time.time()
    text = np.empty(base_n, np.uint8)
    tools().zmt_gen_text(text.ctypes.data, base_n, SEED, 0, threads)
    stream = reference_stream(text, chunk, 1, threads, "libzstdmt_ref.so", "ZSTDCB")
    ro, rl = [], []
    ip = 0
    while ip < len(stream):
        magic, four, csize = struct.unpack_from("<III", stream, ip)
        assert magic == 0x184D2A50 and four == 4
        ro.append(ip)
        rl.append(12 + csize)
        ip += 12 + csize
    gen_s = time.time() - t0
    nb = len(ro)
    nrec = nb * reps
    seg = (len(stream) + 511) & ~255
    rec_off = np.concatenate([np.asarray(ro, np.uint64) + np.uint64(r * seg) for r in range(reps)])
    rec_len = np.tile(np.asarray(rl, np.uint32), reps)
    d_stream = eng.alloc(seg * reps + 512)
    hs = np.frombuffer(stream, np.uint8)
    for r in range(reps):
        eng._ck(L.gpumt_memcpy_h2d(h, d_stream.ptr + r * seg, hs.ctypes.data, hs.nbytes, 0), "h2d")
    eng.sync(0)
    d_ro, d_rl = eng.upload(rec_off), eng.upload(rec_len)
    d_ol, d_oo, d_st = eng.alloc(nrec * 4), eng.alloc((nrec + 1) * 8), eng.alloc(nrec * 4)
    d_out = eng.alloc(n + 64)
    bufs = [d_stream, d_ro, d_rl, d_ol, d_oo, d_st, d_out]

    def step():
        eng.timer_start(3)
        eng.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
        eng.zstd_decompress(d_stream, seg * reps, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
        eng.timer_stop(3)

    for _ in range(warmup):
        step()
    ctx.barrier()
    acc = {"decompress": 0.0, "k_zstd_dec": 0.0}
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        eng.sync(0)
        acc["decompress"] += eng.timer_ms(3)
        acc["k_zstd_dec"] += eng.timer_ms(11)
    wall = time.perf_counter() - t0
    ms = {k: v / steps for k, v in acc.items()}
    if os.environ.get("ZMT_ZREF_PROF"):
        # developer: per-phase cycles of the frame decoder (its profiling build: the general kernel, 12 waves per CU) on these streams
        import ctypes as C
        cnt = (C.c_ulonglong * 16)()
        eng.set_variant("profile", 5)
        L.gpumt_debug_counters(h, cnt, 16)
        step()
        eng.sync(0)
        L.gpumt_debug_counters(h, cnt, 16)
        eng.set_variant("profile", 1)
        c = list(cnt)
        w = max(c[9], 1)
        nm = ["hdr+huftab", "huffman", "seqhdr+tables", "stage+fse", "exec-lit", "exec-match", "other"]
        print("zref frame decoder, Mcycles per record: " + ", ".join(f"{nm[i]}={c[i] / w / 1e6:.2f}" for i in range(7))
              + f" total={c[8] / w / 1e6:.2f}; leg {eng.timer_ms(11):.2f} ms", file=sys.stderr)
    status = eng.download(d_st, nrec * 4, np.uint32)
    bad = int((status != 0).sum())
    # replica 0 against the text byte for byte, every other replica against replica 0 (device-side XXH32 per MiB)
    ok = bad == 0 and bool((eng.download(d_out, base_n) == text).all()) and eng.replicas_equal(d_out, base_n, reps)
    for b in bufs:
        b.free()
    U, Cb = float(n), float(len(stream)) * reps
    alg = U + Cb
    a = alg / (ms["k_zstd_dec"] * 1e-3) / 1e9
    tt = traffic_table(gib, chunk, chunk)
    tr = [tt[k] for k in ("zref:zmt_zstd_seq_kernel", "zref:zmt_zstd_dec_small_kernel", "zref:zmt_zstd_dec_kernel") if k in tt]
    return {
        "metric": f"MB/s decompress, {U / (1 << 30):g} GiB synthetic, zstd-mt streams written by the reference (level 1); % HBM roofline",
        "value": round(U / 1e6 / (wall / steps), 1), "unit": "MB/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(wall / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"zstd-mt decompress, {U / (1 << 30):g} GiB enwik-style synthetic, 1024 KiB chunks compressed at "
                               f"level 1 by the reference build ({base_n >> 20} MiB compressed on the host, replicated x{reps} "
                               "in HBM), device-resident decode", "chunk": chunk, "records_per_gpu": nrec, "level": 1,
                   "ratio": round(U / Cb, 4)},
        "decompress_MBps": round(U / 1e6 / (ms["decompress"] * 1e-3), 1),
        "roofline": {"kernel": "zmt_zstd_seq_kernel + zmt_zstd_dec_small_kernel (sequence pre-pass, then the frame decoder; "
                               "reference-written frames)", "bound": "hbm", "achieved": round(a, 2),
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(a * 1e9 / HBM_PEAK, 5),
                     "alg_bytes_per_launch": alg, "avg_launch_ms": round(ms["k_zstd_dec"], 4),
                     "traffic": sum(tr) if tr else None,
                     "traffic_source": (TRAFFIC_FILE + " (separate rocprofv3 --pmc passes of this leg, not this run)") if tr else None},
        "kernels": {"k_zstd_dec": {"ms": round(ms["k_zstd_dec"], 4)}},
        "decode_errors": bad, "roundtrip_verified": ok, "gen_s": round(gen_s, 2),
    }


def bench_brotli(ctx, gib_args=None, steps=None, warmup=None, cpu=True, main=True):
    """BASELINE configs[4]: brotli-mt decompress.  Input: level-1 streams of the synthetic text at
    1 MiB chunks (the reference's default chunk for level 1, lib/brotli-mt_compress.c:105-109), written
    by the reference build; one step = gpumt_brotli_decompress_batch over all records of the rank,
    input and output resident in HBM.  value = uncompressed MB/s (decompress only: there is no
    compression on this path).  Ranks take contiguous record ranges of the job (strong) or a
    replica each (weak)."""
    import struct
    args, eng, rank, world, dist = ctx.args, ctx.eng, ctx.rank, ctx.world, ctx.dist
    L, h = eng.L, eng.h
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    chunk = (args.chunk if main else 0) or (1 << 20)
    saved_gib = args.gib
    if gib_args is not None:
        args.gib = gib_args
    n, corpus_off, total_n = shard(args, world, rank, chunk)
    gib = args.gib
    args.gib = saved_gib
    base_n = min(n, 1 << 30) // chunk * chunk          # compressed once on the host, replicated in HBM
    reps = (n + base_n - 1) // base_n
    n = base_n * reps
    T = tools()
    threads = min(os.cpu_count() or 1, 128) // max(1, world) or 1
    t0 = time.time()
    text = np.empty(base_n, np.uint8)
    T.zmt_gen_text(text.ctypes.data, base_n, SEED, corpus_off, threads)
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
    bufs = [d_stream, d_ro, d_rl, d_oo, d_oc, d_ol, d_st, d_out]

    def step():
        eng.timer_start(3)
        eng.brotli_decompress(d_stream, d_ro, d_rl, nrec, d_out, d_oo, d_oc, d_ol, d_st)
        eng.timer_stop(3)

    for _ in range(warmup):
        step()
    ctx.barrier()
    acc = {"decompress": 0.0, "k_brotli_dec": 0.0}
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        eng.sync(0)
        acc["decompress"] += eng.timer_ms(3)
        acc["k_brotli_dec"] += eng.timer_ms(12)
    ctx.barrier()
    wall = ctx.max_over_ranks(time.perf_counter() - t0)
    ms = {k: v / steps for k, v in acc.items()}
    status = eng.download(d_st, nrec * 4, np.uint32)
    olen = eng.download(d_ol, nrec * 4, np.uint32)
    bad = int((status != 0).sum())
    U = float(olen.astype(np.uint64).sum())
    # content check of the first replica (always: it is 1 GiB at most)
    ok = bad == 0 and int(U) == n
    if ok:
        got = eng.download(d_out, base_n)              # capacities == chunk sizes for full chunks
        ok = bool((got == text).all()) if int(out_off[nb]) == base_n else None
        if ok:                                         # ... and every other replica against replica 0
            ok = eng.replicas_equal(d_out, base_n, reps)
    ok_all = ctx.all_true(bool(ok)) if ok is not None else None
    bad_all = int(ctx.sum_over_ranks(float(bad)))
    # ---- the device encoder on the same text (not part of `value`: configs[4] is decompress) ----
    own = None
    try:
        if args.no_encoder or not main or world > 1:
            raise RuntimeError("skipped")
        d_text = eng.alloc(n + 64)
        bufs.append(d_text)
        for r in range(reps):
            eng._ck(L.gpumt_memcpy_h2d(h, d_text.ptr + r * base_n, text.ctypes.data, base_n, 0), "h2d")
        eng.sync(0)
        stride = eng.zstd_slot_stride(chunk)
        nrec_c = n // chunk
        d_slots = eng.alloc(nrec_c * stride)
        d_cl, d_co = eng.alloc(nrec_c * 4), eng.alloc((nrec_c + 1) * 8)
        d_cs = eng.alloc(nrec_c * stride)
        bufs += [d_slots, d_cl, d_co, d_cs]
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
        bufs += [d_ro2, d_rl2, d_oo2, d_oc2]
        for it in range(2):
            eng.timer_start(2)
            eng.brotli_decompress(d_cs, d_ro2, d_rl2, nrec_c, d_out, d_oo2, d_oc2, d_ol, d_st)
            eng.timer_stop(2)
            eng.sync(0)
        t_dec = eng.timer_ms(2)
        st2 = eng.download(d_st, nrec_c * 4, np.uint32)
        ok2 = (bool((st2 == 0).all()) and bool((eng.download(d_out, base_n) == text).all())
               and eng.replicas_equal(d_out, base_n, reps))
        own = {"what": "zmt_brotli_enc_kernel(+assemble+compact) on the same text, and the decode of its streams",
               "compress_ms": round(t_enc, 3), "compress_MBps": round(n / 1e6 / (t_enc * 1e-3), 1),
               "ratio": round(n / c_own, 4), "decompress_ms": round(t_dec, 3),
               "decompress_MBps": round(n / 1e6 / (t_dec * 1e-3), 1), "roundtrip_verified": ok2}
    except Exception as e:  # report, never hide
        own = {"skipped": True} if str(e) == "skipped" else {"error": repr(e)}
    U_all = ctx.sum_over_ranks(U)
    for b in bufs:
        b.free()
    if rank != 0:
        return None
    Cb = float(sum(rl) + 16 * nb) * reps
    alg = U + Cb
    step_s = wall / steps
    t_k = ms["k_brotli_dec"] * 1e-3
    a = alg / t_k / 1e9
    tt = traffic_table(gib if world == 1 else -1, chunk, 1 << 20)
    tr = [tt[k] for k in ("zmt_brotli_dec4_kernel", "zmt_brotli_dec_kernel") if tt.get(k) is not None]
    traffic = sum(tr) if tr else None
    res = {
        "metric": f"MB/s decompress, {U_all / (1 << 30):g} GiB synthetic, brotli-mt (level-1 streams); % HBM roofline",
        "value": round(U_all / 1e6 / step_s, 1), "unit": "MB/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(step_s * 1e3, 3),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"brotli-mt decompress, {U_all / (1 << 30):g} GiB enwik-style synthetic "
                               f"({'whole job' if args.scaling == 'strong' or world == 1 else 'per GPU x ' + str(world)}), "
                               f"{chunk // 1024} KiB chunks compressed at level 1 by the reference build "
                               f"({base_n >> 20} MiB compressed on the host per rank, replicated x{reps} in HBM), "
                               f"device-resident decode",
                   "chunk": chunk, "records_per_gpu": nrec, "level": 1, "ratio": round(U / Cb, 4),
                   "parallelism": f"record-sharded x{world}"},
        "decompress_MBps": round(U / 1e6 / (ms["decompress"] * 1e-3) * world, 1),
        "roofline": {"kernel": "zmt_brotli_dec4_kernel (+ zmt_brotli_dec_kernel for the records it hands over)", "bound": "hbm",
                     "achieved": round(a, 2),
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(a * 1e9 / HBM_PEAK, 5),
                     "frac_of_copy_ceiling": round(a * 1e9 / HBM_COPY, 5), "alg_bytes_per_launch": alg,
                     "avg_launch_ms": round(ms["k_brotli_dec"], 4), "traffic": traffic,
                     "traffic_source": (TRAFFIC_FILE + " (separate rocprofv3 --pmc passes, not this run)")
                     if traffic else None},
        "kernels": {"k_brotli_dec": {"ms": round(ms["k_brotli_dec"], 4)}},
        "device_encoder": own,
        "decode_errors": bad_all, "roundtrip_verified": ok_all,
        "gen_s": round(gen_s, 2), "device": eng.name,
    }
    if cpu and not args.no_cpu and world == 1:
        res["cpu_baseline"] = cpu_baseline("brotli", chunk, args.cpu_mib)
    return res


def bench_snappy(ctx, gib_args=None, steps=None, warmup=None, main=True):
    """snappy-mt (not a BASELINE config): device-resident round trip at the reference's default
    64 KiB chunk, one GPU.  compress = zmt_snappy_enc_kernel + compact, decompress = zmt_snappy_dec_kernel
    (one kernel each, so the HIP-event times of the legs are the kernels'); verified byte for byte."""
    import copy
    args, eng = copy.copy(ctx.args), ctx.eng
    if ctx.world != 1:
        raise SystemExit("--codec snappy runs on one GPU")
    if gib_args is not None:
        args.gib = gib_args
    args.steps = steps or args.steps
    args.warmup = args.warmup if warmup is None else warmup
    chunk = (args.chunk if main else 0) or 65536
    n = int(args.gib * (1 << 30)) // chunk * chunk
    nrec = n // chunk
    stride = eng.snappy_slot_stride(chunk)
    d_in, gen_s = generate(ctx, n, 0)
    d_slots, d_rl, d_ro = eng.alloc(nrec * stride), eng.alloc(nrec * 4), eng.alloc((nrec + 1) * 8)
    d_stream = eng.alloc(nrec * stride + 512)
    d_ol, d_st, d_out = eng.alloc(nrec * 4), eng.alloc(nrec * 4), eng.alloc(n + 64)
    d_oo = eng.upload(np.arange(nrec + 1, dtype=np.uint64) * np.uint64(chunk))
    d_oc = eng.upload(np.full(nrec, chunk, np.uint32))
    bufs = [d_in, d_slots, d_rl, d_ro, d_stream, d_ol, d_st, d_out, d_oo, d_oc]

    def compress():
        eng.timer_start(1)
        eng.snappy_compress(d_in, n, chunk, d_slots, stride, d_rl)
        eng.timer_stop(1)
        eng.timer_start(2)
        eng.lz4_compact(d_slots, stride, d_rl, nrec, d_stream, d_ro)
        eng.timer_stop(2)

    compress()
    eng.sync(0)
    # payload table (the 16-byte headers are the host engine's to parse): offsets + 16, lengths - 16
    d_po = eng.upload(eng.download(d_ro, nrec * 8, np.uint64) + np.uint64(16))
    d_pl = eng.upload((eng.download(d_rl, nrec * 4, np.uint32) - 16).astype(np.uint32))
    bufs += [d_po, d_pl]

    def decompress():
        eng.timer_start(3)
        eng.snappy_decompress(d_stream, d_po, d_pl, nrec, d_out, d_oo, d_oc, d_ol, d_st)
        eng.timer_stop(3)

    for _ in range(args.warmup):
        compress()
        decompress()
    ctx.barrier()
    acc = {"compress": 0.0, "compact": 0.0, "decompress": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        compress()
        decompress()
        eng.sync(0)
        for k, slot in (("compress", 1), ("compact", 2), ("decompress", 3)):
            acc[k] += eng.timer_ms(slot)
    wall = time.perf_counter() - t0
    ms = {k: v / args.steps for k, v in acc.items()}
    total_c = int(eng.download(d_ro, 8, np.uint64, offset=nrec * 8)[0])
    bad = int((eng.download(d_st, nrec * 4, np.uint32) != 0).sum())
    ok = bool(eng.equal(d_in, d_out, n)) if args.verify else None
    for b in bufs:
        b.free()
    U, Cb = float(n), float(total_c)
    alg = U + Cb
    step_s = wall / args.steps

    traffic = traffic_table(args.gib, chunk, 65536)

    def roof(kernel, t_ms):
        a = alg / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
        tr = traffic.get(kernel)
        return {"kernel": kernel, "bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(a * 1e9 / HBM_PEAK, 5), "alg_bytes_per_launch": alg, "avg_launch_ms": round(t_ms, 4),
                "traffic": tr,
                "traffic_source": (TRAFFIC_FILE + " (separate rocprofv3 --pmc passes, not this run)") if tr else None}
    return {
        "metric": f"MB/s compress+decompress, {U / (1 << 30):g} GiB synthetic, snappy-mt; % HBM roofline",
        "value": round(U / 1e6 / step_s, 1), "unit": "MB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"snappy-mt, {U / (1 << 30):g} GiB enwik-style synthetic, {chunk // 1024} KiB chunks, "
                               "device-resident", "chunk": chunk, "records_per_gpu": nrec, "ratio": round(U / Cb, 4),
                   "dec_variant": args.snappy_dec,
                   "parity": "decompress-identical (no reference build: its snappy library is outside the tree)"},
        "compress_MBps": round(U / 1e6 / ((ms["compress"] + ms["compact"]) * 1e-3), 1),
        "decompress_MBps": round(U / 1e6 / (ms["decompress"] * 1e-3), 1),
        "roofline": roof("zmt_snappy_enc_kernel", ms["compress"]),
        "roofline_decompress": roof("zmt_snappy_dec2_kernel" if args.snappy_dec == 1 else "zmt_snappy_dec_kernel",
                                    ms["decompress"]),
        "kernels": {k: {"ms": round(v, 4)} for k, v in ms.items()},
        "decode_errors": bad, "roundtrip_verified": ok, "gen_s": round(gen_s, 2), "device": eng.name,
    }


def bench_api(mib):
    """PCIe-inclusive rates of the drop-in APIs (LZ4MT_* / ZSTDCB_* / BROTLIMT_* / SNAPPYMT_* compressCCtx and
    decompressDCtx with memcpy callbacks, round trip checked) -- zstdmt_amd/bin/api_bench, the same
    measurement oracle/cpu_bench makes for the reference libraries."""
    exe = os.path.join(ROOT, "zstdmt_amd", "bin", "api_bench")
    lib = os.path.join(ROOT, "zstdmt_amd", "lib", "libzstdmt_amd.so")
    out = {}
    for key, codec, chunk, level, size in (("lz4", "lz4", 131072, 1, mib << 20), ("zstd", "zstd", 1 << 20, 1, mib << 20),
                                           ("brotli", "brotli", 1 << 20, 1, mib << 20),
                                           # the CLI's default level: LZ4HC hash chain on the device
                                           ("lz4 level 3 (lz4-mt CLI default, LZ4HC)", "lz4", 131072, 3, min(mib, 2048) << 20),
                                           # snappy-mt at its default 64 KiB chunk (the level is unused)
                                           ("snappy", "snappy", 0, 0, min(mib, 2048) << 20)):
        try:
            t0 = time.time()
            # (the level-1 legs also time their callbacks alone: what bounds the leg is in the line next to it)
            env = dict(os.environ, ZMT_API_BOUND="1") if level == 1 and codec != "snappy" else None
            txt = subprocess.check_output([exe, codec, str(size), str(chunk), lib, str(level)], timeout=300,
                                          stderr=subprocess.DEVNULL, env=env)
            r = json.loads(txt.decode().strip().splitlines()[-1])
            r["seconds"] = round(time.time() - t0, 1)
            r["roundtrip_verified"] = True       # api_bench exits non-zero on a mismatch
            out[key] = r
        except Exception as e:  # report, never hide
            out[key] = {"error": repr(e)}
    return out


def cpu_reference_level(level, chunk, cpu_mib):
    """LZ4MT_* of the reference build at `level` on the host cores (bounded sample)"""
    exe = os.path.join(ROOT, "oracle", "cpu_bench")
    ref = os.path.join(ROOT, "oracle", "_ref", "liblz4mt_ref.so")
    if not (os.path.exists(exe) and os.path.exists(ref)):
        return None
    threads = min(os.cpu_count() or 1, 128)
    try:
        r = json.loads(subprocess.check_output([exe, "reference", ref, str(cpu_mib << 20), str(chunk), str(threads),
                                                str(SEED), str(level)], timeout=600))
        return {"compress_MBps": r["compress_MBps"], "decompress_MBps": r["decompress_MBps"], "threads": threads,
                "kind": "reference", "sample_MiB": cpu_mib, "ratio": round(r["bytes"] / r["compressed"], 4)}
    except Exception as e:  # report, never hide
        return {"error": repr(e)}


def rccl_gather(eng, dist, d_buf, sizes, rank, world):
    """gatherv of the per-rank segments to rank 0 over RCCL (zstdmt_amd.shard), timed."""
    import torch
    from zstdmt_amd.shard import gather_segments
    mine = int(sizes[rank])
    # RCCL wants torch tensors: stage the segment into a torch-owned buffer (D2D copy, untimed)
    seg = torch.empty(mine, dtype=torch.uint8, device="cuda")
    eng._ck(eng.L.gpumt_memcpy_d2d(eng.h, seg.data_ptr(), d_buf.ptr, mine, 0), "d2d")
    eng.sync(0)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gather_segments(seg, sizes, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    return round((time.perf_counter() - t0) * 1e3, 3)


def d2h_gather(eng, dist, d_buf, sizes, rank, world):
    """the no-collective reassembly: ONE pinned host buffer mapped by every rank (zstdmt_amd.shard.SharedHostStream),
    each GPU copies its segment to its own offset over its own PCIe link; timed like rccl_gather"""
    import torch
    from zstdmt_amd.shard import SharedHostStream
    total, off, mine = int(sum(sizes)), int(sum(sizes[:rank])), int(sizes[rank])
    name = "zstdmt_amd_bench_%s" % os.environ.get("MASTER_PORT", "0")
    if rank == 0:
        shm = SharedHostStream(name, total, owner=True)
    dist.barrier()
    if rank != 0:
        shm = SharedHostStream(name, total, owner=False)
    base = shm.address()
    eng._ck(eng.L.gpumt_host_register(eng.h, base, shm.total), "host_register")
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mine:
        eng._ck(eng.L.gpumt_memcpy_d2h(eng.h, base + off, d_buf.ptr, mine, 0), "d2h")
    eng.sync(0)
    dist.barrier()
    ms = round((time.perf_counter() - t0) * 1e3, 3)
    eng._ck(eng.L.gpumt_host_unregister(eng.h, base), "host_unregister")
    dist.barrier()
    shm.close()
    return ms


def dry_run(args, rank, local, world, dist):
    """ranks report where they would run; no GPU is touched (CPU test of the launch path)"""
    from zstdmt_amd.shard import shard_range
    chunk = args.chunk or ((1 << 20) if args.codec != "lz4" else 131072)
    nch = int(args.gib * (1 << 30)) // chunk
    lo, hi = shard_range(nch, rank, world) if args.scaling == "strong" else (0, nch)
    me = {"rank": rank, "local_rank": local, "device": local, "chunks": [lo, hi], "pid": os.getpid()}
    ranks = [None] * world
    if dist is not None:
        dist.all_gather_object(ranks, me)
    else:
        ranks = [me]
    if rank == 0:
        # the keys a real multi-GPU line carries beside the contract's: reassembly mode + time, per-rank kernel times
        print(json.dumps({"dry_run": True, "n_gpus": world, "scaling": args.scaling, "codec": args.codec,
                          "mode": args.mode, "gather": args.gather, "gather_ms": None,
                          "per_rank_ms": [{"rank": r["rank"], "k_enc": None, "k_dec": None, "compress_leg": None,
                                           "decompress_leg": None} for r in sorted(ranks, key=lambda x: x["rank"])],
                          "ranks": ranks}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn(args))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "RANK" in os.environ:
        # one process per GPU, RCCL ("nccl") process group.  torch must be imported BEFORE the native
        # library so both share one HIP runtime.
        import torch
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo")
            return dry_run(args, rank, local, world, dist)
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif args.dry_run:
        return dry_run(args, rank, local, world, None)

    os.environ["GPUMT_DEVICE"] = str(local)       # the drop-in APIs of this process use this rank's GPU
    import zstdmt_amd as z
    eng = z.Engine(local)
    eng.set_variant("lz4_dec", args.dec_variant)
    eng.set_variant("lz4_ring", args.lz4_ring)
    eng.set_variant("snappy_dec", args.snappy_dec)
    eng.set_variant("profile", 1)
    ctx = Ctx(args, eng, rank, world, dist)
    eng.set_variant("zstd_seq", args.zstd_seq)
    if args.zref_only:
        print(json.dumps(bench_zstd_ref(ctx, args.gib, args.steps, args.warmup)))
        return

    if args.codec == "brotli":
        res = bench_brotli(ctx)
    elif args.codec == "snappy":
        res = bench_snappy(ctx)
    else:
        res = bench_lz4_zstd(ctx, args.codec)

    default_run = (world == 1 and args.codec == "lz4" and args.mode == "roundtrip" and not args.only)
    if rank == 0 and default_run:
        # the other BASELINE configs, short, in the same line
        cfgs = {}

        def leg(name, fn):
            try:
                cfgs[name] = fn()
            except Exception as e:  # report, never hide
                cfgs[name] = {"error": repr(e)}
        leg(K_DEC, lambda: bench_lz4_zstd(ctx, "lz4", gib_args=args.extra_gib, steps=5, warmup=1, cpu=False, main=False,
                                          dec_only=True))
        leg(K_ZSTD, lambda: bench_lz4_zstd(ctx, "zstd", gib_args=args.extra_gib, steps=2, warmup=1, main=False))
        leg(K_ZREF, lambda: bench_zstd_ref(ctx, args.extra_gib, 2, 1))
        leg(K_BROTLI, lambda: bench_brotli(ctx, gib_args=args.extra_gib, steps=2, warmup=1, main=False))
        leg(K_SNAPPY, lambda: bench_snappy(ctx, gib_args=args.extra_gib, steps=2, warmup=1, main=False))
        eng.close()
        api = bench_api(int(args.api_gib * 1024))
        hc = "lz4 level 3 (lz4-mt CLI default, LZ4HC)"
        if isinstance(api.get(hc), dict) and not args.no_cpu:
            api[hc]["cpu_reference"] = cpu_reference_level(3, 131072, 1024)
        for codec, leg_ in (("lz4", res), ("zstd", cfgs[K_ZSTD]), ("brotli", cfgs[K_BROTLI])):
            cb = (leg_ or {}).get("cpu_baseline") or {}
            if isinstance(api.get(codec), dict) and cb.get("compress_MBps"):
                api[codec]["cpu_reference"] = {"compress_MBps": cb["compress_MBps"],
                                               "decompress_MBps": cb["decompress_MBps"],
                                               "threads": cb.get("cores"), "kind": cb.get("kind")}
        # the decode-only leg's CPU side is the decompress half of the main leg's reference run
        cb = res.get("cpu_baseline") or {}
        if isinstance(cfgs.get(K_DEC), dict) and cb.get("decompress_MBps"):
            cfgs[K_DEC]["cpu_baseline"] = {"value": cb["decompress_MBps"], "unit": "MB/s", "cores": cb.get("cores"),
                                           "kind": cb.get("kind"), "decompress_MBps": cb["decompress_MBps"],
                                           "sample": cb.get("sample")}
        cfgs[K_API] = api
        res["configs"] = cfgs
    if rank == 0:
        if world > 1:
            import torch
            res["world_size"] = dist.get_world_size()     # as RCCL formed it
            try:
                res["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # report, never hide
                res["rccl_version"] = repr(e)
        # everything measured goes to bench_detail.json and to an earlier line; the LAST line is the contract's
        # line, short enough (< 7 KB) for a reader that keeps only the tail of stdout
        try:
            with open(os.path.join(ROOT, "bench_detail.json"), "w") as f:
                json.dump(res, f, indent=1)
        except OSError:
            pass
        print("DETAIL " + json.dumps(res), flush=True)
        print(json.dumps(compact_line(res)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


K_DEC = "lz4-mt decompress-only (configs[2])"
K_ZSTD = "zstd-mt level 1 (configs[3])"
K_ZREF = "zstd-mt decompress of reference-written level-1 streams"
K_BROTLI = "brotli-mt decompress (configs[4])"
K_SNAPPY = "snappy-mt, 64 KiB chunks (SURVEY 8f-4, not a BASELINE config)"
K_API = "drop-in API, PCIe-inclusive (LZ4MT_*/ZSTDCB_*/BROTLIMT_* with memcpy callbacks)"


def _roof_short(r, nested=False):
    if not isinstance(r, dict):
        return r
    keep = ("kernel", "achieved", "frac", "avg_launch_ms", "traffic") if nested else \
        ("kernel", "bound", "achieved", "peak", "unit", "frac", "alg_bytes_per_launch", "avg_launch_ms", "traffic")
    o = {k: r[k] for k in keep if k in r}
    if nested and isinstance(o.get("kernel"), str):
        o["kernel"] = o["kernel"][:96]
    return o


def _cpu_short(c, nested=False):
    if not isinstance(c, dict):
        return c
    keep = ("value", "unit", "cores", "kind", "compress_MBps", "decompress_MBps", "host_cpus", "error")
    o = {k: c[k] for k in keep if k in c}
    if c.get("sample") and not nested:
        o["sample"] = c["sample"][:120]
    return o


def _leg_short(r):
    """one extra leg, as the final line carries it"""
    if not isinstance(r, dict) or "error" in r:
        return r
    o = {k: r[k] for k in ("value", "unit", "steps", "ms_per_step", "compress_MBps", "decompress_MBps", "decode_errors",
                           "roundtrip_verified") if k in r}
    o["workload"] = (r.get("config") or {}).get("workload", "")[:110]
    o["ratio"] = (r.get("config") or {}).get("ratio")
    o["roofline"] = _roof_short(r.get("roofline"), True)
    if r.get("roofline_decompress") and r["roofline_decompress"] != r.get("roofline"):
        o["roofline_decompress"] = _roof_short(r["roofline_decompress"], True)
    if isinstance(r.get("roofline_decompress_path"), dict):   # decode + the checksum verify LZ4F_decompress includes
        o["roofline_decompress_path"] = {k: r["roofline_decompress_path"][k] for k in ("achieved", "frac")}
    if r.get("cpu_baseline"):
        o["cpu_baseline"] = _cpu_short(r["cpu_baseline"], True)
    if isinstance(r.get("device_encoder"), dict) and "compress_MBps" in r["device_encoder"]:
        o["device_encoder"] = {k: r["device_encoder"][k] for k in ("compress_MBps", "ratio", "decompress_MBps",
                                                                   "roundtrip_verified")}
    return o


def compact_line(res):
    """The contract's JSON line: every contract key, the rooflines of both directions, the CPU baseline and one short
    entry per extra leg; per-kernel tables, per-thread CPU runs and the API legs' verbose fields stay in DETAIL /
    bench_detail.json."""
    o = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                             "scaling", "vs_baseline", "dtype", "data", "config") if k in res}
    o["roofline"] = _roof_short(res.get("roofline"))
    for k in ("roofline_decompress", "roofline_compress"):
        if res.get(k) and res[k] != res.get("roofline"):
            o[k] = _roof_short(res[k])
    if isinstance(res.get("roofline_decompress_path"), dict):
        o["roofline_decompress_path"] = {k: res["roofline_decompress_path"][k] for k in ("what", "achieved", "unit", "frac")}
    if res.get("cpu_baseline"):
        o["cpu_baseline"] = _cpu_short(res["cpu_baseline"])
    for k in ("compress_MBps", "decompress_MBps", "decode_errors", "roundtrip_verified", "device", "gather", "gather_ms",
              "value_with_gather", "world_size", "rccl_version", "device_encoder"):
        if k in res:
            o[k] = res[k]
    if res.get("kernels"):
        o["kernels_ms"] = {k: v["ms"] for k, v in res["kernels"].items()}
    if res.get("per_rank_ms") and res.get("n_gpus", 1) > 1:
        o["per_rank_ms"] = res["per_rank_ms"]
    if "configs" in res:
        c = {}
        for name, r in res["configs"].items():
            if name == K_API and isinstance(r, dict):
                c[name] = {k: ({kk: ({"fn_read": v[kk]["fn_read"], "fn_write": v[kk]["fn_write"]}
                                     if kk == "callbacks_alone_MBps" else v[kk])
                                for kk in ("compress_MBps", "decompress_MBps", "callbacks_alone_MBps", "error") if kk in v}
                               if isinstance(v, dict) else v) for k, v in r.items()}
            else:
                c[name] = _leg_short(r)
        o["configs"] = c
    o["detail"] = "bench_detail.json + the DETAIL line above"
    return o


if __name__ == "__main__":
    main()
