"""Per-phase cycle profile of the batch decoder (developer tool)."""
import sys, os, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zstdmt_amd as z
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
# variants: 0 | ring << 4 = frames + parse4 + copy3 (ring 12 if omitted), 1 = frame-serial
variants = [int(v, 0) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1"])]
eng = z.Engine(0); L, h = eng.L, eng.h
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
chunk = 131072
n = int(gib * (1 << 30)) // chunk * chunk
nrec = n // chunk; stride = eng.slot_stride(chunk)
hb = np.empty(n, np.uint8); T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 32)
d_in = eng.upload(hb)
d_slots = eng.alloc(nrec * stride); d_rl = eng.alloc(nrec * 4); d_ro = eng.alloc((nrec + 1) * 8)
d_stream = eng.alloc(nrec * stride); d_ol = eng.alloc(nrec * 4); d_oo = eng.alloc((nrec + 1) * 8)
d_st = eng.alloc(nrec * 4); d_out = eng.alloc(n + 64)
eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl)
eng.lz4_compact(d_slots, stride, d_rl, nrec, d_stream, d_ro)
eng.lz4_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo)
eng.sync()
names = ["stage", "spec", "walk", "decode+scan+classify", "literals", "fence+far", "rounds", "flush", "slowpath", "batches", "seqs", "total"]
for v in variants:
    eng.set_variant("lz4_dec", v & 15)
    eng.set_variant("lz4_ring", ((v >> 4) & 15) or 12)
    eng.set_variant("profile", 1)
    for rep in range(2):
        cnt = (C.c_ulonglong * 16)()
        L.gpumt_debug_counters(h, cnt, 16)
        eng.lz4_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
        eng.sync()
        ms = eng.timer_ms(11)
    st = eng.download(d_st, nrec * 4, np.uint32)
    ok = bool((eng.download(d_out, n) == hb).all())
    print(f"variant {v}: kernel {ms:.3f} ms  ({n/1e6/ms:.1f} GB/s out)  errors={int((st!=0).sum())} data_ok={ok}")
    if (v & 15) == 0 and os.environ.get("K3PROF"):
        eng.set_variant("profile", 8)
        cnt = (C.c_ulonglong * 16)(); L.gpumt_debug_counters(h, cnt, 16)
        eng.lz4_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st); eng.sync()
        L.gpumt_debug_counters(h, cnt, 16); c = list(cnt); nbat = max(c[12], 1)
        nm = ["wait(stage,tokens,stores)", "prefetch+loop", "fields", "scan+check+classify", "far-issue", "literals", "far-land", "match-r1", "rounds", "flush", "singles"]
        print(f"   copy3 prof ring {(v >> 4) & 15} (cycles per batch): " + ", ".join(f"{nm[i]}={c[i]/nbat:.0f}" for i in range(11)) + f" total={c[11]/nbat:.0f} batches={c[12]} rounds/batch={c[13]/nbat:.2f}")
        eng.set_variant("profile", 1)
    if (v & 15) == 0:
        print("   split: frames %.3f ms, parse %.3f ms, copy %.3f ms, xxh %.3f ms" % (eng.timer_ms(13), eng.timer_ms(14), eng.timer_ms(15), eng.timer_ms(12)))
