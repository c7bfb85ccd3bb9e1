"""Per-phase cycle profile of the v3 encoder (developer tool). PADS=0,8192 sweeps occupancy."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zstdmt_amd as z
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
eng = z.Engine(0); L, h = eng.L, eng.h
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
chunk = 131072
n = int(gib * (1 << 30)) // chunk * chunk
nrec = n // chunk; stride = eng.slot_stride(chunk)
hb = np.empty(n, np.uint8); T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 32)
d_in = eng.upload(hb); d_slots = eng.alloc(nrec * stride); d_rl = eng.alloc(nrec * 4)
names = ["search: input+hash+table+dup", "search: gather+vote+insert", "catch-up", "literal emit",
         "match count", "re-match test", "#batches", "loop/other"]
for pad in [int(x) for x in os.environ.get("PADS", "0").split(",")]:
    eng.set_variant("k2x", pad)
    print("== dynamic LDS pad", pad)
    for prof in (1, 4):
        eng.set_variant("profile", prof)
        cnt = (C.c_ulonglong * 16)(); L.gpumt_debug_counters(h, cnt, 16)
        for rep in range(2):
            eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl); eng.sync()
        print(f"profile={prof}: enc kernel {eng.timer_ms(9):.2f} ms")
        if prof == 4:
            L.gpumt_debug_counters(h, cnt, 16); c = list(cnt)
            w = max(c[9], 1)
            print(f"  waves={w} batches/wave={c[6]/w:.0f}  total cycles/wave={c[8]/w:.0f}")
            for i in (0, 1, 2, 3, 4, 5, 7):
                print(f"  {names[i]:32s} {c[i]/w/1e6:8.2f} Mcycles/wave  {100*c[i]/c[8]:5.1f}%")
