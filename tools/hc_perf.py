"""LZ4HC (levels 3..8) device encoder: throughput per level on the bench text (developer tool)."""
import sys, os, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zstdmt_amd as z
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
levels = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "3", "5", "8"])]
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
eng = z.Engine(0)
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(gib * (1 << 30)) // chunk * chunk
nrec = n // chunk; stride = eng.slot_stride(chunk)
hb = np.empty(n, np.uint8); T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 32)
d_in = eng.upload(hb)
d_slots = eng.alloc(nrec * stride); d_rl = eng.alloc(nrec * 4)
for lv in levels:
  for hw in ([int(x) for x in os.environ.get('HCW','0').split(',')] if lv >= 3 else [0]):
    eng.set_variant('hc_waves', hw)
    for rep in range(2):
        eng.sync(); t = time.perf_counter()
        eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl, level=lv)
        eng.sync(); dt = time.perf_counter() - t
    rl = eng.download(d_rl, nrec * 4, np.uint32)
    print(f"level {lv} waves {hw}: {dt*1e3:.1f} ms  {n/1e9/dt:.2f} GB/s  ratio {n/int(rl.astype(np.uint64).sum()):.3f}", flush=True)
