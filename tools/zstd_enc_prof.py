"""Per-phase cycle profile of the zstd block encoder (developer tool)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zstdmt_amd as z

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
chunk = 1 << 20
eng = z.Engine(0)
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(gib * (1 << 30)) // chunk * chunk
nrec = n // chunk
stride = eng.zstd_slot_stride(chunk)
hb = np.empty(n, np.uint8)
T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 32)
d_in = eng.upload(hb)
d_slots = eng.alloc(nrec * stride)
d_rl = eng.alloc(nrec * 4)
eng.set_variant("profile", 1)
for rep in range(2):
    eng.zstd_compress(d_in, n, chunk, d_slots, stride, d_rl)
    eng.sync()
print(f"encode {eng.timer_ms(9):.2f} ms ({n/1e6/eng.timer_ms(9):.1f} GB/s)")
eng.set_variant("profile", 6)
cnt = (C.c_ulonglong * 16)()
eng.L.gpumt_debug_counters(eng.h, cnt, 16)
eng.zstd_compress(d_in, n, chunk, d_slots, stride, d_rl)
eng.sync()
eng.L.gpumt_debug_counters(eng.h, cnt, 16)
c = list(cnt)
nblk = n // 131072
nm = ["parse", "fse", "gather", "huffman", "seq+hdr+rest", "extend", "lookup", "compare"]
print("  Mcycles per 128 KiB block: " + ", ".join(f"{nm[i]}={c[i]/nblk/1e6:.2f}" for i in range(8)) + f" total={c[8]/nblk/1e6:.2f} waves={c[9]}")
