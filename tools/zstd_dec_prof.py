"""Per-phase cycle profile of the zstd decoder on streams written by the device encoder (developer tool)."""
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
d_ro = eng.alloc((nrec + 1) * 8)
d_stream = eng.alloc(nrec * stride)
d_ol, d_oo, d_st = eng.alloc(nrec * 4), eng.alloc((nrec + 1) * 8), eng.alloc(nrec * 4)
d_out = eng.alloc(n + 64)
eng.zstd_compress(d_in, n, chunk, d_slots, stride, d_rl)
eng.lz4_compact(d_slots, stride, d_rl, nrec, d_stream, d_ro)
eng.sync()
eng.set_variant("profile", 1)
for rep in range(2):
    eng.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
    eng.zstd_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
    eng.sync()
print(f"decode {eng.timer_ms(11):.2f} ms ({n/1e6/eng.timer_ms(11):.1f} GB/s)")
eng.set_variant("profile", 5)
cnt = (C.c_ulonglong * 16)()
eng.L.gpumt_debug_counters(eng.h, cnt, 16)
eng.zstd_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo, d_st)
eng.zstd_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st)
eng.sync()
eng.L.gpumt_debug_counters(eng.h, cnt, 16)
c = list(cnt)
w = max(c[9], 1)
nm = ["hdr+huftab", "huffman", "seqhdr+tables", "stage+fse", "exec-lit", "exec-match", "other"]
print("  Mcycles per record: " + ", ".join(f"{nm[i]}={c[i]/w/1e6:.2f}" for i in range(7)) + f" total={c[8]/w/1e6:.2f}")
ok = bool((eng.download(d_out, n) == hb).all())
print("content ok:", ok)
