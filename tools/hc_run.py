import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, zstdmt_amd as z
eng = z.Engine(0)
T = C.CDLL("zstdmt_amd/lib/libzmt_tools.so"); T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = 2 << 30; chunk = 131072; nrec = n // chunk; stride = eng.slot_stride(chunk)
hb = np.empty(n, np.uint8); T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 32)
d_in = eng.upload(hb); d_slots = eng.alloc(nrec * stride); d_rl = eng.alloc(nrec * 4)
eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl, level=3); eng.sync()
