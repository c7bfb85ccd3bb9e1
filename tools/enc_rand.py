"""Developer tool: LZ4 fast encoder on incompressible input (BASELINE configs[0] shape: PRNG bytes, 4 MiB chunks) and on text at
other chunk sizes.  usage: python tools/enc_rand.py [GiB]      GPUMT_LZ4_ENC=3 selects lz4_enc3.hip"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zstdmt_amd as z
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
eng = z.Engine(0); L, h = eng.L, eng.h
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(gib * (1 << 30))
rng = np.random.default_rng(7)
rnd = rng.integers(0, 256, n, dtype=np.uint8)
txt = np.empty(n, np.uint8); T.zmt_gen_text(txt.ctypes.data, n, 20260926, 0, 32)
eng.set_variant("profile", 1)
for name, hb in (("random", rnd), ("text", txt)):
    d_in = eng.upload(hb)
    for chunk in (65536, 131072, 1 << 20, 4 << 20):
        nrec = (n + chunk - 1) // chunk; stride = eng.slot_stride(chunk)
        d_slots = eng.alloc(nrec * stride); d_rl = eng.alloc(nrec * 4)
        for rep in range(2):
            eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl); eng.sync()
        rl = eng.download(d_rl, nrec * 4, np.uint32)
        print(f"{name:7s} chunk {chunk:8d}: enc kernel {eng.timer_ms(9):8.2f} ms per {gib:g} GiB  ratio {n / rl.astype(np.uint64).sum():.3f}")
        d_slots.free(); d_rl.free()
    d_in.free()
