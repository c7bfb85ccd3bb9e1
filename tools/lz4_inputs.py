"""Developer tool: the LZ4 fast encoder and the decode path on other inputs than the bench text -- zeros, random bytes, a mix with
long runs, sparse text -- at several chunk sizes (2 GiB each, device-resident; round trip verified).   python tools/lz4_inputs.py [gib]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zstdmt_amd as z
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
eng = z.Engine(0)
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(gib * (1 << 30)) >> 22 << 22
text = np.empty(n, np.uint8); T.zmt_gen_text(text.ctypes.data, n, 20260926, 0, 32)
rng = np.random.default_rng(7)
rand = rng.integers(0, 256, n, dtype=np.uint8)
zeros = np.zeros(n, np.uint8)
mix = text.copy()                       # text with a 4 KiB run of zeros and 4 KiB of random bytes in every 64 KiB
v = mix.reshape(-1, 65536); v[:, 8192:12288] = 0; v[:, 40000:44096] = rand[:v.shape[0] * 4096].reshape(-1, 4096)
sparse = zeros.copy()                   # 256 bytes of text at the head of every 4 KiB page
sparse.reshape(-1, 4096)[:, :256] = text[:n // 16].reshape(-1, 256)
eng.set_variant("profile", 1)
print("%-8s %8s %10s %10s %10s %8s" % ("input", "chunk", "enc ms", "enc GB/s", "dec GB/s", "ratio"))
for name, hb in (("text", text), ("random", rand), ("zeros", zeros), ("mix", mix), ("sparse", sparse)):
    d_in = eng.upload(hb)
    for chunk in (65536, 131072, 1 << 20, 4 << 20):
        nrec = n // chunk; stride = eng.slot_stride(chunk)
        d_slots = eng.alloc(nrec * stride); d_rl = eng.alloc(nrec * 4); d_ro = eng.alloc((nrec + 1) * 8)
        d_stream = eng.alloc(nrec * stride + 512)
        d_ol, d_oo, d_st = eng.alloc(nrec * 4), eng.alloc((nrec + 1) * 8), eng.alloc(nrec * 4)
        d_out = eng.alloc(n + 64)
        for rep in range(2):
            eng.lz4_compress(d_in, n, chunk, d_slots, stride, d_rl); eng.sync()
        enc = eng.timer_ms(9)
        eng.lz4_compact(d_slots, stride, d_rl, nrec, d_stream, d_ro); eng.sync()
        csize = int(eng.download(d_ro, (nrec + 1) * 8, np.uint64)[nrec])
        for rep in range(2):
            eng.lz4_probe(d_stream, d_ro, d_rl, nrec, d_ol, d_oo)
            eng.lz4_decompress(d_stream, nrec * stride, d_ro, d_rl, nrec, d_out, n, d_oo, d_ol, d_st); eng.sync()
        dec = eng.timer_ms(11) + eng.timer_ms(12)
        ok = bool((eng.download(d_out, n) == hb).all()) and not eng.download(d_st, nrec * 4, np.uint32).any()
        print("%-8s %8d %10.2f %10.1f %10.1f %8.3f %s" % (name, chunk, enc, n / 1e6 / enc, n / 1e6 / dec, n / csize, "" if ok else "ROUND TRIP FAILED"))
        for b in (d_slots, d_rl, d_ro, d_stream, d_ol, d_oo, d_st, d_out):
            b.free()
    d_in.free()
