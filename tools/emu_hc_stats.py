"""Developer tool: path statistics of the LZ4HC window scan (lz4_enc_hc.hip, levels 3..8) on the bench text, emulator build."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import helpers as H, emu_driver as E
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(mib * (1 << 20)) // 131072 * 131072
hb = np.empty(n, np.uint8); T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 8)
data = hb.tobytes()
lib = E.lib()
st = (C.c_ulonglong * 16).in_dll(lib, "zmt_hc_stat")
for i in range(16): st[i] = 0
got, _, _ = E.compress(data, 131072, level)
assert got == H.oracle_compress_level(data, 131072, level)
names = ["windows built", "first matches (sequences started)", "first searches left to the serial search", "wider searches from the window",
         "wider: no window over the position", "wider: more than 16 bytes to look back", "wider: lane left to the serial search"]
for i, nm in enumerate(names): print("%-45s %10d" % (nm, st[i]))
print("window scan: candidates visited %d, of them with the position's four bytes %d (%.1f %%)" % (st[8], st[9], 100.0 * st[9] / max(st[8], 1)))
print("bytes per window %.1f" % (n / max(st[0], 1)))
