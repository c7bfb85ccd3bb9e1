"""Developer tool: path statistics of the window encoder (lz4_enc5.hip) on the bench text, from the emulator build."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import helpers as H, emu_driver as E
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
T = C.CDLL(os.path.join(ROOT, "zstdmt_amd", "lib", "libzmt_tools.so"))
T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = int(mib * (1 << 20)) // 131072 * 131072
hb = np.empty(n, np.uint8); T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 8)
data = hb.tobytes()
lib = E.lib() if hasattr(E, "lib") else C.CDLL(os.path.join(ROOT, "tests", "emu", "libzmt_emu.so"))
st = (C.c_ulonglong * 16).in_dll(lib, "zmt_e5_stat")
for i in range(16): st[i] = 0
got, _, _ = E.compress(data, 131072, 0)
assert got == H.oracle_compress(data, 131072)
names = ["windows", "windows with twins", "searches", "searches through the twin code", "winners with a window candidate",
         "sequences from windows", "  of them by the general extension", "searches continued in probe batches", "searches that left their window", "searches through the chain walk", "sequences from lane-parallel runs", "serial: forward or window undecided (!qstat)", "serial: backward undecided", "serial: no wide window"]
for i, nm in enumerate(names): print("%-45s %10d" % (nm, st[i]))
print("sequences per window %.2f, bytes per window %.1f" % (st[5] / st[0], n / st[0]))
