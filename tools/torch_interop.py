import sys, os, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    print("torch", torch.__version__, torch.cuda.is_available(), flush=True)
    if len(sys.argv) > 2:
        x = torch.zeros(4, device="cuda"); torch.cuda.synchronize(); print("torch cuda ok", flush=True)
import zstdmt_amd as z
print("maps:", [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][::8], flush=True)
e = z.Engine(0)
print("engine", e.name, flush=True)
import helpers as H
from cases import text
d = text(300000)
s, ro, rl = e.compress_bytes(d, 131072)
print("compress ok", s == H.oracle_compress(d, 131072), flush=True)
if mode == "lib_first":
    import torch
    x = torch.zeros(4, device="cuda"); torch.cuda.synchronize(); print("torch after ok", flush=True)
out, st = e.decompress_bytes(s, ro, rl)
print("decompress ok", out == d, flush=True)
