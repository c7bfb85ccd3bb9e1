"""Ad-hoc GPU diagnostics (not a test): prints status words and first mismatches."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
import numpy as np
import helpers as H, emu_driver as E
from cases import text, rnd
import zstdmt_amd as z

eng = z.Engine(0)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1"])]

def check(name, data, chunk):
    s = H.oracle_compress(data, chunk)
    ro, rl = E.walk_records(s)
    for v in variants:
        eng.set_variant("lz4_dec", v)
        for rep in range(2):
            out, st = eng.decompress_bytes(s, ro, rl)
            a = np.frombuffer(out, np.uint8); b = np.frombuffer(data, np.uint8)
            ok = len(a) == len(b) and bool((a == b).all())
            msg = f"{name:28s} v{v} rep{rep} nrec={len(rl)} status={np.unique(st).tolist()} data_ok={ok}"
            if not ok and len(a) == len(b):
                bad = np.nonzero(a != b)[0]
                msg += f" nbad={len(bad)} first={bad[:8].tolist()} chunkpos={[int(x)%chunk for x in bad[:4]]}"
                p = int(bad[0])
                msg += f" got={a[p:p+8].tolist()} want={b[p:p+8].tolist()}"
            print(msg, flush=True)

check("smoke", text(3 * 131072 + 1000) + bytes(5000), 131072)
check("text_1000_zeros_5000", text(1000) + bytes(5000), 131072)
check("zeros_5000", bytes(5000), 131072)
check("zeros_6000", bytes(6000), 131072)
check("text_1000", text(1000), 131072)
check("rnd_8m_chunk4m", rnd(8 << 20, 7), 4 << 20)
check("rnd_4m_chunk4m", rnd(4 << 20, 7), 4 << 20)
check("rnd_1m_chunk1m", rnd(1 << 20, 7), 1 << 20)
check("rnd_64m_chunk4m", rnd(64 << 20, 7), 4 << 20)
check("text_64m_chunk128k", text(64 << 20), 131072)
check("text_8m_chunk4m", text(8 << 20), 4 << 20)

print("---- compress then decompress on the same engine ----")
def check2(name, data, chunk):
    stream, ro, rl = eng.compress_bytes(data, chunk)
    want = H.oracle_compress(data, chunk)
    wro, wrl = E.walk_records(want)
    print(f"{name:24s} stream_ok={stream == want} ro_ok={ro[:len(rl)].tolist() == wro.tolist()} rl_ok={rl.tolist() == wrl.tolist()} ro_last={int(ro[-1])} len={len(stream)}", flush=True)
    for v in variants:
        eng.set_variant("lz4_dec", v)
        for rep in range(2):
            out, st = eng.decompress_bytes(stream, ro, rl)
            ok = out == data
            print(f"   v{v} rep{rep} status={st.tolist()[:20]} data_ok={ok} outlen={len(out)}", flush=True)

check2("smoke", text(3 * 131072 + 1000) + bytes(5000), 131072)
check2("rnd_64m_chunk4m", rnd(64 << 20, 7), 4 << 20)
check2("text_1m", text(1 << 20), 131072)
