import sys, random
sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import helpers as H, emu_driver as E, numpy as np
from test_oracle_vs_ref import _mix
from cases import text, rnd
n0=int(sys.argv[1]) if len(sys.argv)>1 else 0
n1=int(sys.argv[2]) if len(sys.argv)>2 else 40
for seed in range(n0,n1):
    rng=random.Random(seed*7919+13)
    n=rng.choice([rng.randrange(1,300000), rng.randrange(1,30000), 131072, 65536+rng.randrange(0,100)])
    chunk=rng.choice([65536,131072,131072,100000,262144])
    kind=rng.randrange(4)
    if kind==0: data=_mix(rng,n)
    elif kind==1: data=text(n,seed=rng.randrange(1<<30))
    elif kind==2:
        # low-entropy soup: short offsets, overlaps
        k=rng.choice([2,3,4,7]); data=bytes((b%k)+65 for b in rnd(n,rng.randrange(1<<30)))
    else:
        # text with long runs and far repeats
        t=bytearray(text(n,seed=rng.randrange(1<<30)))
        for _ in range(rng.randrange(1,20)):
            if n<100: break
            a=rng.randrange(0,n-50); l=rng.randrange(1,min(5000,n-a)); b=rng.randrange(0,n-l)
            t[b:b+l]=t[a:a+l] if rng.random()<0.7 else bytes([rng.randrange(256)])*l
        data=bytes(t)
    s=H.oracle_compress(data,chunk)
    ok=True
    for v in (0, 13<<4, 14<<4):   # parse3 + copy3 at 4 / 8 / 16 KiB rings
        out,st=E.decompress(s,v)
        ok = ok and (not st.any()) and out==data
        if not ok: print("variant",v); break
    print(seed,kind,n,chunk,"OK" if ok else "FAIL",st.tolist()[:5],flush=True)
    if not ok:
        a=np.frombuffer(out,np.uint8); b=np.frombuffer(data,np.uint8)
        bad=np.nonzero(a!=b)[0]; print("nbad",len(bad),bad[:10]); break
