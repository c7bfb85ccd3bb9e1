import sys, random
sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import helpers as H, emu_driver as E, numpy as np
from test_oracle_vs_ref import _mix
from cases import text, rnd
n0=int(sys.argv[1]) if len(sys.argv)>1 else 0
n1=int(sys.argv[2]) if len(sys.argv)>2 else 40
for seed in range(n0,n1):
    rng=random.Random(seed*104729+7)
    n=rng.choice([rng.randrange(1,200000), rng.randrange(1,20000), 131072, 65536+rng.randrange(0,100), rng.randrange(65000,66000)])
    chunk=rng.choice([65536,131072,131072,100000,262144])
    kind=rng.randrange(5)
    if kind==0: data=_mix(rng,n)
    elif kind==1: data=text(n,seed=rng.randrange(1<<30))
    elif kind==2:
        k=rng.choice([2,3,4,7]); data=bytes((b%k)+65 for b in rnd(n,rng.randrange(1<<30)))
    elif kind==3: data=rnd(n,rng.randrange(1<<30))
    else:
        t=bytearray(text(n,seed=rng.randrange(1<<30)))
        for _ in range(rng.randrange(1,20)):
            if n<100: break
            a=rng.randrange(0,n-50); l=rng.randrange(1,min(5000,n-a)); b=rng.randrange(0,n-l)
            t[b:b+l]=t[a:a+l] if rng.random()<0.7 else bytes([rng.randrange(256)])*l
        data=bytes(t)
    want=H.oracle_compress(data,chunk)
    got,_,_=E.compress(data,chunk,0)
    ok = got==want
    print(seed,kind,n,chunk,"OK" if ok else "FAIL",len(got),len(want),flush=True)
    if not ok:
        a=np.frombuffer(got[:min(len(got),len(want))],np.uint8); b=np.frombuffer(want[:min(len(got),len(want))],np.uint8)
        bad=np.nonzero(a!=b)[0]; print("first diff",bad[:5]); break
