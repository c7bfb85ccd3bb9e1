#!/bin/bash
# round-3 GPU session 4: parse3 / copy3 v2 -- parity, timing by ring size, phase counters, instruction counts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q -m gpu > $O/pytest_lz4.txt 2>&1
tail -3 $O/pytest_lz4.txt
B="python bench.py --only --mode decompress --steps 5 --warmup 1 --no-cpu"
for r in 12 13 14; do
  timeout 300 $B --lz4-ring $r > $O/dec_v2_r$r.json 2> $O/dec_v2_r$r.err
done
K3PROF=1 timeout 300 python tools/dec_prof.py 2 0,208 > $O/dec_prof.txt 2>&1
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"
timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/sq_A -- \
   python bench.py --only --mode decompress --lz4-ring 12 --steps 1 --warmup 0 --no-cpu --no-verify > /dev/null 2> $O/sq_A.err
python - <<'PY'
import json,glob,csv,os
from collections import defaultdict
for f in sorted(glob.glob("gpurun_out/s4/dec_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["kernels"], d["roofline_decompress"]["frac"], d["decode_errors"], d["roundtrip_verified"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
print(open("gpurun_out/s4/dec_prof.txt").read())
fs=glob.glob("gpurun_out/s4/sq_A/**/*_counter_collection.csv",recursive=True)
if fs:
    f=max(fs,key=os.path.getsize)
    tot=defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        tot[row["Kernel_Name"]][row["Counter_Name"]]+=float(row["Counter_Value"])
    for k,v in tot.items():
        if "copy3" in k or "parse3" in k:
            print(k[:30],{a:f"{b:.4g}" for a,b in v.items()}, "total %.4g"%sum(b for a,b in v.items() if a!="SQ_INSTS_SMEM"))
PY
