#!/bin/bash
# round-3 GPU session 9: LZ4 encoder -- parity, timing, phase profile, instruction counts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s9; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q -m gpu > $O/pytest_lz4.txt 2>&1
tail -3 $O/pytest_lz4.txt
timeout 300 python bench.py --only --steps 3 --warmup 1 --no-cpu > $O/enc.json 2> $O/enc.err
timeout 300 python tools/enc_prof.py 2 > $O/enc_prof.txt 2>&1
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"
timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/sq_A -- \
   python bench.py --only --steps 1 --warmup 0 --no-cpu --no-verify > /dev/null 2> $O/sq_A.err
python - <<'PY'
import json,glob,csv,os
from collections import defaultdict
d=json.loads(open("gpurun_out/s9/enc.json").read().strip().splitlines()[-1])
print(d["value"], d["kernels"], d["roofline"]["frac"], d["decode_errors"], d["roundtrip_verified"])
print(open("gpurun_out/s9/enc_prof.txt").read())
fs=glob.glob("gpurun_out/s9/sq_A/**/*_counter_collection.csv",recursive=True)
if fs:
    f=max(fs,key=os.path.getsize)
    tot=defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        tot[row["Kernel_Name"]][row["Counter_Name"]]+=float(row["Counter_Value"])
    for k in tot:
        if "enc3_p17" in k:
            print(k, {c: "%.4g"%v for c,v in sorted(tot[k].items())}, "total %.4g"%sum(tot[k].values()))
PY
