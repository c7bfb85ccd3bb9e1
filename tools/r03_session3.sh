#!/bin/bash
# round-3 GPU session 3: where copy3 / parse3 spend their time (phase counters + SQ instruction counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s3; rm -rf $O; mkdir -p $O
K3PROF=1 timeout 300 python tools/dec_prof.py 2 0,194,210,226 > $O/dec_prof.txt 2>&1
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"
B="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
for pass in A B; do
  eval "ctr=\$$pass"
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/sq_$pass -- \
     python bench.py --only --mode decompress --dec-variant 2 --lz4-ring 13 --steps 1 --warmup 0 --no-cpu --no-verify > /dev/null 2> $O/sq_$pass.err
done
cat $O/dec_prof.txt
python - <<'PY'
import csv,glob
from collections import defaultdict
for p in ("A","B"):
    fs=glob.glob(f"gpurun_out/s3/sq_{p}/**/*_counter_collection.csv",recursive=True)
    if not fs: print("no csv",p); continue
    f=max(fs,key=lambda x:__import__("os").path.getsize(x))
    tot=defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        tot[row["Kernel_Name"]][row["Counter_Name"]]+=float(row["Counter_Value"])
    for k,v in tot.items():
        if "copy3" in k or "parse3" in k:
            print(p,k[:30],{a:f"{b:.4g}" for a,b in v.items()})
PY
