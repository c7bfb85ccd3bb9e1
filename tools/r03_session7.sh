#!/bin/bash
# round-3 GPU session 7: cache policy of the far-match loads of copy3 (time + TCC request sizes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s7; rm -rf $O; mkdir -p $O
B="python bench.py --only --mode decompress --steps 4 --warmup 1 --no-cpu"
P="python bench.py --only --mode decompress --steps 1 --warmup 0 --no-cpu --no-verify"
for v in base far1 far2 far3 far4; do
  if [ $v = base ]; then unset ZMT_LIB; else export ZMT_LIB=$PWD/zstdmt_amd/lib/variants/$v.so; fi
  timeout 200 $B > $O/dec_$v.json 2> $O/dec_$v.err
  timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/tcc_$v -- $P > /dev/null 2> $O/tcc_$v.err
done
unset ZMT_LIB
python - <<'PY'
import json,glob,csv,os
from collections import defaultdict
for v in ("base","far1","far2","far3","far4"):
    try:
        d=json.loads(open(f"gpurun_out/s7/dec_{v}.json").read().strip().splitlines()[-1])
        line=f"{v}: copy {d['kernels']['k_dec_copy']['ms']} ms parse {d['kernels']['k_dec_parse']['ms']} total {d['kernels']['k_lz4_dec']['ms']} ok {d['roundtrip_verified']} err {d['decode_errors']}"
    except Exception as e:
        line=f"{v}: ERR {e}"
    fs=glob.glob(f"gpurun_out/s7/tcc_{v}/**/*_counter_collection.csv",recursive=True)
    if fs:
        f=max(fs,key=os.path.getsize); tot=defaultdict(float)
        for row in csv.DictReader(open(f)):
            if "copy3" in row["Kernel_Name"]: tot[row["Counter_Name"]]+=float(row["Counter_Value"])
        b=tot.get("TCC_EA0_RDREQ_32B_sum",0)*32+tot.get("TCC_EA0_RDREQ_64B_sum",0)*64+tot.get("TCC_EA0_RDREQ_128B_sum",0)*128
        line+=f" | RDREQ {tot.get('TCC_EA0_RDREQ_sum',0):.4g} 32B {tot.get('TCC_EA0_RDREQ_32B_sum',0):.4g} 64B {tot.get('TCC_EA0_RDREQ_64B_sum',0):.4g} 128B {tot.get('TCC_EA0_RDREQ_128B_sum',0):.4g} => {b/1e9:.1f} GB fetched"
    print(line)
PY
