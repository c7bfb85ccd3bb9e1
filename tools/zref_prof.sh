#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the leg "zstd-mt decompress of reference-written streams" by itself and the
# rocprofv3 per-kernel times of the same command.   bash tools/zref_prof.sh ["extra bench flags"]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py --zref-only --gib 8 --steps 3 --warmup 1 --no-cpu $1 > $O/zq.json 2> $O/zq.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/zq.json").read().strip().splitlines()[-1])
    print("zref", d.get("value"), d.get("ms_per_step"), d.get("kernels"), d.get("decode_errors"), d.get("roundtrip_verified"))
except Exception as e:
    print("ERR", e, open("gpurun_out/zq.err").read()[-800:])
P
rm -rf $O/zq_prof && rocprofv3 --kernel-trace --stats -d $O/zq_prof -o zq --output-format csv -- python bench.py --zref-only --gib 8 --steps 3 --warmup 1 --no-cpu $1 > /dev/null 2>&1
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/zq_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), "ms", r["Percentage"])
P
