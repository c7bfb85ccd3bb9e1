#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache counters of the dominant kernels (own rocprofv3 --pmc pass).
#   gpurun --timeout 600 -- 'bash tools/profile_icache.sh [codec ...]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
for codec in ${@:-lz4 zstd brotli}; do
  rm -rf $O/ic_${codec}
  X=""; [ $codec = brotli ] && X="--no-encoder"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ic_${codec} -- \
      python bench.py --only --codec $codec --steps 1 --warmup 0 --no-cpu $X > /dev/null 2> $O/ic_${codec}.err
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/ic_*/")):
    files = glob.glob(d + "**/*_counter_collection.csv", recursive=True)
    if not files:
        print(d, "no counter file"); continue
    f = max(files, key=lambda p: __import__("os").path.getsize(p))
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", d)
    for k, c in tot.items():
        if c.get("SQ_WAVE_CYCLES", 0) < 1e9:
            continue
        req = max(c.get("SQC_ICACHE_REQ", 0), 1)
        print(f"{k[:34]:34s} icache req {req:.3g} hit {c.get('SQC_ICACHE_HITS',0)/req:.3f} miss {c.get('SQC_ICACHE_MISSES',0)/req:.4f} "
              f"ifetch {c.get('SQ_IFETCH',0):.3g} wait_inst/wave_cycles {c.get('SQ_WAIT_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f}")
PY
