#!/bin/bash
# Developer session (GPU box): zstd sequence pre-pass -- parity tests, the reference-stream leg with and without it,
# per-kernel times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_brotli.py tests/test_gpu_zstdmt_api.py -x -q -m gpu > $O/zs_pytest.txt 2>&1
tail -n 4 $O/zs_pytest.txt
python bench.py --zref-only --gib 8 --steps 3 --warmup 1 --no-cpu > $O/zs_zref_seq.json 2> $O/zs_zref_seq.err
python bench.py --zref-only --gib 8 --steps 3 --warmup 1 --no-cpu --zstd-seq 1 > $O/zs_zref_noseq.json 2> $O/zs_zref_noseq.err
python - <<'P'
import json
for f in ("zs_zref_seq", "zs_zref_noseq"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("ms_per_step"), d.get("kernels"), d.get("decode_errors"), d.get("roundtrip_verified"))
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-600:])
P
rm -rf $O/zs_prof && rocprofv3 --kernel-trace --stats -d $O/zs_prof -o zs --output-format csv -- python bench.py --zref-only --gib 8 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/zs_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3), "ms", r["Percentage"])
P
