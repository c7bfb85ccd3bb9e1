#!/bin/bash
# round-3 GPU session 5: full GPU suite after the wave.h changes (ballot builtin, fused DPP scans), decode timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s5; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
B="python bench.py --only --mode decompress --steps 5 --warmup 1 --no-cpu"
timeout 300 $B --dec-variant 0 > $O/dec_v0.json 2> $O/dec_v0.err
for r in 12 13; do
  timeout 300 $B --dec-variant 2 --lz4-ring $r > $O/dec_v2_r$r.json 2> $O/dec_v2_r$r.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s5/dec_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["kernels"], d["roofline_decompress"]["frac"], d["decode_errors"], d["roundtrip_verified"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
