#!/bin/bash
# round-3 GPU session 11: zstd decoder with aligned LDS window reads -- parity, timing, reference-written streams
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s11; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu > $O/pytest_zstd.txt 2>&1
tail -2 $O/pytest_zstd.txt
timeout 300 python bench.py --only --codec zstd --steps 3 --warmup 1 --no-cpu > $O/zstd.json 2> $O/zstd.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s11/zstd.json").read().strip().splitlines()[-1])
print(d["value"], d["kernels"], d.get("decompress_MBps"), d.get("decode_errors"), d.get("roundtrip_verified"))
PY
timeout 300 python tests/perf_zstd_decode.py > $O/perf_ref.txt 2>&1; tail -5 $O/perf_ref.txt
