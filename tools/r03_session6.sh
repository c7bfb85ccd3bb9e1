#!/bin/bash
# round-3 GPU session 6: zstd level tiers (parity + timing), snappy encoder hash width A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstdmt_api.py -x -q -m gpu > $O/pytest_zstd.txt 2>&1
tail -3 $O/pytest_zstd.txt
for lv in 1 3 10; do
  timeout 300 python bench.py --codec zstd --only --no-cpu --steps 2 --warmup 1 --zstd-level $lv > $O/zstd_l$lv.json 2> $O/zstd_l$lv.err
done
for v in sn44 sn55; do
  ZMT_LIB=$PWD/zstdmt_amd/lib/variants/$v.so timeout 300 python bench.py --codec snappy --gib 8 --steps 2 --warmup 1 > $O/snappy_$v.json 2> $O/snappy_$v.err
done
timeout 300 python bench.py --codec snappy --gib 8 --steps 2 --warmup 1 > $O/snappy_66.json 2> $O/snappy_66.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s6/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "c", d.get("compress_MBps"), "d", d.get("decompress_MBps"), "ratio", d["config"]["ratio"], {k:v["ms"] for k,v in d["kernels"].items() if k in ("k_lz4_enc","k_lz4_dec","compress","decompress")}, d["roundtrip_verified"], d["decode_errors"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
