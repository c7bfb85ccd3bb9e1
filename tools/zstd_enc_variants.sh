#!/bin/bash
# Developer tool (GPU box): bench.py --codec zstd with every library under zstdmt_amd/lib/variants/
# (A/B builds of zstd_enc.hip, tools/variant_build.sh) -> one line per variant.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for so in zstdmt_amd/lib/variants/z*.so; do
  n=$(basename $so .so)
  ZMT_LIB=$PWD/$so timeout 400 python bench.py --only --no-cpu --codec zstd --steps 3 --warmup 1 2>gpurun_out/zv_$n.err | python -c "
import sys, json
d = [json.loads(l[7:]) for l in sys.stdin if l.startswith('DETAIL ')][0]
print('$n', 'enc_ms', d['kernels']['k_lz4_enc']['ms'], 'dec_ms', d['kernels']['k_lz4_dec']['ms'], 'ratio', d['config']['ratio'], 'verified', d['roundtrip_verified'], 'value', d['value'])
"
done
