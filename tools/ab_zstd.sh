#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): A/B of zstd encoder library variants built by tools/variant_build.sh -- one line per
# (variant, level): encoder ms, decoder ms of its streams, ratio, round trip verified, value.
#   bash tools/ab_zstd.sh <name>[:<level>] ...      ("base" = the shipped library; level defaults to 1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for spec in "$@"; do
  L=${spec%%:*}; LV=1; [[ "$spec" == *:* ]] && LV=${spec##*:}
  if [ "$L" = base ]; then unset ZMT_LIB; else export ZMT_LIB=$GRAFT_REPO_ROOT/zstdmt_amd/lib/variants/$L.so; fi
  timeout 400 python bench.py --only --no-cpu --codec zstd --zstd-level $LV --steps 3 --warmup 1 2>gpurun_out/zv_$L.err | python -c "
import sys, json
d = [json.loads(l[7:]) for l in sys.stdin if l.startswith('DETAIL ')][0]
print('$L', 'level $LV', 'enc_ms', d['kernels']['k_lz4_enc']['ms'], 'dec_ms', d['kernels']['k_lz4_dec']['ms'], 'ratio', d['config']['ratio'], 'verified', d['roundtrip_verified'], 'value', d['value'])
"
done
