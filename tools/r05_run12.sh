#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5: zstd encoder with the step's matches chosen in vector code (ZE_VPICK).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
one() { # name, lib ("" = shipped), level
  if [ -n "$2" ]; then export ZMT_LIB=$PWD/zstdmt_amd/lib/variants/$2.so; else unset ZMT_LIB; fi
  timeout 400 python bench.py --only --no-cpu --codec zstd --zstd-level $3 --steps 3 --warmup 1 2>$O/zv_$1.err | python -c "
import sys, json
d = [json.loads(l[7:]) for l in sys.stdin if l.startswith('DETAIL ')][0]
print('$1', 'level $3', 'enc_ms', d['kernels']['k_lz4_enc']['ms'], 'dec_ms', d['kernels']['k_lz4_dec']['ms'], 'ratio', d['config']['ratio'], 'verified', d['roundtrip_verified'], 'value', d['value'])
"
}
( one scalar_loop z_novp 1; one shipped "" 1; one lookahead6 z_w6 1; one shipped_l3 "" 3; one scalar_loop_l3 z_novp 3; one shipped_l10 "" 10 ) > $O/r05_zstd_variants8.txt 2>&1
unset ZMT_LIB
cat $O/r05_zstd_variants8.txt
python tools/zstd_enc_prof.py 2 2>&1 | tail -n 2 | tee $O/r05_zstd_enc_prof2.txt
timeout 300 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstdmt_api.py -x -q 2>&1 | tail -n 2
