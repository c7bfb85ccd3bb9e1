#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5: zstd level 1 with 6 bytes hashed / minimum match 6 (ratio 2.588 on the emulator).
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
( one hash6_mm6 z_66 1; one shipped "" 1 ) > $O/r05_zstd_variants7.txt 2>&1
cat $O/r05_zstd_variants7.txt
