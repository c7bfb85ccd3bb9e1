#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5 call 1: GPU test suite, zstd encoder variants, default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_gputests.txt 2>&1; echo "pytest rc $?" >> $O/r05_gputests.txt
tail -3 $O/r05_gputests.txt
bash tools/zstd_enc_variants.sh > $O/r05_zstd_variants.txt 2>&1
echo "shipped:" >> $O/r05_zstd_variants.txt
timeout 120 python bench.py --only --no-cpu --codec zstd --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('shipped', 'enc_ms', d['kernels']['k_lz4_enc']['ms'], 'dec_ms', d['kernels']['k_lz4_dec']['ms'], 'ratio', d['config']['ratio'], 'verified', d['roundtrip_verified'], 'value', d['value'])
" >> $O/r05_zstd_variants.txt
cat $O/r05_zstd_variants.txt
timeout 600 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
tail -c 3000 $O/r05_bench_default.json
