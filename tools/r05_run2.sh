#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5 call 2: zstd encoder variants (look-ahead / repeat offsets), GPU tests,
# brotli API batch-size experiments.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
bash tools/zstd_enc_variants.sh > $O/r05_zstd_variants.txt 2>&1
timeout 400 python bench.py --only --no-cpu --codec zstd --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = [json.loads(l[7:]) for l in sys.stdin if l.startswith('DETAIL ')][0]
print('shipped', 'enc_ms', d['kernels']['k_lz4_enc']['ms'], 'dec_ms', d['kernels']['k_lz4_dec']['ms'], 'ratio', d['config']['ratio'], 'verified', d['roundtrip_verified'], 'value', d['value'])
" >> $O/r05_zstd_variants.txt
cat $O/r05_zstd_variants.txt
LIB=$PWD/zstdmt_amd/lib/libzstdmt_amd.so
AB=zstdmt_amd/bin/api_bench
( echo "default";               timeout 200 $AB brotli 8589934592 1048576 $LIB 1 2>&1 | tail -1
  echo "GPUMT_BATCH_MB=512";    GPUMT_BATCH_MB=512 timeout 200 $AB brotli 8589934592 1048576 $LIB 1 2>&1 | tail -1
  echo "GPUMT_BROTLI_DEC=2";    GPUMT_BROTLI_DEC=2 timeout 200 $AB brotli 8589934592 1048576 $LIB 1 2>&1 | tail -1
  echo "GPUMT_BATCH_MB=512 GPUMT_BROTLI_DEC=2"; GPUMT_BATCH_MB=512 GPUMT_BROTLI_DEC=2 timeout 200 $AB brotli 8589934592 1048576 $LIB 1 2>&1 | tail -1
  echo "GPUMT_BATCH_MB=1024 GPUMT_BROTLI_DEC=2 GPUMT_SLOTS=3"; GPUMT_SLOTS=3 GPUMT_BATCH_MB=1024 GPUMT_BROTLI_DEC=2 timeout 200 $AB brotli 8589934592 1048576 $LIB 1 2>&1 | tail -1
) > $O/r05_brotli_api.txt 2>&1
cat $O/r05_brotli_api.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_gputests.txt 2>&1; echo "pytest rc $?" >> $O/r05_gputests.txt
tail -3 $O/r05_gputests.txt
