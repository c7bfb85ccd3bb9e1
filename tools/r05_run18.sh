#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5: LZ4 encoder, one way out of the batch loop (found flag) against the gotos.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
bash tools/ab_enc.sh 8 base e_prevcf base e_prevcf 2>&1 | grep "===\|profile=1" > $O/r05_enc3_cf.txt
cat $O/r05_enc3_cf.txt
timeout 600 python -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -n 2
