#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5: LZ4 encoder with the sequences written out 64 at a time (ENC3_DEFER).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
bash tools/ab_enc.sh 8 e_nodefer base e_nodefer base > $O/r05_enc3_defer.txt 2>&1
cat $O/r05_enc3_defer.txt
timeout 600 python -m pytest tests/test_gpu_lz4.py tests/test_gpu_lz4mt_api.py -x -q 2>&1 | tail -n 2
