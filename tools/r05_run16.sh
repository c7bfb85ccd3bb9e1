#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5: probe-batch sizes of the LZ4 encoder again, on the deferred-emit kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
bash tools/ab_enc.sh 8 base e_b6 e_b8 e_b12 e_b10_24 base 2>&1 | grep "===\|profile=1" > $O/r05_enc3_batch.txt
cat $O/r05_enc3_batch.txt
