#!/bin/bash
# Runs ON THE GPU BOX: the window encoder with its scalar instructions trimmed (default build) against -DENC5_BRANCHY
# (tools/variant_build.sh lz4_enc5.hip branchy "-DENC5_BRANCHY"), then the LZ4 device tests on the default build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do bash tools/ab_enc5.sh base branchy; done
python -m pytest tests/test_gpu_lz4.py -m gpu -x -q 2>&1 | tail -3
