#!/bin/bash
# Runs ON THE GPU BOX: A/B of window-encoder builds (tools/variant_build.sh lz4_enc5.hip <name> "<flags>"), then the LZ4 device
# tests on the default build.   usage: bash tools/r06_enc5_lean.sh <variant> [<variant> ...]   ("base" = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do bash tools/ab_enc5.sh "$@"; done
python -m pytest tests/test_gpu_lz4.py -m gpu -x -q 2>&1 | tail -3
