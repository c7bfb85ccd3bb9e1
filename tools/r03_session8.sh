#!/bin/bash
# round-3 GPU session 8: the whole GPU test suite, then the profile round (bench line, kernel-trace stats,
# FETCH_SIZE / WRITE_SIZE passes per codec) and the SQ instruction-mix passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 1500 bash tools/profile_round.sh > $O/profile_round.txt 2>&1
timeout 700 bash tools/profile_sq.sh > $O/profile_sq.txt 2>&1
tail -5 $O/profile_sq.txt
