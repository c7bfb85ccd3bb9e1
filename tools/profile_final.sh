#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), the profile round on a final tree -- default bench line, rocprofv3 stats and
# PMC passes of all codecs (tools/profile_round.sh), SQ counters (tools/profile_sq.sh), then the GPU test suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
bash tools/profile_round.sh all > $O/r06_profile_round.log 2>&1
tail -c 1500 $O/bench_default.json
bash tools/profile_sq.sh "lz4 zstd brotli" > $O/r06_profile_sq.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r06_gputests_final.txt 2>&1; echo "pytest rc $?" >> $O/r06_gputests_final.txt
tail -3 $O/r06_gputests_final.txt
