#!/bin/bash
# Runs ON THE GPU BOX: the round's last profile pass after the LZ4 encoder changes -- default bench line, rocprofv3 stats + PMC passes
# of the lz4 legs (tools/profile_round.sh lz4), SQ counters of the lz4 kernels, then the whole GPU test suite and smoke().
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
bash tools/profile_round.sh lz4 > $O/r06b_profile_round.log 2>&1
tail -c 1200 $O/bench_default.json
bash tools/profile_sq.sh "lz4" > $O/r06b_profile_sq.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r06b_gputests_final.txt 2>&1; echo "pytest rc $?" >> $O/r06b_gputests_final.txt
tail -3 $O/r06b_gputests_final.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
