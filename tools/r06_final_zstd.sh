#!/bin/bash
# Runs ON THE GPU BOX: the zstd legs' profile pass after the encoder's last change, the default bench line, then the whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
bash tools/profile_round.sh zstd > $O/r06c_profile_round.log 2>&1
bash tools/profile_sq.sh "zstd" > $O/r06c_profile_sq.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r06c_gputests_final.txt 2>&1; echo "pytest rc $?" >> $O/r06c_gputests_final.txt
tail -3 $O/r06c_gputests_final.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
