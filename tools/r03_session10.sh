#!/bin/bash
# round-3 GPU session 10: fuzz of the final kernels against the oracle on the device (LZ4 every level + both decode
# variants; the four drop-in APIs), then smoke() and the default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s10; rm -rf $O; mkdir -p $O
timeout 400 python tools/gpu_fuzz_lz4.py 240 7000 > $O/fuzz_lz4.txt 2>&1; tail -2 $O/fuzz_lz4.txt
timeout 300 python tools/gpu_fuzz_api.py 150 7000 > $O/fuzz_api.txt 2>&1; tail -2 $O/fuzz_api.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
