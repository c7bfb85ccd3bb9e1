#!/bin/bash
# round-3 GPU session 1 (runs ON THE GPU BOX via gpurun): design measurements before the decoder rework
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s1; rm -rf $O; mkdir -p $O
timeout 120 tools/ubench/store_cost > $O/store_cost.txt 2>&1
B="python bench.py --only --steps 1 --warmup 0 --no-cpu --no-verify"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/prof_tcc1 -- $B > $O/bench_tcc1.json 2> $O/prof_tcc1.err
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/prof_tcc2 -- $B > $O/bench_tcc2.json 2> $O/prof_tcc2.err
timeout 300 python bench.py --codec snappy --gib 8 --snappy-dec 0 > $O/snappy0.json 2> $O/snappy0.err
timeout 300 python bench.py --codec snappy --gib 8 --snappy-dec 1 > $O/snappy1.json 2> $O/snappy1.err
timeout 600 python -m pytest tests/test_gpu_snappy.py -x -q -m gpu > $O/pytest_snappy.txt 2>&1
tail -3 $O/pytest_snappy.txt; cat $O/store_cost.txt; cat $O/snappy0.json $O/snappy1.json | cut -c1-600
