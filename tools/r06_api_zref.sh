#!/bin/bash
# Runs ON THE GPU BOX: (1) the LZ4MT API legs with the per-role pipeline times and the callbacks-alone bound, (2) SQ counters of
# the zstd decode of reference-written streams (sequence pre-pass + frame decoder)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/zstdmt_amd/lib:$LD_LIBRARY_PATH
for i in 1 2; do
GPUMT_TRACE=1 ZMT_API_BOUND=1 zstdmt_amd/bin/api_bench lz4 8589934592 131072 zstdmt_amd/lib/libzstdmt_amd.so 1 2>&1 | tail -12
done > $O/r06_api_lz4_trace.txt 2>&1
GPUMT_TRACE=1 ZMT_API_BOUND=1 zstdmt_amd/bin/api_bench zstd 8589934592 1048576 zstdmt_amd/lib/libzstdmt_amd.so 1 >> $O/r06_api_lz4_trace.txt 2>&1
cat $O/r06_api_lz4_trace.txt
bash tools/sq_any.sh zref zstd python bench.py --only --zref-only --gib 8 --steps 2 --warmup 1 --no-cpu
