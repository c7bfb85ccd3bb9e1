#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): instruction-mix and wait counters of the dominant kernels, in
# separate rocprofv3 --pmc passes (counters only, --kernel-trace; never combined with sys/runtime traces).
#   gpurun --timeout 900 -- 'bash tools/profile_sq.sh [codec]'     then    python tools/sq_summarize.py r01
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"
B="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
for codec in ${1:-lz4 zstd brotli}; do
  for pass in A B; do
    rm -rf $O/sq_${codec}_${pass}
    eval "ctr=\$$pass"
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/sq_${codec}_${pass} -- \
        python bench.py --only --codec $codec --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/sq_${codec}_${pass}.err
  done
done
ls $O | grep sq_
