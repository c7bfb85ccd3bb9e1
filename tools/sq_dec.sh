#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): instruction-mix and wait counters of the LZ4 decode kernels, separate
# rocprofv3 --pmc passes of `bench.py --mode decompress` (counters only, --kernel-trace).  Environment selects
# the kernels (GPUMT_LZ4_PARSE / GPUMT_LZ4_COPY = 3 | 4).   usage: bash tools/sq_dec.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-dec}
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"
B="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
for pass in A B; do
  rm -rf $O/sq_${T}_${pass}
  eval "ctr=\$$pass"
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/sq_${T}_${pass} -- \
      python bench.py --only --mode decompress --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/sq_${T}_${pass}.err
done
python tools/sq_print.py zmt_dec $O/sq_${T}_A $O/sq_${T}_B | tee $O/sq_${T}.txt
