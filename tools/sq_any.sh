#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): SQ instruction-mix counters of the kernels whose name contains <pattern> while
# <command> runs (two separate rocprofv3 --pmc passes, counters only).   usage: bash tools/sq_any.sh <tag> <pattern> <command...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=$1; P=$2; shift; shift
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"
B="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
for pass in A B; do
  rm -rf $O/sq_${T}_${pass}
  eval "ctr=\$$pass"
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/sq_${T}_${pass} -- "$@" > /dev/null 2> $O/sq_${T}_${pass}.err
done
python tools/sq_print.py "$P" $O/sq_${T}_A $O/sq_${T}_B | tee $O/sq_${T}.txt
