#!/bin/bash
# Runs ON THE GPU BOX: the zstd sequence pre-pass at 6 (default) / 5 / 4 / 3 resident waves per CU (GPUMT_ZSEQ_PAD = dynamic-LDS padding):
# does a wave run faster with fewer neighbours on its SIMD?  (reference-written streams, 8 GiB; kernel times from rocprofv3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out
for pad in 0 5500 13500 26700; do
  rm -rf $O/zseq_$pad
  GPUMT_ZSEQ_PAD=$pad rocprofv3 --kernel-trace --stats --output-format csv -d $O/zseq_$pad -- python bench.py --only --zref-only --gib 8 --steps 2 --warmup 1 --no-cpu > /dev/null 2> $O/zseq_$pad.err
  f=$(ls -S $O/zseq_$pad/*/*kernel_stats.csv | head -1)
  echo "pad $pad: $(grep -E 'zmt_zstd_seq_kernel|zmt_zstd_dec_small_kernel' $f | awk -F, '{printf "%s avg %.3f ms; ", $1, $4/1e6}')"
done | tee $O/r06_zseq_waves.txt
