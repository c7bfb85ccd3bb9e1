#!/bin/bash
# Runs ON THE GPU BOX: the API legs as the bench runs them (process unbound), with the engines' reader / writer threads bound to the
# device's host node (default) and not (GPUMT_NUMA=0)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
{
for rep in 1 2; do for numa in 1 0; do for c in "lz4 131072" "zstd 1048576" "brotli 1048576"; do
  set -- $c
  echo "== GPUMT_NUMA=$numa $1"
  GPUMT_NUMA=$numa GPUMT_TRACE=1 ZMT_API_BOUND=1 zstdmt_amd/bin/api_bench $1 8589934592 $2 zstdmt_amd/lib/libzstdmt_amd.so 1 2>&1 | grep "bound to\|callbacks alone\|api" | sort -u
done; done; done
} > $O/r06_api_numa2.txt 2>&1
cat $O/r06_api_numa2.txt
