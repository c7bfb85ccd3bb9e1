#!/bin/bash
# Runs ON THE GPU BOX: the drop-in API legs at other chunk sizes than the bench's (the reference's default is 4 MiB, lib/lz4-mt_compress.c:114)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
{
for codec in lz4 zstd; do for c in 131072 1048576 4194304 16777216; do
  zstdmt_amd/bin/api_bench $codec 8589934592 $c zstdmt_amd/lib/libzstdmt_amd.so 1 2>&1 | tail -1
done; done
} | tee $O/r06_api_chunks.txt
