#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): A/B of encoder library variants built by tools/variant_build.sh.
#   bash tools/ab_enc.sh <GiB> <lib name> [<lib name> ...]     ("base" = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
G=$1; shift
for L in "$@"; do
  if [ $L = base ]; then unset ZMT_LIB; else export ZMT_LIB=$GRAFT_REPO_ROOT/zstdmt_amd/lib/variants/$L.so; fi
  echo "=== $L"
  python tools/enc_prof.py $G 2>&1 | grep -E "enc kernel|Mcycles|waves="
done
