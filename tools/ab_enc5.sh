#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): A/B of library variants of the window encoder (tools/variant_build.sh lz4_enc5.hip ...).
#   bash tools/ab_enc5.sh <lib name> [<lib name> ...]     ("base" = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  if [ $L = base ]; then unset ZMT_LIB; else export ZMT_LIB=$GRAFT_REPO_ROOT/zstdmt_amd/lib/variants/$L.so; fi
  echo "=== $L: $(python tools/enc_prof.py 8 2>&1 | grep -E 'profile=1')"
done
