#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): A/B of library variants built by tools/variant_build.sh.
#   bash tools/ab_libs.sh "<dec_prof variants>" <lib name> [<lib name> ...]     ("base" = the shipped library)
# For each library: kernel times of tools/dec_prof.py at 8 GiB, then the SQ counters of the decode kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1; shift
for L in "$@"; do
  if [ $L = base ]; then unset ZMT_LIB; else export ZMT_LIB=$GRAFT_REPO_ROOT/zstdmt_amd/lib/variants/$L.so; fi
  echo "=== $L"
  python tools/dec_prof.py 8 $V 2>&1 | grep -E "variant|split"
  [ -n "$NOSQ" ] || bash tools/sq_dec.sh $L 2>/dev/null | grep -A2 -E "copy|parse"
done
