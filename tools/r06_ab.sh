#!/bin/bash
# Runs ON THE GPU BOX: round-6 A/B lines.  usage: bash tools/r06_ab.sh enc|dec <args>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
what=$1; shift
if [ $what = enc ]; then
  # LZ4 fast encoder: window encoder (default) against the probe-batch encoder, 8 GiB
  for v in 0 3; do
    echo "=== GPUMT_LZ4_ENC=$v"
    GPUMT_LZ4_ENC=$v python tools/enc_prof.py 8 2>&1 | grep -E "profile=1"
  done
else
  NOSQ=1 bash tools/ab_libs.sh 0 "$@"
fi
