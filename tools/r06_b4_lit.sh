#!/bin/bash
# Runs ON THE GPU BOX: brotli-mt decompress leg (8 GiB, reference-written level-1 streams), library variants A/B
#   bash tools/r06_b4_lit.sh <variant> [<variant> ...]   ("base" = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  if [ $L = base ]; then unset ZMT_LIB; else export ZMT_LIB=$GRAFT_REPO_ROOT/zstdmt_amd/lib/variants/$L.so; fi
  python bench.py --only --codec brotli --steps 2 --warmup 1 --no-cpu --no-encoder 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L brotli', d['value'], d['ms_per_step'], d['roundtrip_verified'])"
done
unset ZMT_LIB
python -m pytest tests/test_gpu_brotli.py tests/test_gpu_brotlimt_api.py -m gpu -x -q 2>&1 | tail -3
