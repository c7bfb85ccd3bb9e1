#!/bin/bash
# Runs ON THE GPU BOX: zstd-mt level 1 round trip (configs[3], 8 GiB), library variants A/B (value, ratio, kernel ms)
#   bash tools/r06_zstd_enc_ab.sh <variant> [<variant> ...]   ("base" = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/zab.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1], "zstd", d["value"], d["config"].get("ratio"), d["kernels"], d["roundtrip_verified"])
PY
for L in "$@"; do
  if [ $L = base ]; then unset ZMT_LIB; else export ZMT_LIB=$GRAFT_REPO_ROOT/zstdmt_amd/lib/variants/$L.so; fi
  python bench.py --only --codec zstd --steps 3 --warmup 1 --no-cpu 2>&1 | grep "^DETAIL" | sed "s/^DETAIL //" | python /tmp/zab.py $L
done
