#!/bin/bash
# Runs ON THE GPU BOX: the zstd / brotli / snappy legs (kernel times, HIP events) -- A/B of the shared copy helpers
cd "$GRAFT_REPO_ROOT"
python bench.py --only --codec zstd --steps 3 --warmup 1 --no-cpu 2>&1 | grep "^DETAIL" | sed "s/^DETAIL //" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('zstd own', d['value'], d['kernels'], d['roundtrip_verified'])"
python bench.py --only --zref-only --gib 8 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('zstd ref', d['ms_per_step'], d['roundtrip_verified'])"
python bench.py --only --codec brotli --steps 2 --warmup 1 --no-cpu --no-encoder 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('brotli', d['value'], d['ms_per_step'], d['roundtrip_verified'])"
python bench.py --only --codec snappy --steps 2 --warmup 1 --no-cpu 2>&1 | grep "^DETAIL" | sed "s/^DETAIL //" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('snappy', d['value'], d['kernels'], d['roundtrip_verified'])"
