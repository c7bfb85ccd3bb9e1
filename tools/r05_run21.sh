#!/bin/bash
# Runs ON THE GPU BOX (through gpurun), round 5: brotli / snappy encoders with the look-ahead: bench legs + their GPU tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 600 python bench.py --only --no-cpu --codec brotli --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = [json.loads(l[7:]) for l in sys.stdin if l.startswith('DETAIL ')][0]
print('brotli', d['value'], json.dumps(d.get('device_encoder')))
" | tee $O/r05_brotli_snappy_enc.txt
timeout 600 python bench.py --only --no-cpu --codec snappy --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = [json.loads(l[7:]) for l in sys.stdin if l.startswith('DETAIL ')][0]
print('snappy', d['value'], d['compress_MBps'], d['decompress_MBps'], d['config']['ratio'], d['roundtrip_verified'], {k: v['ms'] for k, v in d['kernels'].items()})
" | tee -a $O/r05_brotli_snappy_enc.txt
timeout 900 python -m pytest tests/test_gpu_brotli.py tests/test_gpu_brotlimt_api.py tests/test_gpu_snappy.py -x -q 2>&1 | tail -n 2
