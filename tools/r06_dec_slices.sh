#!/bin/bash
# (the slicing this drives was removed again after the measurement: profiles/r06_sweeps/lz4_dec_slices.txt; the script
# works on the commit "LZ4 decode in slices on internal streams")
# Runs ON THE GPU BOX: the decompress-only leg (8 GiB, configs[2]) with the batch's records in S slices on the decode path's two
# internal streams (GPUMT_LZ4_DEC_SLICES): parse under copy, verify under copy.  decompress_leg = probe + decode + verify (HIP events)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
for S in ${@:-1 2 4 8 16 1 4}; do
  GPUMT_LZ4_DEC_SLICES=$S python bench.py --only --mode decompress --steps 6 --warmup 2 --no-cpu 2>&1 | grep "^DETAIL " | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['per_rank_ms'][0]
print('slices $S: decompress leg %.3f ms (wall per step %.3f), decode timer %.3f ms, value %.1f GB/s, verified %s, errors %s' % (r['decompress_leg'], d['ms_per_step'], r['k_dec'], d['value']/1e3, d['roundtrip_verified'], d['decode_errors']))"
done | tee $O/r06_dec_slices.txt
