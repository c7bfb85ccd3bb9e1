#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): resident-wave cap A/B of the LZ4 copy stage (round 6).
# zmt_dec_copy3_w4_kernel declares 6 224 B of LDS = 26 workgroups (waves) per CU; dynamic-LDS padding lowers that
# to 24 / 20 / 16 / 12 / 8, i.e. 6 144 ... 2 048 live 64 KiB windows per chip (384 MiB ... 128 MiB against the 256 MiB
# Infinity Cache).  Per cap: kernel times (HIP events, tools/dec_prof.py) and FETCH_SIZE of the copy kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_wavecap; mkdir -p $O
for cap in 26:0 24:340 20:1584 16:3420 12:6384 8:11988; do
  w=${cap%%:*}; pad=${cap##*:}
  echo "=== waves per CU $w (pad $pad B)"
  GPUMT_LZ4_DEC_PAD=$pad python tools/dec_prof.py 8 0 2>&1 | grep -E "variant|split"
  rm -rf $O/fetch_$w
  GPUMT_LZ4_DEC_PAD=$pad rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$w -- \
      python bench.py --only --mode decompress --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/fetch_$w.err
  python - <<PY
import csv, glob
for f in glob.glob("$O/fetch_$w/**/*counter_collection.csv", recursive=True):
    tot = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "copy3" in r["Kernel_Name"]:
            tot[r["Kernel_Name"]] = tot.get(r["Kernel_Name"], 0) + float(r["Counter_Value"])
    for k, v in tot.items():
        print("   FETCH_SIZE %s: raw %.1f MiB, x2 (gfx950 correction) = %.2f GB per launch" % (k, v / 1024, v * 1024 * 2 / 1e9))
PY
done
