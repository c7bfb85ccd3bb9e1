#!/bin/bash
# (the slicing this drives was removed again after the measurement: profiles/r06_sweeps/lz4_dec_slices.txt; the script
# works on the commit "LZ4 decode in slices on internal streams")
# Runs ON THE GPU BOX: kernel timeline (rocprofv3 --kernel-trace) of the decompress-only leg with S slices
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out; S=${1:-4}
rm -rf $O/tl_dec_$S
GPUMT_LZ4_DEC_SLICES=$S rocprofv3 --kernel-trace --output-format csv -d $O/tl_dec_$S -- python bench.py --only --mode decompress --steps 1 --warmup 1 --no-cpu --no-verify > /dev/null 2> $O/tl_dec_$S.err
python - <<PY
import csv, glob
f = max(glob.glob("$O/tl_dec_$S/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getsize(p))
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode: from the last zmt_dec_nblk_kernel on
idx = max(i for i, r in enumerate(rows) if "zmt_dec_nblk" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + 40]:
    print("%-32s queue %s  start %8.3f ms  end %8.3f ms  (%7.3f ms)  grid %s" % (r["Kernel_Name"][:32], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
