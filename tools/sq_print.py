#!/usr/bin/env python3
"""Print per-kernel averages of the counters found under the given rocprofv3 output directories
(developer tool; used by tools/sq_dec.sh).   usage: sq_print.py <substring of kernel names> <dir> [<dir> ...]"""
import csv, glob, os, sys
from collections import defaultdict
pat = sys.argv[1]
tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for d in sys.argv[2:]:
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not files:
        continue
    f = max(files, key=os.path.getsize)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if pat in k:
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                n[k][row["Counter_Name"]] += 1
for k in sorted(tot):
    e = {c: tot[k][c] / n[k][c] for c in tot[k]}
    ins = sum(e.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"))
    print(k, "launches", max(n[k].values()))
    print("   " + "  ".join(f"{c[3:]}={v / 1e9:.3f}G" for c, v in sorted(e.items())))
    if ins:
        print(f"   total {ins / 1e9:.3f} G wave-instructions = {ins / (8 << 30):.3f} per output byte (8 GiB)")
