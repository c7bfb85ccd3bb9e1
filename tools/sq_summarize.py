#!/usr/bin/env python3
"""Fold gpurun_out/sq_<codec>_<pass> (tools/profile_sq.sh) into profiles/rNN_sq_counters.json:
per kernel and launch the instruction mix (VALU / SALU / LDS / VMEM / SMEM / branch), the instructions
per byte of algorithmic traffic and the share of wave cycles spent waiting."""
import csv, glob, json, os, sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
go = os.path.join(root, "gpurun_out")
KEEP = ("zmt_lz4_enc3_p17_kernel", "zmt_lz4_enc5_p17_kernel", "zmt_dec_parse3_kernel", "zmt_dec_parse4_kernel", "zmt_brotli_dec4_kernel", "zmt_dec_copy3_w4_kernel", "zmt_dec_copy3_w8_kernel", "zmt_dec_copy3_w16_kernel", "zmt_zstd_enc_kernel",
        "zmt_zstd_dec_small_kernel", "zmt_zstd_seq_kernel", "zmt_brotli_dec_kernel", "zmt_brotli_enc_kernel")
out = {"_what": "rocprofv3 --pmc passes (tools/profile_sq.sh) over the bench commands, 8 GiB per launch; counter "
                "values are summed over the device per launch, averaged over the launches of a run",
       "kernels": {}}
for codec in ("lz4", "zstd", "brotli"):
    for p in ("A", "B"):
        files = glob.glob(os.path.join(go, f"sq_{codec}_{p}", "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        newest = max(os.path.getmtime(x) for x in files)   # newest run, its largest file = the bench process
        f = max((x for x in files if newest - os.path.getmtime(x) < 120), key=os.path.getsize)
        tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if k in KEEP:
                    tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    n[k][row["Counter_Name"]] += 1
        for k in tot:
            e = out["kernels"].setdefault(k, {})
            for c in tot[k]:
                e[c] = tot[k][c] / n[k][c]
for k, e in out["kernels"].items():
    ins = sum(e.get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"))
    if ins:
        e["mix"] = {c[9:]: round(e.get(c, 0) / ins, 3) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                                                               "SQ_INSTS_VMEM", "SQ_INSTS_SMEM")}
        e["wave_instructions_per_uncompressed_byte"] = round(ins / (8 << 30), 3)
    if e.get("SQ_WAVE_CYCLES"):
        e["wait_inst_share_of_wave_cycles"] = round(e.get("SQ_WAIT_INST_ANY", 0) / e["SQ_WAVE_CYCLES"], 3)
json.dump(out, open(os.path.join(root, "profiles", f"{rnd}_sq_counters.json"), "w"), indent=1, sort_keys=True)
for k, e in sorted(out["kernels"].items()):
    print(k, e.get("mix"), e.get("wave_instructions_per_uncompressed_byte"), e.get("wait_inst_share_of_wave_cycles"))
