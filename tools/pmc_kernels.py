#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass: python tools/pmc_kernels.py <rocprof output dir> [name filter] [samples]
With `samples` (samples per launch) the SQ counters are also given per sample: vector lane-instructions per sample, VALU
busy fraction (SQ_ACTIVE_INST_VALU * 4 cycles over 1024 SIMDs against SQ_BUSY_CYCLES / 32) and issue cycles per instruction."""
import csv
import glob
import json
import os
import re
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    samples = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit("no counter_collection.csv under " + d)
    acc = {}
    for row in csv.DictReader(open(fs[-1])):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void sgk::", "")
        if flt and flt not in k:
            continue
        acc.setdefault(k, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    out = {k: dict({c: round(sum(v.values()) / len(v)) for c, v in cs.items()}, dispatches=len(next(iter(cs.values()))))
           for k, cs in acc.items()}
    if samples:
        for k, v in out.items():
            if v.get("SQ_INSTS_VALU") and v.get("SQ_BUSY_CYCLES"):
                v["valu_lane_instructions_per_sample"] = round(v["SQ_INSTS_VALU"] * 64 / samples, 1)
                v["salu_instructions_per_64_samples"] = round(v["SQ_INSTS_SALU"] * 64 / samples, 1)
                v["valu_busy_fraction"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / (v["SQ_BUSY_CYCLES"] / 32 * 1024), 3)
                v["issue_cycles_per_valu_instruction"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / v["SQ_INSTS_VALU"], 2)
                v["kernel_ms_at_2.4GHz"] = round(v["SQ_BUSY_CYCLES"] / 32 / 2.4e6, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
