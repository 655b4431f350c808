#!/usr/bin/env python3
"""Summarise ONE rocprofv3 --pmc pass of SQ counters (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY) per kernel of the event path: mean over the dispatches of the run.

usage: python tools/pmc_sq_summary.py <dir with */*counter_collection.csv> <bench json line file> <out.json> [note]"""
import csv
import glob
import json
import os
import subprocess
import sys


def kname(name):
    for k in ("k_event_multi", "k_event_seg", "k_event_fallback", "k_event_rec", "k_seg_plan", "k_order", "k_event"):
        if k in name:
            return k
    return None


def main():
    d, bench_path, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fs = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        raise SystemExit("no counter_collection.csv under " + d)
    acc = {}
    for row in csv.DictReader(open(fs[-1])):
        k = kname(row["Kernel_Name"])
        if k:
            acc.setdefault(k, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    sq = {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    bench = json.loads(open(bench_path).read().strip().splitlines()[-1])
    S = bench["config"]["samples_per_gpu"]
    doc = {
        "_comment": "rocprofv3 --pmc SQ counters per kernel of one bench.py step (mean over the dispatches of the run); "
                    "VALU busy = SQ_ACTIVE_INST_VALU * 4 / SQ_BUSY_CYCLES * 32 / 1024 as in profiles/pmc_traffic.json",
        "source": d, "note": sys.argv[4] if len(sys.argv) > 4 else "",
        "workload": bench["config"]["workload"], "ms_per_step": bench["ms_per_step"],
        "kernels_ms": bench["roofline"]["kernels_ms"],
        "valu_lane_instructions_per_sample": {k: round(v.get("SQ_INSTS_VALU", 0) * 64 / S, 1) for k, v in sq.items()},
        "salu_lane_instructions_per_sample": {k: round(v.get("SQ_INSTS_SALU", 0) * 64 / S, 1) for k, v in sq.items()},
        "valu_busy_fraction": {k: round(v["SQ_ACTIVE_INST_VALU"] * 4 / v["SQ_BUSY_CYCLES"] * 32 / 1024, 3)
                               for k, v in sq.items() if v.get("SQ_BUSY_CYCLES")},
        "sq_counters_per_step": {k: {c: round(v) for c, v in cs.items()} for k, cs in sq.items()},
    }
    try:
        doc["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(__file__) or ".").decode().strip()
    except Exception:
        pass
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1)[:1500])


if __name__ == "__main__":
    main()
