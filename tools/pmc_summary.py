#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/refresh_profiles.sh into profiles/pmc_traffic.json.

usage: python tools/pmc_summary.py gpurun_out/<tag> [out.json]
Reads <tag>/pmc_fetch, pmc_write, pmc_sq (counter_collection.csv of the newest run in each), averages every counter
over the dispatches of each event kernel, and prices the HBM traffic as MI355X_MICROARCH.md (HBM) prescribes (units:
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE tallies 128-byte requests at 64 bytes)."""
import csv
import glob
import json
import os
import subprocess
import sys

KERNELS = ("k_event_detect", "k_event_build", "k_event_fallback", "k_event")


def newest(d):
    fs = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        raise SystemExit("no counter_collection.csv under " + d)
    return fs[-1]


def kname(name):
    # longest match first: "k_event<" is the fused kernel, "k_event_detect<" etc. the separate ones
    for k in ("k_event_detect", "k_event_build", "k_event_fallback"):
        if k in name:
            return k
    return "k_event" if "k_event" in name else None


def per_kernel(path):
    acc = {}
    for row in csv.DictReader(open(path)):
        k = kname(row["Kernel_Name"])
        if k:
            acc.setdefault(k, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    tag = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    fetch = per_kernel(newest(os.path.join(tag, "pmc_fetch")))
    write = per_kernel(newest(os.path.join(tag, "pmc_write")))
    sq = per_kernel(newest(os.path.join(tag, "pmc_sq")))
    bench = json.loads(open(os.path.join(tag, "bench.json")).read().strip().splitlines()[-1])
    S = bench["config"]["samples_per_gpu"]
    E = bench["config"]["events_per_step_rank0"]
    alg = bench["roofline"]["algorithmic_bytes"]
    main_k = "k_event" if "k_event" in fetch else "k_event_detect"
    raw = {k: {"FETCH_SIZE_KiB": round(fetch[k]["FETCH_SIZE"]), "WRITE_SIZE_KiB": round(write[k]["WRITE_SIZE"])}
           for k in fetch if k != "k_event_fallback"}
    rd_raw = sum(v["FETCH_SIZE_KiB"] for v in raw.values()) * 1024
    wr = sum(v["WRITE_SIZE_KiB"] for v in raw.values()) * 1024
    # FETCH_SIZE tallies every 128-byte fabric request at 64 bytes, whatever the shape of the access
    # (tools/fetch_calib.hip, profiles/archive/r03_fetch_calibration.json: coalesced 16 / 32 / 64 bytes per lane, and 32 bytes
    # per lane with every lane on its own stream as in the detector -- the request count changes with the shape, the
    # bytes per request do not): the bytes that crossed the fabric are 2 x the counter.  Nothing is priced.
    rd = 2 * rd_raw
    busy = {k: v["SQ_ACTIVE_INST_VALU"] * 4 / v["SQ_BUSY_CYCLES"] * 32 / 1024 for k, v in sq.items() if "SQ_BUSY_CYCLES" in v}
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(__file__)).decode().strip()
    except Exception:
        commit = None
    doc = {
        "_comment": "HBM bytes per bench step (config 2: 1e9 samples) from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                    "collected in separate passes (tools/refresh_profiles.sh + tools/pmc_summary.py; mean over the "
                    "dispatches of the run).  Counter unit is KiB.  FETCH_SIZE tallies 128-byte requests at 64 bytes "
                    "(MI355X_MICROARCH.md, HBM); calibrated for this kernel's access shapes by tools/fetch_calib.hip "
                    "(profiles/archive/r03_fetch_calibration.json): the factor is 2 for all of them, so read bytes = 2 x the "
                    "counter -- measured, not priced.  WRITE_SIZE is exact for wide stores.",
        "recorded_for": {"commit": commit, "source": tag, "kernel": main_k},
        "raw": raw,
        "read_bytes": {"counter": rd_raw, "calibration_factor": 2.0, "measured": rd},
        "write_bytes": {"counted": wr},
        "hbm_bytes_per_step": rd + wr,
        "algorithmic_bytes_per_step": alg,
        "traffic_over_algorithmic": round((rd + wr) / alg, 3),
        "valu_wave_instructions_per_step": round(sum(v.get("SQ_INSTS_VALU", 0) for k, v in sq.items() if k != "k_event_fallback")),
        "valu_lane_instructions_per_sample": round(sum(v.get("SQ_INSTS_VALU", 0) for k, v in sq.items()
                                                       if k != "k_event_fallback") * 64 / S, 1),
        "salu_wave_instructions_per_step": round(sum(v.get("SQ_INSTS_SALU", 0) for k, v in sq.items() if k != "k_event_fallback")),
        "valu_busy_fraction": {k: round(v, 3) for k, v in busy.items()},
        "sq_counters_per_step": {k: {c: round(v) for c, v in cs.items()} for k, cs in sq.items()},
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
