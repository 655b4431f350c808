#!/usr/bin/env python3
"""Reads the rocprofv3 --pmc passes taken over tools/event_phases.py back into one table: counters per phase of k_event.

    python tools/event_phases_pmc.py <phases.json of a plain run> <out.json> <pmc dir> [<pmc dir> ...]

The k_event dispatches of every pass are, in dispatch order: 2 warm-ups, then `steps` per mode in the order of the plain
run's "modes".  Counters are summed over a dispatch's rows (XCCs / SEs) and averaged over the mode's dispatches, then
given per sample (x 64 lanes for instruction counters: lane-instructions per sample).  Phases by difference:
    detector + bitmap   = mode "detector+bitmap"
    builder walk        = "full-rounds (walk only)" - "detector+bitmap"
    event rounds, no arithmetic = "full-arith (raw events)" - "full-rounds (walk only)"
    create_event arithmetic + stores' second half = "full" - "full-arith (raw events)"
"""
import csv, glob, json, os, sys


def load(d):
    fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        raise SystemExit("no counter_collection.csv under " + d)
    disp = {}
    for row in csv.DictReader(open(fs[-1])):
        if "k_event<" not in row["Kernel_Name"] and "k_eventI" not in row["Kernel_Name"]:
            continue
        disp.setdefault(int(row["Dispatch_Id"]), {}).setdefault(row["Counter_Name"], 0.0)
        disp[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def main():
    plain = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    out = sys.argv[2]
    steps, modes = plain["steps"], plain["modes"]
    S = plain["reads"] * plain["read_len"]
    table = {m["name"]: {"k_event_ms": m["k_event_ms"]} for m in modes}
    for d in sys.argv[3:]:
        ds = load(d)[2:]
        if len(ds) < steps * len(modes):
            raise SystemExit("%s: %d k_event dispatches, expected >= %d" % (d, len(ds) + 2, 2 + steps * len(modes)))
        for i, m in enumerate(modes):
            grp = ds[i * steps:(i + 1) * steps]
            for c in grp[0]:
                table[m["name"]][c] = sum(g[c] for g in grp) / len(grp)
    per = {}
    for name, row in table.items():
        per[name] = {"k_event_ms": row["k_event_ms"]}
        for c, v in row.items():
            if c.startswith("SQ_INSTS") or c == "SQ_IFETCH":
                per[name][c + "_lane_per_sample" if "VALU" in c or c in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT") else c + "_per_64_samples"] = round(v * 64 / S, 3)
            elif c != "k_event_ms":
                per[name][c] = round(v)

    def diff(a, b):
        return {k: round(per[a][k] - per[b][k], 3) for k in per[a] if k in per[b] and isinstance(per[a][k], (int, float))}
    phases = {}
    if all(n in per for n in ("full", "detector+bitmap", "full-rounds (walk only)", "full-arith (raw events)")):
        phases = {"detector + bitmap": per["detector+bitmap"],
                  "builder: sample walk (tile loads, conversions, prefix sums, boundary records, scans)": diff("full-rounds (walk only)", "detector+bitmap"),
                  "builder: event rounds without create_event's arithmetic (LDS look-ups, shifts, carries, 16-byte stores)": diff("full-arith (raw events)", "full-rounds (walk only)"),
                  "builder: create_event arithmetic (reciprocal, two divisions, variance, sqrt)": diff("full", "full-arith (raw events)"),
                  "whole kernel": per["full"]}
    doc = {"_comment": __doc__.strip().splitlines()[0], "workload": "%d reads x %d samples, rna=%d, tail split %s" % (plain["reads"], plain["read_len"], plain["rna"], plain["tail_split"]),
           "samples": S, "per_mode": per, "phases_by_difference": phases, "sources": sys.argv[3:]}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(phases, indent=1))


if __name__ == "__main__":
    main()
