#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass: python tools/pmc_kernels.py <rocprof output dir> [name filter]"""
import csv
import glob
import json
import os
import re
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
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
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
