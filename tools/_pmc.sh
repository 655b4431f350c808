#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc}
mkdir -p $O
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --cpu-reads 0 ${2} > $O/pmc_sq.log 2>&1
python3 - <<PY
import csv, glob, os
fs = sorted(glob.glob("$O/pmc_sq/*/*counter_collection.csv"), key=os.path.getmtime)
acc = {}
for row in csv.DictReader(open(fs[-1])):
    n = row["Kernel_Name"]
    for k in ("k_event_detect", "k_event_build"):
        if k in n:
            acc.setdefault(k, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for k, cs in acc.items():
    print(k, {c: "%.4g" % (sum(v.values()) / len(v)) for c, v in cs.items()})
PY
