#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stat_long.py tests/test_gpu_stat.py tests/test_gpu_cli.py tests/test_gpu_job.py tests/test_gpu_shims.py -x -q 2>&1 | tail -3
python tools/bench_subtools.py --reads 20000 --rna 0 > $O/uniform.json 2>/dev/null
python tools/bench_subtools.py --reads 20000 --rna 0 --one-long 3000001 > $O/one_long.json 2>/dev/null
python tools/bench_subtools.py --reads 20000 --rna 0 --ragged 0.8 > $O/ragged.json 2>/dev/null
python tools/bench_subtools.py --reads 2000 --rna 0 --one-long 3000001 > $O/one_long_2000.json 2>/dev/null
python tools/bench_subtools.py --reads 20000 --rna 1 --one-long 3000001 > $O/one_long_rna.json 2>/dev/null
python tools/bench_subtools.py --reads 20000 --rna 1 > $O/uniform_rna.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04z/*.json")):
    for l in open(f):
        d=json.loads(l)
        if d["subtool"] in ("prefix","jnn"): print(f.split("/")[-1][:-5], d["subtool"], d["ms"], {k:v for k,v in d["kernels_ms"].items() if v>0.05}, d["long_reads"] and list(d["long_reads"].values()))
PY
