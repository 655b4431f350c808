#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_stat_long.py tests/test_gpu_stat.py -x -q 2>&1 | tail -15
for N in 2000 20000; do
python tools/bench_subtools.py --reads $N --rna 0 --one-long 3000001 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print($N, d['subtool'], d['ms'], {k:v for k,v in d['kernels_ms'].items() if v>0.05}, d['long_reads'])"
done
python tools/bench_subtools.py --reads 20000 --rna 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('uniform', d['subtool'], d['ms'], {k:v for k,v in d['kernels_ms'].items() if v>0.05}, d['long_reads'])"
