#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_event.py -x -q 2>&1 | tail -5 | tee $O/test.txt
python bench.py --cpu-reads 200 --steps 10 2>&1 | tail -1 | cut -c1-1200 | tee $O/bench.txt
export BENCH_ARGS=""; bash tools/_bench_one.sh base | tee -a $O/log.txt
