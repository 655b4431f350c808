#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_event.py -x -q 2>&1 | tail -15 | tee $O/test.txt
python bench.py --cpu-reads 200 --steps 10 2>&1 | tail -1 | cut -c1-1500 | tee $O/bench.txt
SGK_EVENT_REC=0 python bench.py --cpu-reads 0 --steps 10 2>&1 | tail -1 | cut -c1-1500 | tee $O/bench_norec.txt
