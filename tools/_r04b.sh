#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
for ARGS in "" "--rna 1"; do export BENCH_ARGS="$ARGS"; bash tools/_bench_one.sh base unb | sed "s/^/$ARGS /" | tee -a $O/log.txt; done
SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_unb.so timeout 600 python -m pytest tests/test_gpu_event.py -x -q 2>&1 | tail -3 | tee -a $O/log.txt
