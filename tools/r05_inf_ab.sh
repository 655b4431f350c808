#!/bin/bash
# development: tools/bench_inflate.py of library variants on one box: tools/r05_inf_ab.sh base norun
cd $GRAFT_REPO_ROOT
for n in 1280 5120 20000; do
for v in "$@"; do
  if [ "$v" = base ]; then unset SIGTK_AMD_LIB SIGTK_AMD_LIB_ANY; else export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_$v.so SIGTK_AMD_LIB_ANY=1; fi
  echo -n "$v $n: "; python tools/bench_inflate.py --reads $n 2>/dev/null | tail -1
done; done
