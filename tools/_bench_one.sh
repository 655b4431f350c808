#!/bin/bash
# prints "tag ms_per_step kernels" for a library variant
for v in "$@"; do
  if [ "$v" = base ]; then unset SIGTK_AMD_LIB; else export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_$v.so; fi
  python bench.py --cpu-reads 0 --steps 10 $BENCH_ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernels_ms'], d['config']['fallback_reads'], d['config']['rerun_chunks'])"
done
