#!/bin/bash
# development, on the GPU box: A/B/A/B of library builds on one box: tools/_ab.sh <out dir> <bench args...> -- <tag> <tag> ...
# (tags: base = the shipped build, else sigtk_amd/_variants/libsigtk_gpu_<tag>.so)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$1; shift
mkdir -p $O
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset SIGTK_AMD_LIB SIGTK_AMD_LIB_ANY; else export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_$v.so SIGTK_AMD_LIB_ANY=1; fi
  timeout -s KILL 300 python bench.py --cpu-reads 0 --steps 20 --warmup 5 "${ARGS[@]}" 2>/dev/null | tail -1 > $O/ab_${v}_$rep.json
  python3 -c "import sys,json; d=json.loads(open('$O/ab_${v}_$rep.json').read()); print('$v', $rep, d['ms_per_step'], d['roofline']['kernels_ms'], d['config'].get('split_reads'))"
done
done
