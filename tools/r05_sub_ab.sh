#!/bin/bash
# development: bench_subtools of library variants side by side on one box: tools/r05_sub_ab.sh base nt ...   (SUB_FILTER=jnn)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset SIGTK_AMD_LIB SIGTK_AMD_LIB_ANY; else export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_$v.so SIGTK_AMD_LIB_ANY=1; fi
  echo "== $v"
  python tools/bench_subtools.py --reads ${SUB_READS:-125000} --rna 0 --steps 5 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    if '${SUB_FILTER}' in d.get('subtool'): print(d.get('subtool'), d.get('ms'), d.get('kernels_ms'))"
done
