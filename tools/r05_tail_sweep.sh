#!/bin/bash
# development, on the GPU box: k_event ms with the tail split off / on over batch sizes (dev build, 100 000-sample reads)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1
mkdir -p $O
export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_dev.so
for n in 500 1000 2000 3500 4000 5000 7000 8000 9300 10000 11000 12000 14000 20000; do
for t in -1 0; do
python tools/event_phases.py --steps 6 --tail $t --reads $n --modes 0,0 > $O/n${n}_t$t.json 2>>$O/err.txt
python3 -c "import json; d=json.loads(open('$O/n${n}_t$t.json').read()); print($n, 'tail', $t, [m['k_event_ms'] for m in d['modes']])"
done; done
unset SIGTK_AMD_LIB
for a in "--ragged 0.8" "--ragged 0.8 --rna 1" "--read-len 5000 --reads 200000" "--read-len 5000 --reads 200000 --rna 1"; do
for lib in r04 base; do
if [ $lib = base ]; then unset SIGTK_AMD_LIB SIGTK_AMD_LIB_ANY; else export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_$lib.so SIGTK_AMD_LIB_ANY=1; fi
python bench.py --cpu-reads 0 --steps 20 --warmup 5 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$a', d['ms_per_step'], d['roofline']['kernels_ms'])"
done; done
