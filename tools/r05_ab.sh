#!/bin/bash
# development, on the GPU box: the dev build's phases (tail split off / on, both priority policies), RNA parameters, and the
# shipped build against the round-4 library on the same box.  tools/r05_ab.sh <out dir> [notest]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1
mkdir -p $O
if [ "$2" != "notest" ]; then
python -m pytest tests/test_gpu_event.py tests/test_gpu_event_long.py tests/test_gpu_event_short.py tests/test_gpu_device_api.py tests/test_gpu_soak.py -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
fi
show() { python3 -c "import json,sys; d=json.loads(open('$1').read()); print('$2', [(m['name'][-14:], m['k_event_ms']) for m in d['modes']])"; }
export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_dev.so
for t in -1 0; do python tools/event_phases.py --steps 5 --tail $t --prio 0,1,0,1 > $O/tail_$t.json 2>$O/err.txt; show $O/tail_$t.json "tail $t"; done
python tools/event_phases.py --steps 5 --rna 1 --prio 0,1,0,1 > $O/rna.json 2>>$O/err.txt; show $O/rna.json rna
bench() { python bench.py --cpu-reads 0 --steps 30 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms'])"; }
export SIGTK_AMD_LIB=$PWD/sigtk_amd/_variants/libsigtk_gpu_r04.so SIGTK_AMD_LIB_ANY=1
for t in 1 0; do echo -n "r04 tail $t: "; SGK_EVENT_TAIL=$t bench; done
echo -n "r04 rna: "; bench --rna 1
unset SIGTK_AMD_LIB SIGTK_AMD_LIB_ANY
for t in 1 0; do echo -n "new tail $t: "; SGK_EVENT_TAIL=$t bench; done
echo -n "new rna: "; bench --rna 1
