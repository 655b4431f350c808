#!/bin/bash
# round 4 experiment a: config 2 with reads cut into segments (VERDICT r03 task 1)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a
mkdir -p $O
run() { # tag, env..., -- args
  tag=$1; shift
  env "$@" python bench.py --cpu-reads 0 --steps 10 $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$ARGS', d['ms_per_step'], d['roofline']['kernels_ms'], d['config']['fallback_reads'], d['config']['rerun_chunks'], d['config']['split_reads'])" | tee -a $O/log.txt
}
for ARGS in "" "--rna 1"; do
export ARGS
run base A=1
for seg in 12288 16384 20480 25600 33792 50176; do
run seg$seg SGK_EVENT_SEG=$seg SGK_EVENT_LONG_MIN=65536
done
done
ARGS="--reads 125000 --steps 3" run base125k A=1
