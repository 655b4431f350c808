#!/bin/bash
# Runs on the GPU box (gpurun -- tools/pmc_wave.sh <tag>): SQ counters of the wave-per-read kernels of stat / jnn /
# prefix on 20 000 x 100 000 samples, DNA- and RNA-headed reads; summaries in gpurun_out/<tag>/wave_pmc_{dna,rna}.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r02}
mkdir -p $O
for k in 0 1; do
  n=$([ $k = 0 ] && echo dna || echo rna)
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_wave_$n -- python3 tools/bench_subtools.py --reads 20000 --rna $k --steps 1 > $O/pmc_wave_$n.log 2>&1
  python tools/pmc_kernels.py $O/pmc_wave_$n wave 2e9 > $O/wave_pmc_$n.json
  find $O/pmc_wave_$n -name "*.csv" -size +5M -delete
done
cat $O/wave_pmc_dna.json | head -40
