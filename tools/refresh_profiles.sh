#!/bin/bash
# Runs on the GPU box (gpurun -- tools/refresh_profiles.sh <tag>): bench lines, rocprofv3 kernel stats and the PMC
# passes the numbers in DESIGN.md / profiles/ come from.  Everything lands in gpurun_out/<tag>/; copy what is to be
# kept into profiles/ (tools/pmc_summary.py writes profiles/pmc_traffic.json).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r05_z}
mkdir -p $O
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout -s KILL 600 python bench.py --rna 1 --cpu-reads 0 > $O/bench_rna.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --ragged 0.8 --cpu-reads 0 > $O/bench_ragged.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --read-len 5000 --reads 200000 --cpu-reads 0 > $O/bench_5k.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --read-len 5000 --reads 200000 --rna 1 --cpu-reads 0 > $O/bench_5k_rna.json 2>> $O/bench.err
# round 3: the shapes the one-wavefront-per-read layout served badly (long reads in segments, short reads packed), and
# the same build with packing off
timeout -s KILL 600 python bench.py --ragged 0.8 --rna 1 --cpu-reads 0 > $O/bench_ragged_rna.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --read-len 12000 --reads 83333 --rna 1 --cpu-reads 0 > $O/bench_12k_rna.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --read-len 30000 --reads 33333 --rna 1 --cpu-reads 0 > $O/bench_30k_rna.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --ragged 0.8 --read-len 20000 --reads 50000 --rna 1 --cpu-reads 0 > $O/bench_ragged_20k_rna.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --ragged 0.8 --read-len 5000 --reads 200000 --cpu-reads 0 > $O/bench_ragged_5k.json 2>> $O/bench.err
timeout -s KILL 600 python bench.py --ragged 0.8 --read-len 5000 --reads 200000 --rna 1 --cpu-reads 0 > $O/bench_ragged_5k_rna.json 2>> $O/bench.err
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --cpu-reads 0 > $O/prof.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-reads 0 > $O/pmc_fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-reads 0 > $O/pmc_write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --cpu-reads 0 > $O/pmc_sq.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_rna -- python3 bench.py --steps 2 --warmup 1 --cpu-reads 0 --rna 1 > $O/pmc_sq_rna.log 2>&1
# round 4: the packed short reads (k_event_multi) and the chained segments (ragged batch: k_event's first workgroups)
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_5k -- python3 bench.py --steps 2 --warmup 1 --cpu-reads 0 --read-len 5000 --reads 200000 > $O/pmc_sq_5k.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_ragged -- python3 bench.py --steps 2 --warmup 1 --cpu-reads 0 --ragged 0.8 > $O/pmc_sq_ragged.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged -- python3 bench.py --steps 5 --warmup 2 --cpu-reads 0 --ragged 0.8 > $O/prof_ragged.log 2>&1
# small and odd batch sizes
for n in 1000 4000 9300; do
timeout -s KILL 600 python bench.py --reads $n --cpu-reads 0 > $O/bench_${n}.json 2>> $O/bench.err
done
if [ "$2" != "nosub" ]; then
timeout -s KILL 900 python bench.py --config 3 > $O/bench_c3.json 2>> $O/bench.err
timeout -s KILL 900 python bench.py --config 4 --steps 5 > $O/bench_c4.json 2>> $O/bench.err
timeout -s KILL 900 python bench.py --config 5 --steps 3 > $O/bench_c5.json 2>> $O/bench.err
timeout -s KILL 600 python tools/bench_subtools.py --reads 125000 --rna 0 > $O/subtools_c4.json 2>> $O/bench.err
timeout -s KILL 600 python tools/bench_subtools.py --reads 50000 --rna 1 > $O/subtools_c3.json 2>> $O/bench.err
timeout -s KILL 600 python tools/bench_subtools.py --ragged 0.8 > $O/subtools_ragged.json 2>> $O/bench.err
timeout -s KILL 900 python -m pytest tests/test_gpu_device_api.py -q -k config3 > $O/test_config3.txt 2>&1; tail -3 $O/test_config3.txt
# the subtool kernels at 125 000 x 100 000: kernel stats, HBM traffic, SQ counters
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sub -- python3 tools/bench_subtools.py --reads 125000 --rna 0 --steps 3 > $O/prof_sub.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_sub_fetch -- python3 tools/bench_subtools.py --reads 125000 --rna 0 --steps 1 > $O/pmc_sub_fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_sub_write -- python3 tools/bench_subtools.py --reads 125000 --rna 0 --steps 1 > $O/pmc_sub_write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/pmc_sub_sq -- python3 tools/bench_subtools.py --reads 125000 --rna 0 --steps 1 > $O/pmc_sub_sq.log 2>&1
python tools/pmc_kernels.py $O/pmc_sub_fetch > $O/pmc_sub_fetch.json 2>> $O/bench.err
python tools/pmc_kernels.py $O/pmc_sub_write > $O/pmc_sub_write.json 2>> $O/bench.err
python tools/pmc_kernels.py $O/pmc_sub_sq "" 12500000000 > $O/pmc_sub_sq.json 2>> $O/bench.err
fi
find $O -name "*.csv" -size +20M -delete
ls -la $O
tail -1 $O/bench.json | cut -c1-400
