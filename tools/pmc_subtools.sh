# SQ / GRBM counters of the lane-per-read subtool kernels (tools/bench_subtools.py under rocprofv3 --pmc)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_sub
mkdir -p $O
timeout -s KILL 600 rocprofv3 --pmc ${1:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD} --kernel-trace --output-format csv -d $O/${2:-sq} -- python3 tools/bench_subtools.py --reads ${3:-125000} --rna 0 --steps 1 > $O/${2:-sq}.log 2>&1
tail -3 $O/${2:-sq}.log | cut -c1-200
find $O -name "*.csv" -size +20M -delete
