cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01e
mkdir -p $O
timeout -s KILL 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout -s KILL 300 python bench.py --rna 1 --cpu-reads 0 > $O/bench_rna.json 2>> $O/bench.err
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --cpu-reads 0 > $O/prof.log 2>&1
find $O -name "*kernel_trace.csv" -size +5M -delete
cat $O/prof/*/*kernel_stats.csv | head -5
tail -1 $O/bench.json | cut -c1-200
tail -1 $O/bench_rna.json | cut -c1-200
