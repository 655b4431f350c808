cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/clk
(timeout -s KILL 120 python tools/bench_subtools.py --reads 125000 --rna 0 --steps 40 > gpurun_out/clk/sub.json 2>/dev/null) &
BP=$!
sleep 20
for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.25; done > gpurun_out/clk/clk.txt
wait $BP
sort gpurun_out/clk/clk.txt | uniq -c
