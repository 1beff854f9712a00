cd $GRAFT_REPO_ROOT
(for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3 | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/clk_elim.txt &
timeout 120 python scripts/elimination_manhattan.py --passes 200 --structures 1 2>&1 | grep -E "per pass" 
wait
(for i in $(seq 1 20); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1 | tr '\n' ' '; echo; sleep 0.25; done) > gpurun_out/clk_bench.txt &
timeout 120 python bench.py --steps 200000 --warmup 2000 --no-cpu-baseline --no-modes 2>&1 | tail -1 | cut -c1-200
wait
echo ELIM; sort gpurun_out/clk_elim.txt | uniq -c | sort -rn | head -8; echo BENCH; sort gpurun_out/clk_bench.txt | uniq -c | sort -rn | head -5
