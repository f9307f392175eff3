set -e
bash tools/profile_round.sh lusgs --workload lusgs > gpurun_out/final_lusgs.log 2>&1
bash tools/profile_round.sh rk4 --workload rk4 > gpurun_out/final_rk4.log 2>&1
R=$(pwd); out=$R/gpurun_out/prof_dplur8; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $R/bench.py --workload dplur8 --no-cpu-baseline --steps 6 --warmup 2 > $out/bench_under_rocprof.json 2> $out/trace.log
cp $out/trace/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || true
cd $R
echo profiled
python bench.py --steps 20 --warmup 4 > gpurun_out/final_bench_lusgs.json 2> gpurun_out/final_bench_lusgs.err
python bench.py --workload rk4 --steps 40 --warmup 8 > gpurun_out/final_bench_rk4.json 2> gpurun_out/final_bench_rk4.err
python bench.py --workload dplur8 --steps 6 --warmup 2 > gpurun_out/final_bench_dplur8.json 2> gpurun_out/final_bench_dplur8.err
echo benched
