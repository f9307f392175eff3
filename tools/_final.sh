set -e
python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { tail -20 gpurun_out/final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/final_gpu_tests.log
python bench.py > gpurun_out/final_bench_lusgs.json 2> gpurun_out/final_bench_lusgs.err
python bench.py --workload rk4 --steps 40 --warmup 8 > gpurun_out/final_bench_rk4.json 2> gpurun_out/final_bench_rk4.err
python bench.py --workload dplur8 --steps 6 --warmup 2 > gpurun_out/final_bench_dplur8.json 2> gpurun_out/final_bench_dplur8.err
echo benched
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --size 64 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/final_bench_2rank_gloo.json 2> gpurun_out/final_bench_2rank_gloo.err || { tail -5 gpurun_out/final_bench_2rank_gloo.err; exit 1; }
tail -c 600 gpurun_out/final_bench_2rank_gloo.json
