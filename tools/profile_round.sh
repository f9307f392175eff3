#!/bin/bash
# usage (on the GPU box, repo root): tools/profile_round.sh <tag> [bench args]
# 1. rocprofv3 --kernel-trace --stats of the bench command (STEPS / WARMUP: its --steps / --warmup)
# 2. separate --pmc passes for HBM traffic (FETCH_SIZE, WRITE_SIZE)
# Everything lands in gpurun_out/prof_<tag>/ ; copy the summaries to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- \
  python3 $R/bench.py --no-cpu-baseline --no-extra --steps ${STEPS:-20} --warmup ${WARMUP:-4} "$@" > $out/bench_under_rocprof.json 2> $out/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c -d $out/pmc_$c -o p --output-format csv -- \
    python3 $R/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 "$@" > $out/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_summary.py $out > $out/pmc.json
cp $out/trace/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
ls $out
