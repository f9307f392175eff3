#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# Separate rocprofv3 --pmc passes (never combined with tracing) over bench.py;
# per-kernel sums are written to gpurun_out/pmc_<tag>.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $line -d $R/gpurun_out/pmc_$tag/p$i -o p --output-format csv -- \
    python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $R/gpurun_out/pmc_$tag.p$i.log 2>&1
done < ${PMC_LIST:-$R/tools/pmc_full.txt}
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$tag > $R/gpurun_out/pmc_$tag.json
cat $R/gpurun_out/pmc_$tag.json
