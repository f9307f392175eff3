#!/bin/bash
# scratch: rans kernel timings of the default library and of variants in build/
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/rans_probe; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for v in default "$@"; do
  if [ $v = default ]; then unset AGX_RANS_LIB; else export AGX_RANS_LIB=$R/build/$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/$v -o t --output-format csv -- \
    python3 $R/bench.py --workload rans4 --steps 6 --warmup 2 --no-cpu-baseline > $out/$v.json 2> $out/$v.err || exit 1
  echo "== $v"; cut -c1-160 $out/$v.json | grep -o '"ms_per_step": [0-9.]*'
  head -12 $out/$v/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
done
