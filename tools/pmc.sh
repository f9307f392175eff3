#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# Separate rocprofv3 --pmc passes (never combined with tracing) over bench.py;
# per-kernel sums are written to gpurun_out/pmc_<tag>.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $line -d $R/gpurun_out/pmc_$tag/p$i -o p --output-format csv -- \
    python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $R/gpurun_out/pmc_$tag.p$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
LIST
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$tag > $R/gpurun_out/pmc_$tag.json
cat $R/gpurun_out/pmc_$tag.json
