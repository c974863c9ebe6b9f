#!/bin/bash
# What do the waves of the three big MFMA kernels wait for?  Three rocprofv3 PMC passes (counters only, --kernel-trace) of
# tools/kernel_bench.py stalls, folded per kernel by tools/pmc_raw.py.
#   usage: tools/profile_stalls.sh <tag>
set -e -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stalls_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SVR2_KB_ITERS=2 SVR2_KB_WARM=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o p -- python3 $ROOT/tools/kernel_bench.py stalls > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL --kernel-trace --output-format csv -d $OUT/b -o p -- python3 $ROOT/tools/kernel_bench.py stalls > $OUT/b.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/c -o p -- python3 $ROOT/tools/kernel_bench.py stalls > $OUT/c.log 2>&1
for k in conv_halo_wr gemm4 attn_d512; do
  for d in a b c; do python3 $ROOT/tools/pmc_raw.py $OUT/$d $k; done
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
