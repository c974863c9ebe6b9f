#!/bin/bash
# Window-attention measurement on the GPU box: same-process A/B timing of the SVR2_ATTN_V variants, then rocprofv3 PMC
# passes per variant (counters only with --kernel-trace; FETCH_SIZE / WRITE_SIZE in passes of their own).
#   usage: tools/profile_attn.sh <tag> "<versions: 5 = default kernel, 1 = the plain kernel of the same geometry>"
set -e -o pipefail
TAG=${1:-r03}; VERS=${2:-1,5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/attn_$TAG
mkdir -p $OUT
cd $ROOT
SVR2_KB_ATTN_VERS=$VERS python3 tools/kernel_bench.py attn4k > $OUT/ab_timing.log 2>&1
cat $OUT/ab_timing.log
cd /tmp && export TMPDIR=/tmp
export SVR2_KB_ITERS=3 SVR2_KB_WARM=1
for v in ${VERS//,/ }; do
  export SVR2_KB_ATTN_VERS=$v
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/v${v}_mfma -o p -- python3 $ROOT/tools/kernel_bench.py attn4k > $OUT/v${v}_mfma.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/v${v}_wait -o p -- python3 $ROOT/tools/kernel_bench.py attn4k > $OUT/v${v}_wait.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/v${v}_fetch -o p -- python3 $ROOT/tools/kernel_bench.py attn4k > $OUT/v${v}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/v${v}_write -o p -- python3 $ROOT/tools/kernel_bench.py attn4k > $OUT/v${v}_write.log 2>&1
  echo "=== SVR2_ATTN_V=$v" >> $OUT/pmc_summary.txt
  python3 $ROOT/tools/pmc_summary.py $OUT/v${v}_mfma $OUT/v${v}_wait $OUT/v${v}_fetch $OUT/v${v}_write --match flash_attn --split regular,shifted >> $OUT/pmc_summary.txt 2>&1
done
cat $OUT/pmc_summary.txt
# keep the merged output small: drop the raw per-dispatch csv files of everything but the attention kernel
find $OUT -name "*agent_info.csv" -delete
