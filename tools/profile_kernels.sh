#!/bin/bash
# rocprofv3 PMC passes (counters with --kernel-trace only; FETCH_SIZE / WRITE_SIZE in passes of their own) of one
# tools/kernel_bench.py mode, folded per kernel by tools/pmc_summary.py.
#   usage: tools/profile_kernels.sh <kernel_bench mode> <tag> [kernel-name substring]
set -e -o pipefail
MODE=${1:-halopmc}; TAG=${2:-r02}; MATCH=${3:-kernel}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_${MODE}_$TAG
mkdir -p $OUT
cd $ROOT
python3 tools/kernel_bench.py $MODE > $OUT/timing.log 2>&1
cat $OUT/timing.log
cd /tmp && export TMPDIR=/tmp
export SVR2_KB_ITERS=3 SVR2_KB_WARM=1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o p -- python3 $ROOT/tools/kernel_bench.py $MODE > $OUT/mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/wait -o p -- python3 $ROOT/tools/kernel_bench.py $MODE > $OUT/wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $ROOT/tools/kernel_bench.py $MODE > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python3 $ROOT/tools/kernel_bench.py $MODE > $OUT/write.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/mfma $OUT/wait $OUT/fetch $OUT/write --match "$MATCH" | cut -c1-160 > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
