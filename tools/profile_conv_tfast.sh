#!/bin/bash
# FETCH_SIZE of the three dominant halo-conv shapes under both tile orders (SVR2_CONV_TFAST=0/1), folded per kernel.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tfast
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SVR2_KB_ITERS=3 SVR2_KB_WARM=1
for v in 0 1; do
  export SVR2_CONV_TFAST=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch$v -o p -- python3 $ROOT/tools/kernel_bench.py halopmc > $OUT/fetch$v.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/mfma$v -o p -- python3 $ROOT/tools/kernel_bench.py halopmc > $OUT/mfma$v.log 2>&1
  echo "== SVR2_CONV_TFAST=$v" >> $OUT/pmc_summary.txt
  python3 $ROOT/tools/pmc_summary.py $OUT/mfma$v $OUT/fetch$v --match conv_halo | cut -c1-200 >> $OUT/pmc_summary.txt
done
cat $OUT/pmc_summary.txt
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
