#!/bin/bash
# rocprofv3 kernel stats of the DynGESN encoder on the METR-LA shape: tools/time_gesn.py runs the
# persistent path (gesn_persistent<6>, one launch per 256 steps) and then the stepwise path
# (gemm_nt_kernel + gesn_update_kernel per step and layer) in one process.
set -u
export TMPDIR=/tmp
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/prof_gesn
mkdir -p $OUT
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python $ROOTD/tools/time_gesn.py 2000 > $OUT/trace.log 2>&1
cd $ROOTD
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/gesn_metrla_kernel_stats.csv \;
grep "gesn" $OUT/trace.log
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
