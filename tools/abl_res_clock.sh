#!/bin/bash
# Is the staging tax of spmm_res cycles or clock?  GPU cycles (GRBM_GUI_ACTIVE, summed over the 8 XCDs) and
# kernel time of the full kernel (ABL 0) and the no-staging build (ABL 1), T = 128.  Needs the abl variant.
set -u
export TMPDIR=/tmp SGP_AMD_LIB=$PWD/tools/variants/abl/libsgp_amd.so SGP_FORCE=res
ROOTD=$PWD
for v in 0 1; do
  OUT=$ROOTD/gpurun_out/prof_abl_clock_$v
  mkdir -p $OUT
  (cd /tmp && SGP_TUNE=abl=$v rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python $ROOTD/tools/prof_kernels.py spmm 128 > $OUT/trace.log 2>&1)
  (cd /tmp && SGP_TUNE=abl=$v rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/pmc -o p -- python $ROOTD/tools/prof_kernels.py spmm 128 > $OUT/pmc.log 2>&1)
  echo "== ABL $v"
  grep spmm_res $OUT/trace/*kernel_stats.csv | cut -c1-160
  python tools/summarize_prof.py $OUT 2>/dev/null | grep -A5 "kernel: void (anonymous namespace)::spmm_res" | head -8
  find $OUT -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
done
