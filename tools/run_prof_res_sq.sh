#!/bin/bash
# usage: tools/run_prof_res_sq.sh <tag> <res|res256> <T>  -- kernel stats, SQ and fabric counters of the reservoir kernel
set -u
TAG=$1; WHAT=$2; T=$3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ROOTD=$PWD
cd /tmp
CMD="python $ROOTD/tools/prof_kernels.py $WHAT $T"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
cd $ROOTD
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
