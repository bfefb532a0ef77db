#!/bin/bash
# usage: tools/run_prof.sh <tag> <what> <T>   -- runs on the GPU box (inside gpurun)
set -u
TAG=$1; WHAT=$2; T=$3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python $OLDPWD/tools/prof_kernels.py $WHAT $T > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc1 -o p -- python $OLDPWD/tools/prof_kernels.py $WHAT $T > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- python $OLDPWD/tools/prof_kernels.py $WHAT $T > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc3 -o p -- python $OLDPWD/tools/prof_kernels.py $WHAT $T > $OUT/pmc3.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc4 -o p -- python $OLDPWD/tools/prof_kernels.py $WHAT $T > $OUT/pmc4.log 2>&1
cd $OLDPWD
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
# keep only small files
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
