#!/bin/bash
# usage: tools/run_prof_sq.sh <tag> <T>  -- SQ stall counters of the SpMM kernel (SGP_FORCE selects it)
set -u
TAG=$1; T=$2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ROOTD=$PWD
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc1 -o p -- python $ROOTD/tools/prof_kernels.py spmm $T > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD -d $OUT/pmc2 -o p -- python $ROOTD/tools/prof_kernels.py spmm $T > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/pmc3 -o p -- python $ROOTD/tools/prof_kernels.py spmm $T > $OUT/pmc3.log 2>&1
cd $ROOTD
python tools/summarize_prof.py $OUT 2>&1 | grep -A12 "spmm_" > $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
