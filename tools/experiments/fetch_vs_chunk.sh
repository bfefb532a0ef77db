#!/bin/bash
# FETCH_SIZE (fabric reads) of the SpMM kernel as a function of the time-chunk length
export TMPDIR=/tmp
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/fetch_chunk
mkdir -p $OUT
cd /tmp
for c in ${CHUNKS:-4 8 16 32 64}; do
  SGP_TUNE=spmm_chunk=$c SGP_FORCE=res timeout 150 rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/c$c -o p -- python $ROOTD/tools/prof_kernels.py spmm ${T:-256} > $OUT/c$c.log 2>&1
  echo "chunk $c"; python $ROOTD/tools/summarize_prof.py $OUT/c$c | grep -A3 "spmm_pipe"
done
