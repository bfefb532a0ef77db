#!/bin/bash
# usage: tools/run_prof_cmd.sh <tag> <command...>   -- rocprofv3 kernel stats + HBM counter passes
set -u
TAG=$1; shift
export TMPDIR=/tmp
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_fetch -o p -- "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/pmc_write -o p -- "$@" > $OUT/pmc_write.log 2>&1
cd $ROOTD
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
