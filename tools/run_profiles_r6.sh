#!/bin/bash
# Round-6 evidence: one bench line + rocprofv3 kernel stats (+ HBM counter passes) per workload, SQ counters of the
# split-fp16 hop.  Every profiler invocation sits under `timeout` (a rocprofv3 run without it once ate a 15-minute lease).
# usage (on the GPU box, from the repo root): bash tools/run_profiles_r6.sh [workloads...]
set -u
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/r6
mkdir -p $OUT
export TMPDIR=/tmp
for w in ${@:-target}; do
  steps=3; [ $w = c5 ] && steps=1
  timeout 600 python bench.py --workload $w --steps $steps --warmup 1 > $OUT/bench_${w}_line.json 2> $OUT/bench_$w.err
  P=$ROOTD/gpurun_out/prof_r6_$w
  mkdir -p $P
  ( cd /tmp
    timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $P/trace -o t -- python $ROOTD/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $P/trace.log 2>&1
    timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d $P/pmc_fetch -o p -- python $ROOTD/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-exact-line > $P/pmc_fetch.log 2>&1
    timeout 400 rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_MISS_sum -d $P/pmc_write -o p -- python $ROOTD/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-exact-line > $P/pmc_write.log 2>&1 )
  python tools/summarize_prof.py $P > $P/summary.txt 2>&1
  cp $P/summary.txt $OUT/${w}_summary.txt 2>/dev/null
  cp $P/trace/*kernel_stats.csv $OUT/${w}_kernel_stats.csv 2>/dev/null || find $P/trace -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_kernel_stats.csv \;
  find $P -name "*.csv" -size +1M -delete; find $P -name "*.db" -delete
  echo "== $w"; head -c 600 $OUT/bench_${w}_line.json; echo
done
if [ "${SQ:-0}" = 1 ]; then
  SGP_FORCE=split timeout 900 bash tools/run_prof_sq.sh r6_split_sq 128
  cp gpurun_out/prof_r6_split_sq/summary.txt $OUT/spmm_split_T128_sq_summary.txt 2>/dev/null
fi
