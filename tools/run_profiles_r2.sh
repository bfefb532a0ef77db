#!/bin/bash
# Round-2 evidence: one bench line + rocprofv3 kernel stats (+ HBM counter passes) per workload.
# usage (on the GPU box, from the repo root): bash tools/run_profiles_r2.sh [workloads...]
set -u
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/r2
mkdir -p $OUT
for w in ${@:-target c1 c2 c3 c4 c5}; do
  steps=2; [ $w = c5 ] && steps=1
  timeout 600 python bench.py --workload $w --steps $steps --warmup 1 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  timeout 900 bash tools/run_prof_cmd.sh r2_$w python $ROOTD/bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline
  cp gpurun_out/prof_r2_$w/summary.txt $OUT/${w}_summary.txt 2>/dev/null
  cp gpurun_out/prof_r2_$w/trace/*kernel_stats.csv $OUT/${w}_kernel_stats.csv 2>/dev/null
  echo "== $w"; cat $OUT/bench_$w.json | head -c 600; echo
done
