#!/bin/bash
# Round-4 evidence: one bench line + rocprofv3 kernel stats (+ HBM counter passes) per workload, SQ
# counters and the per-wave timeline of the split-fp16 hop kernel.
# usage (on the GPU box, from the repo root): bash tools/run_profiles_r4.sh [workloads...]
set -u
ROOTD=$PWD
OUT=$ROOTD/gpurun_out/r4
mkdir -p $OUT
for w in ${@:-target}; do
  steps=3; [ $w = c5 ] && steps=1
  timeout 600 python bench.py --workload $w --steps $steps --warmup 1 > $OUT/bench_${w}_line.json 2> $OUT/bench_$w.err
  timeout 900 bash tools/run_prof_cmd.sh r4_$w python $ROOTD/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline
  cp gpurun_out/prof_r4_$w/summary.txt $OUT/${w}_summary.txt 2>/dev/null
  cp gpurun_out/prof_r4_$w/trace/*kernel_stats.csv $OUT/${w}_kernel_stats.csv 2>/dev/null
  echo "== $w"; cat $OUT/bench_${w}_line.json | head -c 700; echo
done
# SQ counters of the split-fp16 hop (T = 128) and its per-wave timeline
if [ "${SQ:-0}" = 1 ]; then
  SGP_FORCE=split bash tools/run_prof_sq.sh r4_split_sq 128
  cp gpurun_out/prof_r4_split_sq/summary.txt $OUT/spmm_split_T128_sq_summary.txt 2>/dev/null
fi
# per-wave timeline of the split-fp16 hop (ablation build: tools/build_variant.sh abl -DSGP_ABLATION)
if [ "${TIMELINE:-0}" = 1 ]; then
  SGP_AMD_LIB=$ROOTD/tools/variants/abl/libsgp_amd.so T=64 timeout 300 python tools/probe_split_abl.py 256 2>&1 | grep -A64 "spmm_split timeline" | head -70 > $OUT/spmm_split_timeline.txt
fi
# SQ counters of the bf16-piece reservoir kernels (narrow: target layer; wide: C5 layer)
if [ "${RES_SQ:-0}" = 1 ]; then
  bash tools/run_prof_res_sq.sh r4_bf3 res 256; cp gpurun_out/prof_r4_bf3/summary.txt $OUT/res_bf3_sq_summary.txt 2>/dev/null
  bash tools/run_prof_res_sq.sh r4_sbf3 res256 64; cp gpurun_out/prof_r4_sbf3/summary.txt $OUT/res_stream_bf3_sq_summary.txt 2>/dev/null
fi
