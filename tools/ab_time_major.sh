#!/bin/bash
# A/B of the split hop's time-major XCD mapping (SGP_TUNE=split_time_major=1[,split_tc=..]) on the small operators
# usage: tools/ab_time_major.sh <out.log> [workloads...]
OUT=$1; shift
WLS=${@:-"c4 c3 c4full"}
: > $OUT
for wl in $WLS; do
  for t in "" "split_time_major=1" "split_time_major=1,split_tc=16" "split_time_major=1,split_tc=8"; do
    echo "== $wl SGP_TUNE=$t" >> $OUT
    SGP_TUNE=$t timeout 600 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-exact-line 2>/dev/null \
      | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['ms_per_launch'], r['roofline']['frac'], r.get('verified'))" >> $OUT 2>&1
  done
done
cat $OUT
