#!/bin/bash
# Equal-cost tiles (SGP_TUNE=equal_cost_tiles=1) vs uniform tiles for spmm_res: time and fabric reads.
export TMPDIR=/tmp SGP_PROBE_CHECK=${CHECK:-1} SGP_PROBE_T=${T:-512} SGP_PROBE_KERNELS=res
ROOTD=$PWD
for eq in 0 1; do
  for q in ${QS:-0.15}; do
  export SGP_TUNE=equal_cost_tiles=$eq,equal_cost_q=$q
  echo "equal-cost $eq q=$q: $(timeout 300 python $ROOTD/tools/probe_blk.py 2>&1 | grep -E 'cfg=|csr')"
  if [ "${PMC:-1}" = 1 ]; then
  (cd /tmp && SGP_PROBE_CHECK=0 SGP_PROBE_T=256 timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/eq_${eq}_$q -o p -- python $ROOTD/tools/probe_blk.py > /tmp/eq_${eq}_$q.log 2>&1)
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob('/tmp/eq_${eq}_$q/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmm_' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print('   T=256 per launch:', {k: '%.4g' % (acc[k]/max(n[k],1)) for k in sorted(acc)}, ' (slab 6.4e6 KB; FETCH_SIZE x2 on gfx950)')
PY
  fi
  [ $eq = 0 ] && break
  done
done
