#!/bin/bash
# Fabric reads and time of spmm_blk against the time-chunk length (T = 256, slab 6.4e6 KB).
export TMPDIR=/tmp SGP_PROBE_CHECK=0 SGP_PROBE_T=256 SGP_PROBE_CFGS=0 SGP_PROBE_KERNELS=${KERNELS:-blk}
ROOTD=$PWD
for c in ${CHUNKS:-4 8 16 32}; do
  export SGP_SPMM_CHUNK=$c
  echo "chunk $c: $(python $ROOTD/tools/probe_blk.py 2>&1 | grep 'cfg=')"
  (cd /tmp && rocprofv3 --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d /tmp/fvc_$c -o p -- python $ROOTD/tools/probe_blk.py > /tmp/fvc_$c.log 2>&1)
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob('/tmp/fvc_$c/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmm_' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(acc): print('   ',k, acc[k]/max(n[k],1))
PY
done
