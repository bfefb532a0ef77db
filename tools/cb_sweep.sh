#!/bin/bash
# column-blocked hop: time per launch against the L2 budget of a block (SGP_COLBLOCK_L2_MB), random workload
for mb in 1.0 1.5 2.0 3.2; do
  SGP_COLBLOCK_L2_MB=$mb timeout 600 python bench.py --workload random --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/cb_$mb.json
  python -c "
import json; d=json.load(open('gpurun_out/cb_$mb.json')); print('$mb', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['ms_per_launch'])"
done
