#!/bin/bash
# staging-volume ablations of both hop kernels: a third fewer (512) / half (1024) of the staging pieces, none (1).
# Needs tools/build_variant.sh abl -DSGP_ABLATION.  Results are WRONG by construction; the times are the point.
export SGP_AMD_LIB=$PWD/tools/variants/abl/libsgp_amd.so
for k in mix res; do for v in 0 512 1024 1; do
  echo "$k ABL $v"
  SGP_TUNE=abl=$v timeout 200 python tools/probe_mix.py 100000 512 5 $k 2>&1 | grep "^$k"
done; done
