#!/bin/bash
# spmm_res ablations behind the "wide tile" question: one barrier per step (256), a third fewer staging
# pieces (512), both (768), no staging at all (1), no staging + one barrier (257).  Needs
# tools/build_variant.sh abl -DSGP_ABLATION.  Results are WRONG by construction; the times are the point.
export SGP_AMD_LIB=$PWD/tools/variants/abl/libsgp_amd.so
for v in 0 256 512 768 1 257; do
  echo "ABL $v"
  SGP_TUNE=abl=$v timeout 200 python tools/probe_mix.py 100000 512 5 res 2>&1 | grep "^res"
done
