#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags]  -- builds sgp_amd/csrc into
# tools/variants/<name>/libsgp_amd.so (e.g. `abl -DSGP_ABLATION`); use with SGP_AMD_LIB=<that path>.
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/variants/$NAME
mkdir -p $OUT
cd $ROOT/sgp_amd/csrc
SRCS=$(ls *.hip)
pids=()
for s in $SRCS; do
  /opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -c $s -o $OUT/${s%.hip}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsgp_amd.so $OUT/*.o
rm -f $OUT/*.o
echo built $OUT/libsgp_amd.so
