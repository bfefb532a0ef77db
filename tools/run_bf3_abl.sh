# ablations of the bf3 reservoir kernel (tools/variants/bf3abl<bits>), ms per 256 steps of the target layer
for v in ${ABLS:-0 1 2 4 8 16 24 7}; do
  if [ $v = 0 ]; then lib=sgp_amd/csrc/libsgp_amd.so; else lib=tools/variants/bf3abl$v/libsgp_amd.so; fi
  echo "abl $v: $(SGP_AMD_LIB=$PWD/$lib SGP_TUNE=res_bf3=1 python tools/probe_res_bf3.py child 100000 256 64 64 2>&1 | grep ms | cut -c1-60)"
done
