# ablations of the streamed bf3 reservoir kernel (tools/variants/sbf3abl<bits>), ms per 64 steps of the C5 layer
for v in ${ABLS:-0 1 2 32 64 33 97 99}; do
  if [ $v = 0 ]; then lib=sgp_amd/csrc/libsgp_amd.so; else lib=tools/variants/sbf3abl$v/libsgp_amd.so; fi
  echo "abl $v: $(SGP_AMD_LIB=$PWD/$lib SGP_TUNE=res_bf3=1 python tools/probe_res_bf3.py child 100000 64 128 256 2>&1 | grep ms | cut -c1-60)"
done
