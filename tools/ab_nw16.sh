# A/B of the split hop with 8 waves x 32 rows (default build) and 16 waves x 16 rows (tools/variants/nw16), same lease
for r in ${REPS:-1 2 3}; do
  for lib in ${LIBS:-sgp_amd/csrc/libsgp_amd.so tools/variants/nw16/libsgp_amd.so}; do
    echo "$lib: $(SGP_AMD_LIB=$PWD/$lib T=512 python tools/probe_split_abl.py child 2>&1 | grep ms)"
  done
done
