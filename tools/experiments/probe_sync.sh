export SGP_PROBE_KERNELS=res SGP_PROBE_CHECK=0
for eq in ${EQS:-1}; do for w in ${WS:-0 1000 31 8}; do
echo "equal-cost=$eq drift=$w: $(SGP_EQUAL_COST_TILES=$eq SGP_SPMM_SYNC=$w timeout 200 python tools/probe_blk.py 2>&1 | grep -E 'cfg=' | tr '\n' ' ')"
done; done
