# ablation timings of spmm_pipe (build with `make EXTRA=-DSGP_ABLATION`): SGP_PIPE_ABL bit 0 = no
# staging DMA, bit 1 = no quad loops, bit 2 = staging always reads the chunk's first step (L2 hits)
for c in ${CHUNKS:-32}; do for a in ${ABLS:-0 1 2 3 6}; do echo "CHUNK $c ABL $a"; SGP_SPMM_CHUNK=$c SGP_PIPE_ABL=$a SGP_PROBE=pipe timeout 300 python tools/probe_kernels.py spmm 2>&1 | grep "spmm pipe"; done; done
