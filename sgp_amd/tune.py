"""The ONE debug / tuning hook of the package: the environment variable ``SGP_TUNE``, a comma-separated list
of ``key=value`` pairs, read by the Python layer (here) and by ``libsgp_amd.so`` (``sgp::tune`` in
``csrc/api.hip``).  Nothing in it changes results beyond the documented tolerance; production use needs none
of it.  Keys:

Python layer
    hop=split|exact         hop kernel family on one GPU: split-fp16 (default) or the exact-fp32 kernels
    exact=mix|res           first choice among the exact-fp32 matrix-core kernels (default mix)
    exact_plans=lazy|eager  where the split-fp16 hop is the default: build the exact kernels' tile / mix plans only after an
                            admission flag came back 0 (until then the generic CSR kernel sits behind the predicate), or up front
    split_guard=1|0, split_passes=1|0   admission statistics of the split-fp16 hop; long-row operators in passes
    time_parallel=1|0, time_parallel_warm=<steps>, time_parallel_tol=1e-6   small graphs, contractive reservoirs: time pieces computed side by side from a
                            warm-up, accepted by a device-side comparison at every splice (0: one sequential chain)
    mix_thr=4               a column goes through the dense 16x16x4 part when >= thr of a block's 4 groups use it
    mix_min_share=0.25      least share of (group, column) pairs in the dense part for the mixed kernel to be chosen
    colblock=1|0            column-blocked hop for graphs without locality (0: generic CSR kernel)
    colblock_l2_mb=2.5      L2 budget of one column block
    overlap_chunks=16       small graphs: time pieces of the reservoir-over-hops pipeline (SGPEncoder.encode_device)
    overlap_cu_mask=1|0     small graphs: the reservoir chain and the hops beside it run on disjoint compute units (0: ordinary streams)
    equal_cost_tiles=0|1, equal_cost_q=0.15   experiment: tiles cut for equal cost

Library (integers)
    res_bf3=1               reservoirs with R = 32 / 64 (F = 16 / 32 / 64) and R = 256 (F = 32 / 64 / 128): products from three bf16
                            pieces per operand on the 16-bit matrix cores (csrc/reservoir_bf3.h; fp32-grade, no bound on the
                            operands); 0 = exact-fp32 MFMAs
    res_h16=1               bf16-piece reservoirs, tanh: recurrent products from two fp16 pieces of the bounded state and of W_hh's
                            scaled rows (csrc/reservoir_splitj_bf3.h, reservoir_bf3.h); 0 = three bf16 pieces there too
    res_pair=1              large-N bf16-piece reservoir, 5-6 tiles per SIMD: a wave multiplies its two tiles against every fragment
                            read (0: one tile after the other)
    res_tail_beside=1       the split-J tail of a large layer runs on a side lane beside the main part (0: after it)
    spmm_chunk=32           time steps per workgroup of the staged exact kernels
    split_time_major=0|1, split_tc=16   split-fp16 hop: an XCD walks its own TIME chunks over all tiles (default: on where a step's
                            source rows fit an L2, i.e. n_cols x feat x 4 <= 4 MB) instead of its own tiles over all time
    spmm_variant=1          inner-loop variant of sgp_spmm_tiled_f32
    mix_mode=6, res_cfg=0   launch shapes of sgp_spmm_mix_f32 / sgp_spmm_res_f32
    abl=0                   ablation bits of res / mix (only in builds with -DSGP_ABLATION)
    split_abl=0             ablation bits of sgp_spmm_split_f32, -DSGP_ABLATION builds only (1 no loads, 2 no MFMAs, 4 no stores, 8 no
                            conversion, 16 shared rows per XCD, 32 unpaired stores, 64 always step 0,
                            128 the two waves of a SIMD take the phases in opposite order, 256 per-wave timeline of workgroup 0)
    gesn_persistent=1, gesn_dbg=0, stack_debug=0, res_splitj_max, res_tail=1, res_stream8=1   reservoir / DynGESN experiments
"""
import os


def _table():
    out = {}
    for item in os.environ.get("SGP_TUNE", "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def get(key, default=None, cast=str):
    v = _table().get(key)
    if v is None or v == "":
        return default
    try:
        return cast(v)
    except ValueError:
        raise ValueError(f"SGP_TUNE: {key}={v!r} is not a valid {cast.__name__}") from None
