"""The precision contract of the split-fp16 hop, through the DEFAULT dispatch (``propagate`` with no ``force``, the
encoders) -- reference: ``x = adj @ x`` is plain fp32 for ANY x (lib/sgp_preprocessing.py:200-203; raw [B, N, F]
batches in lib/nn/models/sgp_model.py:169-181; relu reservoirs, reservoir.py:37-41).

The split kernel carries |x| >= 2^-16 B_c of a column with bound B_c to 2^-23 relative and everything smaller to
2^-38 B_c absolute (include/sgp_amd.h).  ``sgp_split_prepare_f32`` admits it only where that absolute term is below
2^-22 of the column's RMS and the exact-fp32 kernel runs otherwise -- a device-side choice.  Criterion of every
test here: PER FEATURE COLUMN, relative Frobenius error against the dense fp64 product <= 1e-5 (north_star) and
<= 4x the error of the reference's own arithmetic (a CPU fp32 sparse product) + 1e-7."""
import numpy as np
import pytest
import torch

import sgp_amd
from sgp_amd import graph, hip, synthetic
from sgp_amd.sgp_preprocessing import propagate_into

pytestmark = pytest.mark.gpu

N, K, T, D = 2600, 40, 6, 64


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.require_gpu()


@pytest.fixture(scope="module")
def op():
    ei, ew, _ = synthetic.knn_graph(N, K, seed=7)
    return graph.ShiftOperator.from_edges(ei, ew, N)


def products(op, x):
    xc = x.cpu()
    a64 = torch.sparse_csr_tensor(op.rowptr.long(), op.col.long(), op.val.double(), (op.num_nodes, op.num_cols))
    a32 = torch.sparse_csr_tensor(op.rowptr.long(), op.col.long(), op.val, (op.num_nodes, op.num_cols))
    ref64 = torch.stack([a64 @ xc[b].double() for b in range(xc.shape[0])])
    cpu32 = torch.stack([a32 @ xc[b] for b in range(xc.shape[0])])
    return ref64, cpu32


def col_err(y, ref64):
    """Relative Frobenius error of every feature column (zero columns: absolute)."""
    d = (y.double() - ref64).flatten(0, 1).norm(dim=0)
    n = ref64.flatten(0, 1).norm(dim=0)
    return torch.where(n > 0, d / n.clamp_min(1e-300), d)


def check_columns(y, ref64, cpu32):
    assert torch.isfinite(y).all()
    e, e_cpu = col_err(y.cpu(), ref64), col_err(cpu32, ref64)
    assert float(e.max()) <= 1e-5, (float(e.max()), int(e.argmax()))
    worst = (e - 4 * e_cpu).max()
    assert float(worst) <= 1e-7, (float(worst), e.tolist()[:8], e_cpu.tolist()[:8])
    return e, e_cpu


def default_hop(op, x, bound=None):
    xg = x.cuda()
    y = torch.full((x.shape[0], op.num_nodes, x.shape[2]), float("nan"), device="cuda")
    assert op.split_eligible(xg, y)
    op.propagate(xg, y, x_bound=bound)
    return y, op.resolved_kernel()


def test_columns_of_very_different_scale_in_one_slice(op):
    """Columns scaled 1e6 .. 1e-6 inside ONE 16-feature slice: every column has its own power-of-two scale, so the
    split kernel serves all of them at fp32 accuracy (a single scale per tensor lost the small columns)."""
    torch.manual_seed(0)
    x = torch.randn(T, N, D)
    scales = torch.tensor([1e6, 1e3, 1.0, 1e-1, 1e-2, 1e-3, 1e-6, 1e4] * (D // 8))
    x = x * scales
    y, kernel = default_hop(op, x)
    assert kernel == "spmm_split"
    check_columns(y, *products(op, x))


def test_one_huge_column_beside_unit_columns(op):
    torch.manual_seed(1)
    x = torch.randn(T, N, D)
    x[:, :, 5] *= 1e8
    y, kernel = default_hop(op, x)
    assert kernel == "spmm_split"
    check_columns(y, *products(op, x))


def test_one_outlier_entry(op):
    """ONE entry of 1e9 in a tensor of unit entries: only its own column sees the large bound; every other column
    keeps its accuracy, and the outlier's column is dominated by the outlier."""
    torch.manual_seed(2)
    x = torch.randn(T, N, D)
    x[3, 1234, 17] = 1e9
    y, _ = default_hop(op, x)
    e, e_cpu = check_columns(y, *products(op, x))
    others = torch.arange(D) != 17
    assert float(e[others].max()) <= 5e-7


def test_outliers_that_hide_a_column_send_the_hop_to_the_exact_kernel(op):
    """A column of entries ~1e-9 with a few entries of 1: the bound is 2^30 above what most of the column holds, the
    absolute error term would be 2^-8 of those values -- the device-side test fails and the exact kernel (enqueued
    behind the split kernel under the opposite predicate) computes the hop."""
    torch.manual_seed(3)
    x = torch.randn(T, N, D)
    x[:, :, 9] *= 1e-9
    x[:, ::1300, 9] = 1.0                                              # two nodes of 2600 hold 1, the rest ~1e-9
    x[:, :, 41] *= 1e-14
    x[0, 5, 41] = 100.0                                                # one entry of 100 in a column of ~1e-14
    y, kernel = default_hop(op, x)                                     # (whichever kernel the device picks)
    check_columns(y, *products(op, x))
    # a-priori bound far above the data (a tanh reservoir with a tiny input scaling): the bound is 2^20 x the RMS
    small = torch.randn(T, N, D) * 1e-6
    y, kernel = default_hop(op, small, bound=1.0)
    assert kernel != "spmm_split"
    check_columns(y, *products(op, small))
    long = torch.randn(70, N, 16) * 1e-6                                # (sampled statistics: every second step)
    y, kernel = default_hop(op, long, bound=1.0)
    assert kernel != "spmm_split"
    check_columns(y, *products(op, long))
    # the same operand with its bound measured is served by the split kernel
    y, kernel = default_hop(op, small)
    assert kernel == "spmm_split"
    check_columns(y, *products(op, small))


def test_data_beyond_the_callers_bound_and_non_finite_values(op):
    """A bound the data exceed (a leaking rate outside [0, 1], a state handed in from elsewhere) must not reach the
    fp16 range silently: the sampled statistics see it and the exact kernel runs; inf / NaN propagate like fp32."""
    torch.manual_seed(4)
    x = torch.randn(T, N, D) * 3.0
    y, kernel = default_hop(op, x, bound=1.0)
    assert kernel != "spmm_split"
    check_columns(y, *products(op, x))
    x = torch.randn(T, N, D)
    x[2, 100, 3] = float("inf")
    xg = x.cuda()
    yg = torch.full_like(xg, 7.0)
    op.propagate(xg, yg)
    assert op.resolved_kernel() != "spmm_split"
    # the exact kernel's own result on this operand (the mixed kernel multiplies explicit zeros of its dense blocks, so
    # a non-finite entry also reaches the other rows of its 16-row block as NaN: round-3 behaviour, outside the
    # reference's use; the generic CSR kernel shows exactly the reference's pattern)
    ref = torch.empty_like(xg)
    op.propagate(xg, ref, x_bound=float("inf"))
    assert torch.equal(torch.isfinite(yg), torch.isfinite(ref)) and not torch.isfinite(yg).all()
    fin = torch.isfinite(ref)
    assert torch.equal(yg[fin], ref[fin])
    csr = torch.empty_like(xg)
    op.propagate(xg, csr, force="csr")
    assert bool((~torch.isfinite(csr) <= ~torch.isfinite(yg)).all())     # wherever fp32 is non-finite, so are we
    assert torch.allclose(yg[fin], csr[fin], rtol=1e-5, atol=1e-5)


def test_zero_columns_and_zero_operand(op):
    """Identically zero columns (dead relu units) are exact and do not block the split kernel; a zero tensor too."""
    torch.manual_seed(5)
    x = torch.relu(torch.randn(T, N, D))
    x[:, :, 7] = 0.0
    x[:, :, 20:24] = 0.0
    y, kernel = default_hop(op, x)
    assert kernel == "spmm_split"
    assert float(y[:, :, 7].abs().max()) == 0.0 and float(y[:, :, 20:24].abs().max()) == 0.0
    check_columns(y, *products(op, x))
    y, _ = default_hop(op, torch.zeros(2, N, D))
    assert float(y.abs().max()) == 0.0
    empty = torch.empty(0, N, D, device="cuda")
    assert op.propagate(empty, torch.empty_like(empty)).shape == (0, N, D)        # an empty time chunk is not an error


def test_profile_kernels_against_numpy(op):
    """``sgp_col_stats_f32`` and ``sgp_split_prepare_f32`` themselves: statistics of strided samples, scales that put
    every bound in [2^13, 2^14], the next hop's bounds, the flag."""
    torch.manual_seed(6)
    x = (torch.randn(70, 500, 48) * torch.logspace(-3, 3, 48)).cuda()
    for stride, rs in ((1, 1), (2, 1), (9, 3), (1, 7)):
        st = hip.col_stats(x, stride, r_stride=rs).cpu()
        xs = x.cpu()[::stride, ::rs].flatten(0, 1)
        assert torch.equal(st[0].view(torch.int32), xs.abs().max(0).values.view(torch.int32))
        assert torch.allclose(st[1], (xs.double() ** 2).sum(0).float(), rtol=1e-4)
    prof = hip.split_profile(x, None, None, norm_inf=2.5)
    tab, flag, nxt = prof.tab.cpu(), int(prof.flag.item()), prof.bound_out.tensor.cpu()
    amax = x.cpu().flatten(0, 1).abs().max(0).values
    scaled = amax * tab[0]
    assert flag == 1 and (scaled > 2.0 ** 13 - 1).all() and (scaled <= 2.0 ** 14).all()
    assert torch.equal(tab[0] * tab[1], torch.ones(48)) and (np.frexp(tab[0].numpy())[0] == 0.5).all()
    assert torch.allclose(nxt, amax * 2.5, rtol=1e-5)
    # the previous hop's bounds as input, sampled statistics
    prof2 = hip.split_profile(x, None, hip.ColumnBound((amax * 1.5).cuda()), norm_inf=1.0)
    assert int(prof2.flag.item()) == 1
    assert (amax * 1.5 * prof2.tab.cpu()[0] <= 2.0 ** 14).all()
    # bounds 2^20 above the data: not admitted
    prof3 = hip.split_profile(x, None, hip.ColumnBound((amax * 2.0 ** 20).cuda()))
    assert int(prof3.flag.item()) == 0
    # a bound the data exceed: not admitted
    prof4 = hip.split_profile(x, None, hip.ColumnBound((amax * 0.5).cuda()))
    assert int(prof4.flag.item()) == 0


def test_raw_features_through_the_spatial_encoder_and_a_relu_encoder():
    """``SGPSpatialEncoder`` on raw un-normalised features (columns of mixed units) and a relu ``SGPEncoder``
    against the CPU oracle, column by column, through the default dispatch."""
    from oracle import sgp_oracle as O
    from test_gpu_parity import layers_of
    torch.manual_seed(8)
    n, t = 2500, 12
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=5)
    x = torch.randn(t, n, 16) * torch.tensor([1e5, 300.0, 1.0, 1e-4] * 4) + torch.tensor([0.0, 1e3, 0.0, 0.0] * 4)
    senc = sgp_amd.SGPSpatialEncoder(receptive_field=3, bidirectional=True, undirected=False, global_attr=False)
    y = senc(x.cuda(), ei, ew).cpu()
    ref = O.spatial_encoder_forward(x.double(), ei, ew, 3, True, False, False, sparse=True)
    e = col_err(y, ref)
    assert float(e.max()) <= 1e-5, float(e.max())
    enc = sgp_amd.SGPEncoder(input_size=3, reservoir_size=32, reservoir_layers=1, leaking_rate=0.9,
                             spectral_radius=0.9, density=0.7, input_scaling=1., receptive_field=2,
                             bidirectional=False, alpha_decay=False, global_attr=False, reservoir_activation="relu")
    xr = torch.randn(t, n, 3)
    y = enc(xr.cuda(), ei, ew).cpu()
    ref = O.sgp_encoder_forward(xr, ei, ew, layers_of(enc.reservoir), 2, bidirectional=False, global_attr=False,
                                activation="relu", dtype=torch.float64, sparse=True)
    e = col_err(y, ref.double())
    assert float(e.max()) <= 1e-5, float(e.max())


def test_tanh_encoder_with_a_tiny_input_scaling():
    """``input_scaling = 1e-6`` (the reference accepts any float, reservoir.py:60-62): with the reference's bias
    ~U(-1, 1) the states stay of order 1; with the bias scaled down too they are ~1e-6 under the a-priori bound 1 of
    a tanh reservoir.  Neither the reservoir (tanh with relative accuracy since round 5) nor the hops (their statistics
    send such states to the exact kernel) may lose them: the whole embedding against the fp64 oracle, column by column."""
    from oracle import sgp_oracle as O
    from test_gpu_parity import layers_of
    torch.manual_seed(9)
    n, t = 2500, 70
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=5)
    kw = dict(input_size=3, reservoir_size=32, reservoir_layers=1, leaking_rate=0.9, spectral_radius=0.9, density=0.7,
              input_scaling=1e-6, receptive_field=2, bidirectional=False, alpha_decay=False, global_attr=False)
    x = torch.randn(t, n, 3)
    for tiny_bias in (False, True):
        enc = sgp_amd.SGPEncoder(**kw)
        if tiny_bias:
            with torch.no_grad():
                for l in enc.reservoir.reservoir_layers:
                    l.b_ih.mul_(1e-6)
        y = enc(x.cuda(), ei, ew).cpu()
        if tiny_bias:
            assert 1e-7 < float(y[:, :, :32].abs().max()) < 1e-4              # tiny states
        ref = O.sgp_encoder_forward(x, ei, ew, layers_of(enc.reservoir), 2, bidirectional=False, global_attr=False,
                                    dtype=torch.float64, sparse=True)
        e = col_err(y, ref.double())
        assert float(e.max()) <= 1e-5, (tiny_bias, float(e.max()))


def test_leaking_rate_outside_the_unit_interval_is_measured():
    """``leaking_rate = 1.7`` (the reference accepts any float): states leave [-1, 1], the a-priori bound does not
    hold and the encoder must not claim it."""
    enc = sgp_amd.SGPEncoder(input_size=3, reservoir_size=32, reservoir_layers=1, leaking_rate=1.7,
                             spectral_radius=0.9, density=0.7, input_scaling=1., receptive_field=1,
                             bidirectional=False, alpha_decay=False, global_attr=False)
    assert enc._state_bound() is None
    ok = sgp_amd.SGPEncoder(input_size=3, reservoir_size=32, reservoir_layers=1, leaking_rate=0.9,
                            spectral_radius=0.9, density=0.7, input_scaling=1., receptive_field=1,
                            bidirectional=False, alpha_decay=False, global_attr=False)
    assert ok._state_bound() == 1.0
    assert ok._state_bound(torch.zeros(1, 4, 32)) is None              # a state from elsewhere: measured


def test_tiny_values_inside_a_large_column_meet_the_documented_absolute_bound(op):
    """The contract's residue, pinned (round-5 review): one column holds ~1e-9 values on half the graph and O(1) values on
    the other half.  The column's RMS is large, so the admission test passes and the SPLIT kernel runs; values 2^16
    below the column's bound are then carried to 2^-38 B ABSOLUTE, not relative.  What is promised and asserted: every
    result entry within 2^-37 B ||A||_inf (B rounded up to its power of two) + fp32's own 2^-21 sum_j |a_ij x_j|, and the encoder's allclose(1e-5, 1e-5); what is NOT promised: fp32's element-wise relative accuracy on the
    tiny half (an fp32 product resolves 1e-9 beside 1; here those entries carry up to ~1e-2 relative error)."""
    torch.manual_seed(11)
    x = torch.randn(T, N, D)
    c = 21
    x[:, :N // 2, c] *= 1e-9
    y, kernel = default_hop(op, x)
    assert kernel == "spmm_split"                                # admitted: the column's RMS is ~0.7
    ref64, cpu32 = products(op, x)
    yc = y.cpu()
    assert torch.allclose(yc, ref64.float(), rtol=1e-5, atol=1e-5)
    check_columns(y, ref64, cpu32)
    bound = float(x[:, :, c].abs().max())
    # (fp32's own term is relative to sum_j |a_ij| |x_j|, not to the -- possibly cancelled -- result)
    a_abs = torch.sparse_csr_tensor(op.rowptr.long(), op.col.long(), op.val.double().abs(), (op.num_nodes, op.num_cols))
    mag = torch.stack([a_abs @ x[b, :, c:c + 1].double().abs() for b in range(T)])[:, :, 0]
    limit = 2.0 ** -37 * bound * op.norm_inf() + 2.0 ** -21 * mag
    err = (yc[:, :, c].double() - ref64[:, :, c]).abs()
    assert bool((err <= limit).all()), float((err - limit).max())
    # the residue is real (this is what the bound is about): rows whose neighbours all lie in the tiny half
    tiny_rows = ref64[:, :, c].abs() < 1e-8
    assert int(tiny_rows.sum()) > 100
    rel = (err / ref64[:, :, c].abs().clamp_min(1e-300))[tiny_rows]
    assert 1e-4 < float(rel.median()) < 0.1                      # ~1e-2 RELATIVE on such entries (2^-37 B / |value|) ...
    rel32 = ((cpu32[:, :, c].double() - ref64[:, :, c]).abs() / ref64[:, :, c].abs().clamp_min(1e-300))[tiny_rows]
    assert float(rel32.median()) < 1e-6                          # ... where plain fp32 resolves them


def test_streamed_equals_one_pass_to_1e6_with_a_measured_bound():
    """relu reservoir (no a-priori bound: every hop MEASURES its operand): the per-column scales then depend on the
    time chunk, so a streamed encoding is no longer bit-identical to a single pass (tanh: a-priori bound 1, identical
    bits -- tests/test_gpu_parity.py::test_streamed_encoding_equals_single_pass).  Both are fp32-equivalent
    evaluations of the same products: they agree to 1e-6 of the block's scale (INTEGRATION.md says so)."""
    torch.manual_seed(12)
    n, t = 2600, 96
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=9)
    enc = sgp_amd.SGPEncoder(input_size=3, reservoir_size=64, reservoir_layers=1, leaking_rate=.9,
                             spectral_radius=.5, density=.7, input_scaling=1., receptive_field=3,
                             bidirectional=False, alpha_decay=False, global_attr=False, reservoir_activation="relu")
    x = torch.randn(t, n, 3)
    ops = enc.sgp_encoder.operators(n, ei, ew)
    full = enc.encode_device(x.cuda(), ops).cpu()
    assert ops[0].resolved_kernel() == "spmm_split"
    chunked = enc.encode_streamed(x, ops, 13)
    scale = float(full.abs().max())
    assert torch.equal(chunked[:, :, :64], full[:, :, :64])          # the recurrence itself is carried exactly
    assert float((chunked - full).abs().max()) <= 1e-6 * max(1.0, scale)


def test_exact_plans_are_built_only_after_a_flag_asked_for_them():
    """Round 6: where the split hop is the default, the exact kernels' tile / mix plans (16 s of host work on the target
    graph) are NOT built up front -- the generic CSR kernel sits behind the predicate.  A rejected operand is therefore
    computed by the CSR kernel the first time; the flag, copied to pinned memory behind the launch and read at a later
    call without a synchronisation, then makes the operator plan its exact kernels, which serve from there on."""
    ei, ew, _ = synthetic.knn_graph(N, K, seed=7)
    op = graph.ShiftOperator.from_edges(ei, ew, N)
    torch.manual_seed(5)
    good = torch.randn(T, N, D)
    bad = good * 3.0                                                    # with the caller's bound 1: data beyond it -> refused
    ref64, cpu32 = products(op, bad)
    y, kernel = default_hop(op, good)
    assert kernel == "spmm_split" and not any(k[0] is True for k in op._plans if isinstance(k, tuple) and len(k) == 2)
    assert op.prepare(D, torch.device("cuda"))[0] == "split"
    xg = bad.cuda()
    y1 = torch.full_like(xg, float("nan"))
    op.propagate(xg, y1, x_bound=1.0)
    assert op.last_exact_kernel == "spmm_csr_rows" and not op._exact_seen
    torch.cuda.synchronize()
    y2 = torch.full_like(xg, float("nan"))
    op.propagate(xg, y2, x_bound=1.0)                                   # the poll at the top of this call reads the flag ...
    assert op._exact_seen and op.last_exact_kernel in ("spmm_mix", "spmm_res", "spmm_tiled")   # ... and the planned kernels serve
    assert op.resolved_kernel() == op.last_exact_kernel
    for y in (y1, y2):
        check_columns(y, ref64, cpu32)
    # an accepted operand afterwards still takes the split kernel
    y3, kernel = default_hop(op, good)
    assert kernel == "spmm_split"


def test_a_nan_that_enters_the_recurrence_between_the_sampled_steps_is_seen():
    """Round-5 advice: with an a-priori bound the admission statistics read ~8 steps.  A NaN in the encoder's INPUT at an
    unsampled step turns that node's state NaN from there on -- the statistics now also read every row of the LAST step,
    where it still is, so the hops fall to the exact kernel and the NaN reaches the node's graph neighbours only (the
    reference's sparse product), not every row of the tiles that stage it."""
    torch.manual_seed(13)
    n, t = 2600, 40
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=3)
    enc = sgp_amd.SGPEncoder(input_size=3, reservoir_size=64, reservoir_layers=1, leaking_rate=.9, spectral_radius=.9,
                             density=.7, input_scaling=1., receptive_field=2, bidirectional=False, alpha_decay=False,
                             global_attr=False)
    x = torch.randn(t, n, 3)
    x[37, 777, 1] = float("nan")                                   # steps 0, 5, .., 35 are the strided sample
    ops = enc.sgp_encoder.operators(n, ei, ew)
    y = enc.encode_device(x.cuda(), ops)
    assert ops[0].resolved_kernel() != "spmm_split"
    # the reference's pattern: the generic CSR kernel (a sparse fp32 product) hop by hop on the states this run produced
    ref = y.clone()
    for h in range(2):
        ops[0].propagate(ref[:, :, 64 * h:64 * (h + 1)], ref[:, :, 64 * (h + 1):64 * (h + 2)], force="csr")
    # wherever the sparse fp32 product is NaN, so are we; beyond it at most the 16-row blocks of those rows (once a flag has
    # come back 0 the operator plans its exact matrix-core kernel, whose dense 16-row blocks multiply explicit zeros --
    # round-3 behaviour, test_data_beyond_the_callers_bound_and_non_finite_values; which hop that is depends on when the flag
    # arrives), never the 221-row tiles of the split kernel
    nan_y, nan_ref = torch.isnan(y), torch.isnan(ref)
    assert bool((nan_ref <= nan_y).all())
    for h in (1, 2):
        rows_y = nan_y[:, :, 64 * h:64 * (h + 1)].any(2).any(0)
        rows_ref = nan_ref[:, :, 64 * h:64 * (h + 1)].any(2).any(0)
        assert 0 < int(rows_ref.sum()) <= int(rows_y.sum()) <= 16 * int(rows_ref.sum())
    assert int(nan_y[:, :, 64:128].any(2).any(0).sum()) < 600
    fin = ~nan_y
    assert torch.allclose(y[fin], ref[fin], rtol=1e-5, atol=1e-5)
