"""GPU parity of the split-fp16 hop ``sgp_spmm_split_f32`` (reference: ``x = adj @ x``,
lib/sgp_preprocessing.py:200-203) through the C ABI.  Criteria: the north_star tolerance against the dense
fp64 product (``allclose(rtol = atol = 1e-5)`` and relative Frobenius error <= 1e-5) AND -- because this
kernel does not form exact fp32 products -- a bound TEN times tighter on its distance to the fp64 product:
max |y - fp64| <= 1e-6 of the result's scale, and no more than 4x the distance of the reference's own
arithmetic (a CPU fp32 sparse product, what ``adj @ x`` runs) plus 1e-7 of the scale.  Measured: 1-5e-7 of
the scale, between the fp32 CPU product and the GPU's exact-fp32 kernels (hi + lo carries 22 bits per
operand; the three partial products are exact and summed in fp32 by the matrix core)."""
import numpy as np
import pytest
import torch

from sgp_amd import graph, hip, partition, synthetic
from test_gpu_parity import close, dense_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.require_gpu()


def run_split(op, x, bound=None, t_chunk=None):
    y = torch.full((x.shape[0], op.num_nodes, x.shape[2]), float("nan"), device="cuda")
    if t_chunk is None:
        op.propagate(x.cuda(), y, force="split", x_bound=bound)
        assert op.last_kernel == "spmm_split"
    else:
        hip.spmm_split(op.split_plan(torch.device("cuda")), x.cuda(), y,
                       float(x.abs().max()) if bound is None else bound, t_chunk=t_chunk)
    return y


def as_good_as_fp32(op, x, y):
    """max |y - fp64| <= 1e-6 scale and <= 4 max |CPU fp32 product - fp64| + 1e-7 scale."""
    xc = x.cpu()
    a64 = torch.sparse_csr_tensor(op.rowptr.long(), op.col.long(), op.val.double(), (op.num_nodes, op.num_cols))
    a32 = torch.sparse_csr_tensor(op.rowptr.long(), op.col.long(), op.val, (op.num_nodes, op.num_cols))
    ref64 = torch.stack([a64 @ xc[b].double() for b in range(xc.shape[0])])
    cpu32 = torch.stack([a32 @ xc[b] for b in range(xc.shape[0])])
    e_split = float((y.cpu().double() - ref64).abs().max())
    e_cpu = float((cpu32.double() - ref64).abs().max())
    scale = float(ref64.abs().max()) or 1.0
    assert e_split <= 1e-6 * scale, (e_split, scale)
    assert e_split <= 4 * e_cpu + 1e-7 * scale, (e_split, e_cpu, scale)
    return e_split, e_cpu


@pytest.mark.parametrize("n,k,feat,t", [(1500, 20, 64, 5), (1500, 100, 64, 5), (900, 33, 128, 3),
                                        (3000, 100, 64, 40), (700, 20, 192, 2), (207, 8, 64, 7),
                                        (2500, 7, 16, 9), (1200, 50, 32, 70), (4000, 100, 48, 3)])
def test_split_knn_graphs(n, k, feat, t):
    """k-NN graphs from 7 to 100 neighbours, feature widths 16 .. 192 (1 .. 12 slices of 16), t up to 70 steps
    (several time chunks, odd unit counts, the paired stores of even / odd slices)."""
    torch.manual_seed(n + k)
    ei, ew, _ = synthetic.knn_graph(n, k, seed=7)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.tanh(torch.randn(t, n, feat))
    y = run_split(op, x, bound=1.0)
    close(y, dense_ref(op, x))
    as_good_as_fp32(op, x, y)


@pytest.mark.parametrize("scale", [1e-6, 1e-2, 37.0, 3e4, 1e12])
def test_split_operand_scales(scale):
    """Operands far from 1: the measured bound (``x_bound=None``: sgp_abs_max_f32) sets the scale; relative
    accuracy is that of the unit-scale case."""
    torch.manual_seed(3)
    n, k, feat, t = 1100, 40, 64, 6
    ei, ew, _ = synthetic.knn_graph(n, k, seed=2)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.randn(t, n, feat) * scale
    assert abs(hip.abs_max(x.cuda()) - float(x.abs().max())) == 0.0
    y = run_split(op, x)                                              # bound measured on the device
    ref = dense_ref(op, x)
    close(y / scale, ref / scale)
    as_good_as_fp32(op, x, y)


def test_split_unnormalised_weights_duplicates_empty_rows():
    """Weights up to ~17 after duplicate edges are summed, empty rows, ragged degrees, a zero operand."""
    rng = np.random.default_rng(0)
    n, feat, t = 2300, 64, 5
    deg = rng.integers(0, 60, n)
    deg[::7] = 0
    tgt = np.repeat(np.arange(n), deg)
    src = np.clip(tgt + rng.integers(-40, 41, tgt.size), 0, n - 1)
    ei = torch.from_numpy(np.stack([src, tgt]))
    op = graph.ShiftOperator(torch.from_numpy(np.concatenate([[0], np.cumsum(deg)])), torch.from_numpy(src),
                             torch.from_numpy(rng.random(tgt.size).astype(np.float32) * 3 + .1), n)
    x = torch.randn(t, n, feat)
    y = run_split(op, x)
    close(y, dense_ref(op, x), atol=1e-5 * float(op.norm_inf()))
    as_good_as_fp32(op, x, y)
    empty = (op.rowptr[1:] == op.rowptr[:-1]).nonzero().flatten()
    assert empty.numel() > 0 and float(y[:, empty].abs().max()) == 0.0
    y0 = run_split(op, torch.zeros(2, n, feat))
    assert float(y0.abs().max()) == 0.0


def test_split_in_place_slots_and_bound_bookkeeping():
    """Hop k reads slot k-1 and writes slot k of the SAME [T, N, P*D] buffer (row stride P*D); the bound of
    hop k is bound_{k-1} * ||A||_inf, as ``propagate_into`` keeps it."""
    from sgp_amd.sgp_preprocessing import propagate_into
    torch.manual_seed(0)
    n, t, d, p = 2600, 6, 64, 4
    ei, ew, _ = synthetic.knn_graph(n, 60, seed=3)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    out = torch.randn(t, n, p * d, device="cuda")
    out[:, :, :d] = torch.tanh(out[:, :, :d])
    first = out[:, :, :d].clone()
    assert op.split_eligible(out[:, :, :d], out[:, :, d:2 * d])
    propagate_into(out, d, [op], p - 1, x_bound=1.0)
    assert op.last_kernel == "spmm_split"
    for k in range(1, p):
        close(out[:, :, k * d:(k + 1) * d], dense_ref(op, out[:, :, (k - 1) * d:k * d]))
    assert torch.equal(out[:, :, :d], first)
    out2 = out.clone()
    propagate_into(out2, d, [op], p - 1)                               # bound measured once, then propagated
    close(out2, out, rtol=1e-6, atol=1e-6)


def test_split_time_chunks_and_run_to_run_determinism():
    """Every chunk length gives the same bits (a unit's arithmetic does not depend on its neighbours), and so
    do repeated runs (fixed summation order)."""
    torch.manual_seed(5)
    n, feat, t = 3100, 64, 37
    ei, ew, _ = synthetic.knn_graph(n, 100, seed=4)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.tanh(torch.randn(t, n, feat))
    y = run_split(op, x, bound=1.0, t_chunk=64)
    for tc in (1, 2, 5, 8, 37):
        assert torch.equal(run_split(op, x, bound=1.0, t_chunk=tc), y)
    for _ in range(20):
        assert torch.equal(run_split(op, x, bound=1.0, t_chunk=8), y)
    close(y, dense_ref(op, x))


def test_split_is_not_chosen_where_it_cannot_serve():
    """Widths that are not multiples of 16, small graphs and non-finite bounds keep the exact-fp32 kernels; a row
    longer than a wave's column budget is served in several passes (``splitplan.build_split_passes``)."""
    n = 3000
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=1)
    hub = torch.stack([torch.arange(400), torch.full((400,), 5)])      # row 5 gains 400 columns
    op = graph.ShiftOperator.from_edges(torch.cat([ei, hub], 1), torch.cat([ew, torch.ones(400)]), n)
    passes = op.split_plan(torch.device("cuda"))
    assert isinstance(passes, list) and len(passes) >= 2 and not passes[0].accumulate and passes[1].accumulate
    x = torch.randn(2, n, 64, device="cuda")
    y = torch.full_like(x, float("nan"))
    op.propagate(x, y)
    assert op.resolved_kernel() == "spmm_split"
    close(y, dense_ref(op, x))
    y.fill_(float("nan"))
    op.propagate(x, y, force="split")
    close(y, dense_ref(op, x))
    as_good_as_fp32(op, x, y)
    op2 = graph.ShiftOperator.from_edges(ei, ew, n)
    x20 = torch.randn(2, n, 20, device="cuda")
    op2.propagate(x20, torch.empty_like(x20))
    assert op2.last_kernel != "spmm_split"
    op2.propagate(x, y, x_bound=float("inf"))                           # a non-finite bound falls back
    assert op2.last_kernel != "spmm_split"
    close(y, dense_ref(op2, x))
    with pytest.raises(ValueError):
        op2.propagate(x, y, force="split", x_bound=float("inf"))


@pytest.mark.parametrize("n,deg", [(5016, 740), (3000, 495)])
def test_split_long_rows_full_graph_shapes(n, deg):
    """The reference's FULL large-scale graphs (config/largescale/sgp_pv.yaml with adj_knn=None,
    experiments/run_largescale_sgp.py:167-170: ~740 entries per row on 5 016 nodes; CER-En ~495): every group of 16
    rows has its column union cut into segments of one wave's budget (the kernel's wide form: 448 columns), one launch per
    segment, results accumulated in place; SGP_TUNE=split_wide=0: the standard form's 224-column segments.  Parity against the dense fp64 product, per column, and against the CPU fp32 product."""
    torch.manual_seed(n)
    ei, ew, _ = synthetic.threshold_graph(n, deg, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    assert op.max_degree() > 256
    passes = op.split_plan(torch.device("cuda"))
    assert isinstance(passes, list) and len(passes) >= 3
    # most rows are long: the WIDE form (8 waves x 448 columns per wave) -- half the passes of the standard form
    assert all(tuple(p.afr.shape[1:3]) == (8, 14) for p in passes) and len(passes) <= 5
    assert passes[0].stats["rows_per_wave"] > 12 and passes[0].stats["staged_per_row"] < 8
    t, feat = 5, 128
    out = torch.randn(t, n, 3 * feat, device="cuda")
    out[:, :, :feat] = torch.tanh(out[:, :, :feat])
    x, y = out[:, :, :feat], out[:, :, feat:2 * feat]                   # strided slots, as the encoder uses them
    op.propagate(x, y, x_bound=1.0)
    assert op.resolved_kernel() == "spmm_split"
    ref = dense_ref(op, x)
    close(y, ref)
    as_good_as_fp32(op, x.contiguous(), y)
    y2 = out[:, :, 2 * feat:]
    op.propagate(x, y2, force="mix") if op.mix_plan(feat, x.device, strict=False) is not None else op.propagate(x, y2, force="csr")
    close(y, y2, rtol=1e-6, atol=1e-6, fro=2e-6)


def test_split_encoder_matches_the_oracle_and_the_exact_kernels(monkeypatch):
    """SGPEncoder on a graph large enough for the split hop: against the CPU oracle (1e-5) and against the
    same encoder with SGP_TUNE=hop=exact."""
    import sgp_amd
    from oracle import sgp_oracle as O
    from test_gpu_parity import layers_of
    torch.manual_seed(11)
    n, t, f = 2500, 24, 3
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=5)
    enc = sgp_amd.SGPEncoder(input_size=f, reservoir_size=32, reservoir_layers=2, leaking_rate=0.9,
                             spectral_radius=0.9, density=0.7, input_scaling=1., receptive_field=3,
                             bidirectional=True, alpha_decay=False, global_attr=True)
    x = torch.randn(t, n, f)
    y = enc(x.cuda(), ei, ew).cpu()
    ref = O.sgp_encoder_forward(x, ei, ew, layers_of(enc.reservoir), 3, bidirectional=True, global_attr=True,
                                sparse=True)
    close(y, ref)
    monkeypatch.setenv("SGP_TUNE", "hop=exact")
    y_exact = enc(x.cuda(), ei, ew).cpu()
    close(y, y_exact, rtol=1e-6, atol=1e-6)


def test_split_full_size_target_graph_properties():
    """N = 100 000, 100-NN (the target line's graph), T = 9 (units of two time chunks): agreement with the
    generic CSR kernel, rows sum to 1 (A 1 = 1), linearity."""
    n, d, t = 100000, 64, 9
    ei, ew, _ = synthetic.knn_graph(n, 100, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    plan = op.split_plan(torch.device("cuda"))
    assert plan is not None and plan.stats["staged_per_row"] < 3.5
    g = torch.Generator(device="cuda").manual_seed(0)
    x1 = torch.tanh(torch.randn(t, n, d, device="cuda", generator=g))
    x2 = torch.tanh(torch.randn(t, n, d, device="cuda", generator=g))
    ya, yb, yc, yr = (torch.empty_like(x1) for _ in range(4))
    op.propagate(x1, ya, force="split", x_bound=1.0); op.propagate(x2, yb, force="split", x_bound=1.0)
    op.propagate(2 * x1 - 3 * x2, yc, force="split", x_bound=5.0)
    close(yc, 2 * ya - 3 * yb, rtol=1e-5, atol=1e-5, fro=2e-6)
    op.propagate(x1, yr, force="csr")
    close(ya, yr, rtol=1e-5, atol=1e-5, fro=2e-6)
    op.propagate(torch.ones_like(x1), ya, force="split", x_bound=1.0)
    assert float((ya - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("world", [2, 3])
def test_split_partitioned_blocks_with_halo(world):
    """Local blocks of a node partition: halo rows arrive as a second source in the [rows, T, D] layout the
    all_to_all produces (row stride T * D, batch stride D); the bound covers both sources."""
    torch.manual_seed(world)
    n, t, d = 7500, 6, 64                                           # blocks of >= 2048 rows: the split kernel's floor
    ei, ew, _ = synthetic.knn_graph(n, 60, seed=9)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.tanh(torch.randn(t, n, d))
    ref = dense_ref(op, x)
    bounds = partition.partition_bounds(n, world)
    for r in range(world):
        blk = partition.split_operator(op, bounds, r)
        assert blk.n_halo > 0
        xo = x[:, blk.lo:blk.hi].cuda().contiguous()
        recv = x[:, blk.halo_global].permute(1, 0, 2).contiguous().cuda()
        halo = recv.permute(1, 0, 2)
        y = torch.full((t, blk.n_own, d), float("nan"), device="cuda")
        blk.op.propagate(xo, y, halo=halo, x_bound=1.0)
        assert blk.op.last_kernel == "spmm_split"                   # the default on a block of this size
        close(y, ref[:, blk.lo:blk.hi])
        y2 = torch.empty_like(y)
        blk.op.propagate(xo, y2, halo=halo)                         # bound measured over both sources
        close(y2, y, rtol=1e-6, atol=1e-6)


def test_wide_form_on_partitioned_blocks_with_halo():
    """The register-staged WIDE form (long rows: 4-pass plans, accumulating) with a halo source: the two blocks of a
    2-way partition of a ~420-entries-per-row graph, halo rows in the all_to_all layout, 3 time chunks."""
    torch.manual_seed(5)
    n, t, d = 4300, 40, 128
    ei, ew, _ = synthetic.threshold_graph(n, 420, seed=6)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.tanh(torch.randn(t, n, d))
    ref = dense_ref(op, x)
    bounds = partition.partition_bounds(n, 2)
    for r in range(2):
        blk = partition.split_operator(op, bounds, r)
        plan = blk.op.split_plan(torch.device("cuda"))
        assert isinstance(plan, list) and len(plan) >= 2 and tuple(plan[0].afr.shape[1:3]) == (8, 14) and blk.n_halo > 0
        xo = x[:, blk.lo:blk.hi].cuda().contiguous()
        recv = x[:, blk.halo_global].permute(1, 0, 2).contiguous().cuda()
        y = torch.full((t, blk.n_own, d), float("nan"), device="cuda")
        blk.op.propagate(xo, y, halo=recv.permute(1, 0, 2), x_bound=1.0)
        assert blk.op.resolved_kernel() == "spmm_split"
        close(y, ref[:, blk.lo:blk.hi])
        for _ in range(5):                                          # run to run: the same bits
            y2 = torch.empty_like(y)
            blk.op.propagate(xo, y2, halo=recv.permute(1, 0, 2), x_bound=1.0)
            assert torch.equal(y2, y)
