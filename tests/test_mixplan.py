"""Host plan of the mixed dense / sparse hop kernel (sgp_amd/mixplan.py) checked WITHOUT a GPU: the
product evaluated from the plan arrays (``mix_reference``: staged rows, 4-row-group streams, dense
16x16x4 instruction lists in lane order, row map) equals the dense product, and the structure the
kernel relies on holds."""
import numpy as np
import pytest
import torch

from sgp_amd import graph, mixplan, partition, synthetic

LIMITS = dict(max_union=448, max_tile_rows=64, max_row_edges=128)
DH = 10


def _plan(op, thr=4, dh=DH, order=None):
    args = (op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), op.num_nodes)
    base = graph.build_tile_plan(*args, **LIMITS) if order is None else \
        graph.build_reordered_plan(*args, order, **LIMITS)
    assert base is not None and base.pipe is not None
    return mixplan.build_mix_plan(*args, base, thr=thr, dh=dh, order=order)


def _check(op, plan, feat=5):
    a = plan.arrays()
    x = np.random.default_rng(0).standard_normal((op.num_cols, feat))
    y = mixplan.mix_reference(a, plan.n_tiles, x)
    ref = op.to_dense().double().numpy() @ x
    assert np.abs(y - ref).max() < 1e-6
    assert (a["usplit"] % 4 == 0).all() and (a["gidx"] % 256 == 0).all() and (a["didx"] % 256 == 0).all()
    nd = np.diff(a["dptr"])
    assert nd.min() >= 0 and nd.max() == plan.max_dense <= plan.dh
    # staged-row offsets stay inside the tile's stage; a segment's dense columns lie in that segment
    U = np.diff(a["uptr"])
    for t in range(plan.n_tiles):
        for b in range(4):
            for ph in range(2):
                m0, m1 = a["dptr"][(t * 4 + b) * 2 + ph], a["dptr"][(t * 4 + b) * 2 + ph + 1]
                slots = a["didx"][m0:m1] // 256
                if slots.size:
                    assert slots.max() < U[t]
                    assert ((slots >= a["usplit"][t]) == bool(ph)).all()
    rows = a["rowmap"]
    assert sorted(rows[rows >= 0].tolist()) == list(range(op.num_nodes))


@pytest.mark.parametrize("n,k,thr", [(600, 20, 4), (1300, 100, 4), (1300, 100, 3), (300, 7, 4), (1000, 60, 2)])
def test_plan_reproduces_the_operator(n, k, thr):
    ei, ew, _ = synthetic.knn_graph(n, k, seed=3)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    plan = _plan(op, thr=thr)
    _check(op, plan)
    if k == 100:
        assert plan.dense_share > 0.4 and plan.max_dense == DH      # about half of the pairs are dense


def test_dense_cap_demotes_to_the_sparse_stream():
    ei, ew, _ = synthetic.knn_graph(1300, 100, seed=3)
    op = graph.ShiftOperator.from_edges(ei, ew, 1300)
    wide, narrow = _plan(op, dh=10), _plan(op, dh=3)
    assert narrow.max_dense == 3 and narrow.dense_share < wide.dense_share
    _check(op, narrow)


def test_ragged_and_empty_rows():
    torch.manual_seed(5)
    n = 700
    deg = torch.randint(0, 60, (n,))
    deg[::7] = 0
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    src = (tgt + torch.randint(-40, 41, tgt.shape)).clamp(0, n - 1)
    op = graph.ShiftOperator.from_edges(torch.stack([src, tgt]), torch.rand(tgt.numel()) + .1, n)
    _check(op, _plan(op))


def test_reordered_plan_keeps_original_ids():
    n = 2000
    ei, ew, _ = synthetic.knn_graph(n, 15, seed=6)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    op = graph.ShiftOperator.from_edges(perm[ei], ew, n)
    order = graph.locality_order(op.rowptr.numpy(), op.col.numpy(), n)
    plan = _plan(op, order=order)
    assert plan.reordered
    _check(op, plan)


def test_rectangular_block_of_a_node_partition():
    n = 1500
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=9)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    blk = partition.split_operator(op, partition.partition_bounds(n, 3), 1)
    assert blk.op.num_cols > blk.op.num_nodes
    _check(blk.op, _plan(blk.op, dh=8))


def test_operator_policy_and_classes_balance():
    """ShiftOperator.mix_plan: a 100-NN graph gets the kernel, a sparse k-NN graph does not (too few
    shared columns) unless forced; the four SIMD classes of a tile carry about the same work."""
    ei, ew, _ = synthetic.knn_graph(2000, 100, seed=2)
    op = graph.ShiftOperator.from_edges(ei, ew, 2000)
    limits = dict(max_union=448, max_tile_rows=64, max_row_edges=128)
    op.tile_plan(64, torch.device("cpu"), limits=limits)
    mp = op.mix_plan(64, torch.device("cpu"))
    assert mp is not None and mp.dense_share > 0.4
    base_cost = float(op.tile_plan(64, torch.device("cpu"), tall=False).pipe["phase_cost"].mean())
    assert mp.mean_phase_cost <= 1.03 * base_cost            # demotion keeps the padding out
    ei, ew, _ = synthetic.knn_graph(2000, 8, seed=2)
    op2 = graph.ShiftOperator.from_edges(ei, ew, 2000)
    op2.tile_plan(64, torch.device("cpu"), limits=limits)
    assert op2.mix_plan(64, torch.device("cpu")) is None
    assert op2.mix_plan(64, torch.device("cpu"), strict=False) is not None
