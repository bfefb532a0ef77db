"""Host plan of the row-block SpMM kernel (sgp_amd/rowblock.py) checked WITHOUT a GPU: the
product evaluated from the plan arrays (``plan_reference``: staged rows, per-class streams,
lane-layout weights, row map) equals the dense product."""
import numpy as np
import pytest
import torch

from sgp_amd import graph, partition, rowblock, synthetic

MAX_UNION, WAVES = 608, 8


def _check(op, plan, feat=5):
    x = np.random.default_rng(0).standard_normal((op.num_cols, feat))
    y = rowblock.plan_reference(plan, x)
    ref = op.to_dense().double().numpy() @ x
    assert np.abs(y - ref).max() < 1e-6
    # structure the kernel relies on
    wptr, ns = plan.wptr.numpy(), plan.nsteps.numpy()
    assert (wptr % 4 == 0).all() and (np.diff(wptr) >= ns).all() and (np.diff(wptr) < ns + 4).all()
    assert (plan.usplit.numpy() % 4 == 0).all()
    assert plan.max_union <= MAX_UNION and int(np.diff(plan.uptr.numpy()).max()) == plan.max_union
    assert (plan.soff.numpy() % 256 == 0).all()
    rows = plan.rowmap.numpy()
    assert sorted(rows[rows >= 0].tolist()) == list(range(op.num_nodes))


@pytest.mark.parametrize("n,k", [(600, 20), (1300, 100), (300, 7)])
def test_plan_reproduces_the_operator(n, k):
    ei, ew, _ = synthetic.knn_graph(n, k, seed=3)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    plan = rowblock.build_rowblock_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n,
                                        MAX_UNION, WAVES)
    assert plan is not None and plan.n_tiles == -(-n // 128) and 0.3 < plan.fill <= 1.0
    _check(op, plan)


def test_ragged_and_empty_rows_duplicate_free_streams():
    torch.manual_seed(5)
    n = 700
    deg = torch.randint(0, 60, (n,))
    deg[::7] = 0
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    src = (tgt + torch.randint(-40, 41, tgt.shape)).clamp(0, n - 1)
    op = graph.ShiftOperator.from_edges(torch.stack([src, tgt]), torch.rand(tgt.numel()) + .1, n)
    plan = rowblock.build_rowblock_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n,
                                        MAX_UNION, WAVES)
    _check(op, plan)


def test_reordered_plan_keeps_original_ids():
    n = 2000
    ei, ew, _ = synthetic.knn_graph(n, 15, seed=6)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    op = graph.ShiftOperator.from_edges(perm[ei], ew, n)
    plain = rowblock.build_rowblock_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n,
                                         MAX_UNION, WAVES)
    assert plain is None or plain.fill < 0.4        # no locality in the numbering
    order = graph.locality_order(op.rowptr.numpy(), op.col.numpy(), n)
    plan = rowblock.build_rowblock_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n,
                                        MAX_UNION, WAVES, order=order)
    assert plan is not None and plan.reordered and plan.fill > 0.5 and plan.tile_rows == 128
    _check(op, plan)


def test_rectangular_block_of_a_node_partition():
    n = 1500
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=9)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    blk = partition.split_operator(op, partition.partition_bounds(n, 3), 1)
    b = blk.op
    assert b.num_cols > b.num_nodes
    plan = rowblock.build_rowblock_plan(b.rowptr.numpy(), b.col.numpy(), b.val.numpy(), b.num_nodes,
                                        MAX_UNION, WAVES)
    _check(b, plan)
