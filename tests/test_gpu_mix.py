"""GPU parity of the mixed dense (v_mfma_f32_16x16x4_f32) / sparse (4x4x1 row groups) hop kernel
``sgp_spmm_mix_f32`` (reference: ``x = adj @ x``, lib/sgp_preprocessing.py:200-203) through the C ABI:
against the dense fp64 product (rtol = atol = 1e-5 and relative Frobenius error <= 1e-5, the tolerance
north_star states) and against the generic CSR kernel.  ``force="mix"`` plans every graph, also those the
product path would not give to this kernel (few shared columns), so the zero-dense and overflow paths run."""
import numpy as np
import pytest
import torch

from sgp_amd import graph, hip, mixplan, partition, synthetic
from test_gpu_parity import close, dense_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.require_gpu()


def run_mix(op, x, thr=None, halo=None):
    y = torch.full((x.shape[0], op.num_nodes, x.shape[2]), float("nan"), device="cuda")
    if thr is None:
        op.propagate(x.cuda(), y, force="mix", halo=halo)
        assert op.last_kernel == "spmm_mix"
        return y
    dev = torch.device("cuda")
    base = op.tile_plan(x.shape[2], dev, tall=False)
    lib = hip.load()
    plan = mixplan.build_mix_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), op.num_nodes, base,
                                  thr=thr, dh=lib.sgp_spmm_mix_max_dense(int(halo is not None)))
    hip.spmm_mix(plan.to(dev), x.cuda(), y, halo, op.num_nodes)
    return y


@pytest.mark.parametrize("n,k,feat,t", [(1500, 20, 64, 5), (1500, 100, 64, 5), (900, 33, 128, 3),
                                        (3000, 100, 64, 40), (700, 20, 192, 2), (207, 8, 64, 7),
                                        (2500, 7, 64, 3)])
@pytest.mark.parametrize("thr", [4, 3, 2])
def test_mix_knn_graphs(n, k, feat, t, thr):
    """k-NN graphs from dense-heavy (k = 100: half of the pairs go through 16x16x4) to hardly any shared
    column (k = 7); t = 40 spans two time chunks; thr = columns shared by >= thr of a block's 4 groups."""
    torch.manual_seed(n + k)
    ei, ew, _ = synthetic.knn_graph(n, k, seed=7)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.randn(t, n, feat)
    y = run_mix(op, x, thr=thr)
    close(y, dense_ref(op, x))
    y2 = torch.empty_like(y)
    op.propagate(x.cuda(), y2, force="csr")
    close(y, y2, rtol=1e-6, atol=1e-6)


def test_mix_ragged_empty_rows_and_long_ranges():
    """Empty rows, ragged degrees, and row groups that mix two distant neighbourhoods: their column
    lists exceed the resident super-steps (extension quads, then the LDS walk)."""
    torch.manual_seed(6)
    n, feat, t = 2048, 64, 35
    deg = torch.randint(0, 60, (n,))
    deg[::7] = 0
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    src = (tgt + torch.randint(-40, 41, tgt.shape)).clamp(0, n - 1)
    far = torch.arange(0, n, 5)
    far_t = torch.repeat_interleave(far, 100)
    far_s = (far_t + 150 + torch.randint(0, 160, far_t.shape)) % n
    ei = torch.stack([torch.cat([src, far_s]), torch.cat([tgt, far_t])])
    op = graph.ShiftOperator.from_edges(ei, torch.rand(ei.shape[1]) + .1, n)
    mp = op.mix_plan(feat, torch.device("cuda"), strict=False)
    assert mp is not None and mp.max_range_steps > 28          # beyond what the registers hold
    x = torch.randn(t, n, feat)
    y = run_mix(op, x)
    close(y, dense_ref(op, x))
    empty = (op.rowptr[1:] == op.rowptr[:-1]).nonzero().flatten()
    assert empty.numel() > 0 and float(y[:, empty].abs().max()) == 0.0


def test_mix_traffic_graph_and_in_place_slots():
    ei, ew = synthetic.sparse_traffic_graph(325, 2369, seed=2)
    op = graph.ShiftOperator.from_edges(ei, ew, 325)
    x = torch.randn(100, 325, 128)
    close(run_mix(op, x), dense_ref(op, x))
    # hop k reads slot k-1 and writes slot k of the SAME [T, N, P*D] buffer
    torch.manual_seed(0)
    n, t, d, p = 1100, 6, 64, 4
    ei, ew, _ = synthetic.knn_graph(n, 60, seed=3)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    out = torch.randn(t, n, p * d, device="cuda")
    first = out[:, :, :d].clone()
    for k in range(1, p):
        op.propagate(out[:, :, (k - 1) * d:k * d], out[:, :, k * d:(k + 1) * d], force="mix")
    for k in range(1, p):
        close(out[:, :, k * d:(k + 1) * d], dense_ref(op, out[:, :, (k - 1) * d:k * d]))
    assert torch.equal(out[:, :, :d], first)


@pytest.mark.parametrize("world", [2, 3])
def test_mix_partitioned_blocks_with_halo(world):
    """Local blocks of a node partition: halo rows arrive as a second source in the [rows, T, D] layout
    the all_to_all produces (arbitrary strides)."""
    torch.manual_seed(world)
    n, t, d = 2000, 6, 64
    ei, ew, _ = synthetic.knn_graph(n, 60, seed=9)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.randn(t, n, d)
    ref = dense_ref(op, x)
    bounds = partition.partition_bounds(n, world)
    for r in range(world):
        blk = partition.split_operator(op, bounds, r)
        assert blk.n_halo > 0
        xo = x[:, blk.lo:blk.hi].cuda().contiguous()
        recv = x[:, blk.halo_global].permute(1, 0, 2).contiguous().cuda()
        y = run_mix(blk.op, xo, halo=recv.permute(1, 0, 2))
        close(y, ref[:, blk.lo:blk.hi])


def test_mix_scrambled_labels_use_the_reordered_plan():
    n, t, d = 4000, 4, 64
    ei, ew, _ = synthetic.knn_graph(n, 40, seed=3)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    op = graph.ShiftOperator.from_edges(perm[ei], ew, n)
    base = op.tile_plan(d, torch.device("cuda"), tall=False)
    assert base is not None and base.reordered
    x = torch.randn(t, n, d)
    y = run_mix(op, x)
    close(y, dense_ref(op, x))


def test_mix_full_size_target_graph_properties():
    """N = 100 000, 100-NN (the target line's graph), properties that need no oracle at this size:
    agreement with the generic CSR kernel, rows sum to 1 (A 1 = 1), linearity."""
    n, d, t = 100000, 64, 3
    ei, ew, _ = synthetic.knn_graph(n, 100, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    mp = op.mix_plan(d, torch.device("cuda"))
    assert mp is not None and mp.dense_share > 0.4
    g = torch.Generator(device="cuda").manual_seed(0)
    x1 = torch.randn(t, n, d, device="cuda", generator=g)
    x2 = torch.randn(t, n, d, device="cuda", generator=g)
    ya, yb, yc, yr = (torch.empty_like(x1) for _ in range(4))
    op.propagate(x1, ya, force="mix"); op.propagate(x2, yb, force="mix")
    op.propagate(2 * x1 - 3 * x2, yc, force="mix")
    close(yc, 2 * ya - 3 * yb, rtol=1e-5, atol=1e-5, fro=2e-6)
    op.propagate(x1, yr, force="csr")
    close(ya, yr, rtol=1e-5, atol=1e-5, fro=2e-6)
    op.propagate(torch.ones_like(x1), ya, force="mix")
    assert float((ya - 1).abs().max()) < 1e-5
