"""Run-to-run determinism stress of the hand-scheduled hop kernels (round-3 review: ``spmm_mix`` keeps LDS reads
in flight past loop exits and relies on pinned registers; one flaky failure was seen during development).
Every kernel runs 50 times on four graphs -- k-NN, ragged with empty rows, a halo block of a node partition,
scrambled labels -- with T spanning at least three time chunks: every run must reproduce the first BIT FOR BIT
and the first must match the dense fp64 product (``x = adj @ x``, lib/sgp_preprocessing.py:200-203)."""
import pytest
import torch

from sgp_amd import graph, hip, partition, synthetic
from test_gpu_parity import close, dense_ref

pytestmark = pytest.mark.gpu
REPEATS = 50


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.require_gpu()


def _cases():
    torch.manual_seed(0)
    ei, ew, _ = synthetic.knn_graph(3000, 100, seed=4)
    yield "knn100", graph.ShiftOperator.from_edges(ei, ew, 3000), None, 70
    n = 2500
    deg = torch.randint(0, 60, (n,)); deg[::7] = 0
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    src = (tgt + torch.randint(-40, 41, tgt.shape)).clamp(0, n - 1)
    yield "ragged", graph.ShiftOperator.from_edges(torch.stack([src, tgt]), torch.rand(tgt.numel()) + .1, n), None, 67
    ei, ew, _ = synthetic.knn_graph(7000, 60, seed=9)
    full = graph.ShiftOperator.from_edges(ei, ew, 7000)
    yield "halo", full, 1, 97
    ei, ew, _ = synthetic.knn_graph(4000, 30, seed=6)
    perm = torch.randperm(4000, generator=torch.Generator().manual_seed(1))
    yield "scrambled", graph.ShiftOperator.from_edges(perm[ei], ew, 4000), None, 70


@pytest.mark.parametrize("force", ["mix", "res", "split"])
def test_hop_kernels_repeat_bit_for_bit(force):
    for name, op, rank, t in _cases():
        d = 64
        x = torch.tanh(torch.randn(t, op.num_nodes, d, generator=torch.Generator().manual_seed(7)))
        ref = dense_ref(op, x)
        halo = None
        if rank is not None:                                   # local block of a 3-way partition
            bounds = partition.partition_bounds(op.num_nodes, 3)
            blk = partition.split_operator(op, bounds, rank)
            assert blk.n_halo > 0
            recv = x[:, blk.halo_global].permute(1, 0, 2).contiguous().cuda()
            halo = recv.permute(1, 0, 2)
            xg = x[:, blk.lo:blk.hi].cuda().contiguous()
            ref = ref[:, blk.lo:blk.hi]
            run_op = blk.op
        else:
            xg, run_op = x.cuda(), op
        first = torch.full((t, run_op.num_nodes, d), float("nan"), device="cuda")
        kw = dict(x_bound=1.0) if force == "split" else {}
        run_op.propagate(xg, first, force=force, halo=halo, **kw)
        close(first, ref)
        y = torch.empty_like(first)
        for i in range(REPEATS):
            y.fill_(float("nan"))
            run_op.propagate(xg, y, force=force, halo=halo, **kw)
            assert torch.equal(y, first), (name, force, i, float((y - first).abs().max()))
