"""Column-blocked hop (graphs without locality): host plan checks on CPU, kernel parity on the GPU."""
import numpy as np
import pytest
import torch

from sgp_amd import colblock, graph, synthetic


def _plan(n, k, feat=64, seed=1, **kw):
    ei, ew = synthetic.random_graph(n, k, seed=seed)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    rowptr, col, val = op.csr()
    return op, colblock.build_colblock_plan(rowptr.numpy(), col.numpy(), val.numpy(), n, n, feat, **kw)


@pytest.mark.parametrize("n,k,kw", [(300, 7, dict(n_wg=5, l2_bytes=64 * 256)),
                                    (2000, 30, dict(n_wg=16, l2_bytes=300 * 256, rows_cap=64)),
                                    (50, 50, dict(n_wg=256)),
                                    (1000, 3, dict(n_wg=7, round_pad=2))])
def test_plan_encodes_the_matrix(n, k, kw):
    op, p = _plan(n, k, **kw)
    assert np.array_equal(colblock.plan_to_dense(p).astype(np.float32), op.to_dense().numpy())
    seg = p.segptr.numpy()
    pad = kw.get("round_pad", 4)
    assert seg[0] == 0 and np.all(np.diff(seg) >= 0) and np.all(np.diff(seg) % pad == 0)
    assert (seg[-1] + 2 * pad) * 64 == p.entries.shape[0]
    b = p.wg_row0.numpy()
    assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) > 0) and np.diff(b).max() <= kw.get("rows_cap", 511)
    # every entry addresses a column of its segment's block; a real entry a row that its slot owns
    e = p.entries.numpy()
    ex = e[:, 0].view(np.uint32).astype(np.int64).reshape(-1, 64)[:seg[-1]]
    w = e[:, 1].view(np.float32).reshape(-1, 64)[:seg[-1]]
    seg_of = np.repeat(np.arange(p.n_wg * p.n_blocks), np.diff(seg))
    assert np.array_equal((ex & 0x3fffff) // p.cols_per_block, np.repeat((seg_of % p.n_blocks)[:, None], 64, 1))
    lrow = (ex >> 22) & 511
    real = w != 0
    owner = {}                                                   # a row belongs to exactly one slot of its workgroup
    for i, sl in zip(*np.nonzero(real)):
        key = (seg_of[i] // p.n_blocks, lrow[i, sl])
        assert owner.setdefault(key, sl) == sl
    assert np.all((lrow < np.diff(b)[seg_of // p.n_blocks][:, None])[real])
    assert np.all(lrow[~real] == colblock.PAD_ROW)               # padding sums into the spare row


def test_plan_ragged_and_empty_rows():
    n = 400
    rng = np.random.default_rng(0)
    deg = rng.integers(0, 40, n); deg[::7] = 0; deg[5] = 300
    row = np.repeat(np.arange(n), deg)
    col = rng.integers(0, n, row.size)
    ei = torch.from_numpy(np.stack([col, row]))                    # (source, target) like the reference's edge_index
    op = graph.ShiftOperator.from_edges(ei, torch.rand(row.size), n)
    rowptr, c, v = op.csr()
    p = colblock.build_colblock_plan(rowptr.numpy(), c.numpy(), v.numpy(), n, n, 64, n_wg=9, l2_bytes=100 * 256)
    np.testing.assert_allclose(colblock.plan_to_dense(p), op.to_dense().double().numpy(), rtol=0, atol=1e-6)
    assert colblock.build_colblock_plan(rowptr.numpy(), c.numpy(), v.numpy(), n, 1 << 22, 64) is None


def test_row_ranges_balance_edges():
    op, p = _plan(5000, 40, n_wg=32)
    rowptr = op.csr()[0].numpy()
    b = p.wg_row0.numpy()
    per = np.diff(rowptr[b])
    assert per.max() - per.min() <= 2 * 40


# --------------------------------------------------------------------------------------------- GPU
def dense_ref(op, x):
    return torch.einsum("ij,tjf->tif", op.to_dense().double(), x.double().cpu()).float()


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,feat,t", [(3000, 40, 64, 5), (1200, 100, 128, 3), (20000, 30, 64, 4), (64, 9, 64, 7)])
def test_colblock_matches_dense_product(n, k, feat, t):
    torch.manual_seed(n + k)
    ei, ew = synthetic.random_graph(n, k, seed=n)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.randn(t, n, feat)
    y = torch.full((t, n, feat), float("nan"), device="cuda")
    op.propagate(x.cuda(), y, force="colblock")
    assert op.last_kernel == "spmm_colblock"
    ref = dense_ref(op, x) if n <= 3000 else None
    y2 = torch.empty_like(y)
    op.propagate(x.cuda(), y2, force="csr")
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)
    if ref is not None:
        assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_colblock_ragged_rows_strided_slots_and_auto_choice():
    """Rows without edges, one very long row, hop k reading slot k-1 and writing slot k of the same
    buffer; and the automatic choice: a random graph whose time step exceeds an L2 takes the
    column-blocked kernel, a small one stays with the generic CSR kernel."""
    torch.manual_seed(3)
    n, t, d, p = 700, 6, 64, 3
    rng = np.random.default_rng(1)
    deg = rng.integers(0, 60, n); deg[::5] = 0; deg[11] = 650
    row = np.repeat(np.arange(n), deg)
    col = rng.integers(0, n, row.size)
    op = graph.ShiftOperator.from_edges(torch.from_numpy(np.stack([col, row])), torch.rand(row.size), n)
    buf = torch.randn(t, n, p * d, device="cuda")
    out = buf.clone()
    for k in range(1, p):
        op.propagate(out[:, :, (k - 1) * d:k * d], out[:, :, k * d:(k + 1) * d], force="colblock")
    for k in range(1, p):
        ref = dense_ref(op, out[:, :, (k - 1) * d:k * d])
        assert torch.allclose(out[:, :, k * d:(k + 1) * d].cpu(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(ref.abs().max())))
    assert torch.equal(out[:, :, :d], buf[:, :, :d])
    # automatic choice
    big_n = 16000                                            # 16000 x 256 B = 4 MB per step > L2 budget
    ei, ew = synthetic.random_graph(big_n, 100, seed=4)
    big = graph.ShiftOperator.from_edges(ei, ew, big_n)
    x = torch.randn(4, big_n, 64, device="cuda")
    y = torch.empty_like(x)
    big.propagate(x, y)
    assert big.last_kernel == "spmm_colblock"
    y2 = torch.empty_like(x)
    big.propagate(x, y2, force="csr")
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)
    ei, ew = synthetic.random_graph(500, 20, seed=5)
    small = graph.ShiftOperator.from_edges(ei, ew, 500)
    xs = torch.randn(4, 500, 64, device="cuda")
    small.propagate(xs, torch.empty_like(xs))
    assert small.last_kernel != "spmm_colblock"
    with pytest.raises(NotImplementedError):
        small.propagate(torch.randn(4, 500, 48, device="cuda"), torch.empty(4, 500, 48, device="cuda"), force="colblock")


@pytest.mark.gpu
def test_colblock_nan_in_one_source_row_stays_in_the_rows_that_reference_it():
    """Padding entries read a real column with weight 0; 0 * NaN must not leak into a neighbouring
    row's sum (they end a run of a spare LDS row).  Also: more workgroups than CUs (no rendezvous) and
    two feature tiles."""
    torch.manual_seed(8)
    n, t, feat = 12000, 3, 128
    ei, ew = synthetic.random_graph(n, 40, seed=11)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    plan = op.colblock_plan(feat, torch.device("cuda"))
    assert plan.n_blocks >= 2
    bad = [0, plan.cols_per_block, n - 1]                       # first column of two blocks (what padding reads) and the last one
    x = torch.randn(t, n, feat)
    x[:, bad] = float("nan")
    y = torch.empty(t, n, feat, device="cuda")
    op.propagate(x.cuda(), y, force="colblock")
    rowptr, col, _ = op.csr()
    hit = torch.isin(col, torch.tensor(bad))
    touched = torch.zeros(n, dtype=torch.bool)
    touched[torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])[hit]] = True
    got = torch.isnan(y).any(2).any(0).cpu()
    assert torch.equal(got, touched), (int(got.sum()), int(touched.sum()))
    x0 = torch.nan_to_num(x, nan=0.0)
    y0 = torch.empty_like(y)
    op.propagate(x0.cuda(), y0, force="colblock")
    ok = ~touched
    assert torch.allclose(y[:, ok.cuda()].cpu(), y0[:, ok.cuda()].cpu(), rtol=0, atol=0)
    # a plan with more workgroups than the chip has CUs: the per-step rendezvous is skipped, results unchanged
    from sgp_amd import colblock, hip
    rowptr, col, val = op.csr()
    many = colblock.build_colblock_plan(rowptr.numpy(), col.numpy(), val.numpy(), n, n, feat, n_wg=400,
                                        rows_cap=hip.load().sgp_spmm_colblock_rows_cap(),
                                        round_pad=hip.load().sgp_spmm_colblock_round_pad()).to(torch.device("cuda"))
    assert many.n_wg >= 300
    y1 = torch.empty_like(y0)
    hip.spmm_colblock(many, x0.cuda(), y1)
    assert torch.allclose(y1, y0, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_colblock_partitioned_blocks_with_halo(world):
    """Local blocks of a node partition of a graph WITHOUT locality (round 4: halo source): the rows arrive as a
    second source in the [rows, T, D] layout of the exchange; chosen automatically once a time step's source rows
    exceed an L2."""
    from sgp_amd import partition
    torch.manual_seed(world)
    n, t, d = 24000, 5, 64
    ei, ew = synthetic.random_graph(n, 100, seed=3)               # 100 columns per row: no 16-row tile fits the LDS
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    x = torch.randn(t, n, d)
    whole = torch.empty(t, n, d, device="cuda")
    op.propagate(x.cuda(), whole, force="csr")
    bounds = partition.partition_bounds(n, world)
    for r in range(world):
        blk = partition.split_operator(op, bounds, r)
        assert blk.n_halo >= (n - blk.n_own) * 9 // 10               # no locality: the halo is (nearly) everything else
        xo = x[:, blk.lo:blk.hi].cuda().contiguous()
        recv = x[:, blk.halo_global].permute(1, 0, 2).contiguous().cuda()
        y = torch.full((t, blk.n_own, d), float("nan"), device="cuda")
        blk.op.propagate(xo, y, halo=recv.permute(1, 0, 2))
        assert blk.op.last_kernel == "spmm_colblock"
        assert torch.allclose(y, whole[:, blk.lo:blk.hi], rtol=1e-5, atol=1e-5)
        y2 = torch.empty_like(y)
        blk.op.propagate(xo, y2, force="csr", halo=recv.permute(1, 0, 2))
        assert torch.allclose(y, y2, rtol=1e-6, atol=1e-6)
