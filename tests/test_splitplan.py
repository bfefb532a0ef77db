"""Host plan of the split-fp16 hop (``sgp_amd/splitplan.py``; reference product: ``x = adj @ x``,
lib/sgp_preprocessing.py:200-203): the plan arrays, read back the way the kernel reads them, are the
operator (to the 2^-22 of the two fp16 pieces); the deal respects every limit the kernel assumes."""
import numpy as np
import pytest
import torch

from sgp_amd import graph, splitplan, synthetic

LIM = dict(waves=16, chunks=7, max_union=768)


def _unpack(adr):
    """[..., 64] packed a0 | a1 << 16 -> [..., 2, 64] byte addresses of the two transpose reads."""
    return np.stack([adr & 0xFFFF, (adr >> 16) & 0xFFFF], axis=-2)


def _op(ei, ew, n):
    return graph.ShiftOperator.from_edges(ei, ew, n)


def _plan(op, **kw):
    return splitplan.build_split_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), op.num_nodes,
                                      op.num_cols, **{**LIM, **kw})


def _check_limits(plan, n_rows, waves=16, chunks=7, max_union=768):
    hdr = plan.hdr.numpy()
    cnt, union = hdr[:, waves:2 * waves], hdr[:, 2 * waves]
    assert cnt.max() <= 16 and cnt.min() >= 0 and union.max() <= max_union
    assert int(cnt.sum()) == n_rows
    # every row sits in exactly one slot; a wave's slots are filled from 0
    rowid = plan.rowid.numpy()
    assert sorted(rowid[rowid >= 0].tolist()) == list(range(n_rows))
    assert ((rowid >= 0).sum(2) == cnt).all()
    assert ((rowid >= 0) == (np.arange(rowid.shape[2]) < cnt[:, :, None])).all()
    ucol = plan.ucol.numpy()
    for t in range(plan.n_tiles):
        u = int(union[t])
        assert (ucol[t, :u] >= 0).all() and (ucol[t, u:] == -1).all()
        assert len(np.unique(ucol[t, :u])) == u
    adr = _unpack(plan.adr.numpy())
    srow = (adr // 512) * 8 + (adr % 512) // 32
    assert (srow < np.maximum(union, 1)[:, None, None, None, None]).all()     # every address is a staged row
    used = cnt > 0
    assert ((adr[used] % 32) // 8 == (np.arange(64) & 3)).all()
    # every filled slot carries the inverse of a power-of-two row scale, empty slots 0
    rinv = plan.rinv.numpy()
    assert ((rinv > 0) == (rowid >= 0)).all()
    m, _ = np.frexp(rinv[rinv > 0])
    assert (m == 0.5).all()


@pytest.mark.parametrize("n,k", [(700, 20), (500, 100), (300, 7), (1300, 60)])
def test_plan_is_the_operator(n, k):
    ei, ew, _ = synthetic.knn_graph(n, k, seed=3)
    op = _op(ei, ew, n)
    plan = _plan(op)
    assert plan is not None
    _check_limits(plan, n)
    dense = op.to_dense().numpy().astype(np.float64)
    got = splitplan.plan_matrix(plan, n, n)
    assert np.abs(got - dense).max() <= 2.0 ** -21 * np.abs(dense).max()
    assert abs(plan.norm_inf - np.abs(dense).sum(1).max()) < 1e-6


def test_duplicates_empty_rows_and_ragged_degrees():
    rng = np.random.default_rng(0)
    n = 400
    deg = rng.integers(0, 50, n)
    deg[::9] = 0
    tgt = np.repeat(np.arange(n), deg)
    src = np.clip(tgt + rng.integers(-30, 31, tgt.size), 0, n - 1)           # duplicates are likely
    ei = torch.from_numpy(np.stack([src, tgt]))
    ew = torch.from_numpy(rng.random(tgt.size).astype(np.float32) + 0.1)
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    order = np.argsort(tgt, kind="stable")
    plan = splitplan.build_split_plan(rowptr, src[order], ew.numpy()[order], n, n, **LIM)   # un-coalesced CSR
    assert plan is not None
    _check_limits(plan, n)
    dense = np.zeros((n, n))
    np.add.at(dense, (tgt, src), ew.numpy().astype(np.float64))
    got = splitplan.plan_matrix(plan, n, n)
    assert np.abs(got - dense).max() <= 2.0 ** -20 * dense.max()


def test_rows_beyond_a_waves_budget_have_no_plan():
    n = 600
    src = np.concatenate([np.arange(n), np.arange(300)])                      # row 5 touches 300 columns (7 chunks hold 224)
    tgt = np.concatenate([np.arange(n), np.full(300, 5)])
    op = _op(torch.from_numpy(np.stack([src, tgt])), None, n)
    assert _plan(op) is None
    assert _plan(op, chunks=10, max_union=1024) is not None


def test_long_rows_are_cut_into_passes():
    """Rows beyond a wave's column budget: ``build_split_passes`` cuts every 16-row group's column union into
    segments; the passes add up to the operator, the first one holds every row, the later ones accumulate."""
    n = 1200
    ei, ew, _ = synthetic.threshold_graph(n, 420, seed=2)
    op = _op(ei, ew, n)
    assert op.max_degree() > 224 and _plan(op) is None
    passes = splitplan.build_split_passes(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n, n, **LIM)
    assert passes is not None and len(passes) >= 3
    assert not passes[0].accumulate and all(p.accumulate for p in passes[1:])
    assert sorted(passes[0].rowid.numpy()[passes[0].rowid.numpy() >= 0].tolist()) == list(range(n))
    assert passes[0].stats["rows_per_wave"] > 12
    dense = op.to_dense().numpy().astype(np.float64)
    got = sum(splitplan.plan_matrix(p, n, n) for p in passes)
    assert np.abs(got - dense).max() <= 2.0 ** -21 * np.abs(dense).max()
    # an operator without long rows is one pass, the plan build_split_plan makes
    ei, ew, _ = synthetic.knn_graph(700, 20, seed=3)
    one = splitplan.build_split_passes(*( _op(ei, ew, 700).rowptr.numpy(), _op(ei, ew, 700).col.numpy(), _op(ei, ew, 700).val.numpy()), 700, 700, **LIM)
    assert len(one) == 1 and not one[0].accumulate


def test_small_budgets_cut_waves_and_tiles():
    ei, ew, _ = synthetic.knn_graph(900, 40, seed=5)
    op = _op(ei, ew, 900)
    plan = _plan(op, chunks=3, max_union=256)
    assert plan is not None and plan.n_tiles > 4
    _check_limits(plan, 900, chunks=3, max_union=256)
    dense = op.to_dense().numpy().astype(np.float64)
    assert np.abs(splitplan.plan_matrix(plan, 900, 900) - dense).max() <= 2.0 ** -21 * dense.max()


def test_split_fp16_pieces():
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.standard_normal(4096) * 1000, rng.standard_normal(4096) * 1e-3,
                        [0.0, 16384.0, -16384.0, 2.0 ** -15, 6e-8]]).astype(np.float32)
    hi, lo = splitplan.split_fp16(v)
    back = hi.astype(np.float64) + lo.astype(np.float64)
    err = np.abs(back - v.astype(np.float64))
    assert (err <= np.maximum(np.abs(v) * 2.0 ** -21, 2.0 ** -25)).all()
    big = np.abs(v) >= 2.0 ** -14
    assert (np.abs(hi.astype(np.float32))[big] <= np.abs(v)[big]).all()      # truncated towards zero


def test_locality_order_serves_scrambled_numberings():
    """Scrambled node labels: dealt in the given numbering a wave's 16 rows share nothing (few rows per wave,
    many staged rows per result row); dealt in a locality order of the graph the plan is as good as on the
    ordered graph -- and still addresses rows and columns by their original ids."""
    n = 3000
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=6)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    op = _op(perm[ei], ew, n)
    plain = _plan(op)
    order = graph.locality_order(op.rowptr.numpy(), op.col.numpy(), n)
    plan = _plan(op, order=order)
    assert plan.stats["rows_per_wave"] > 2 * plain.stats["rows_per_wave"]
    assert plan.stats["staged_per_row"] < 0.5 * plain.stats["staged_per_row"]
    _check_limits(plan, n)
    dense = op.to_dense().numpy().astype(np.float64)
    assert np.abs(splitplan.plan_matrix(plan, n, n) - dense).max() <= 2.0 ** -21 * dense.max()


def test_other_build_shapes_and_row_scales():
    """Another build shape of the kernel (12 waves, 8 chunks), and rows of very different magnitude: every row is
    scaled by its own power of two, so a row of weights ~1e-9 beside rows of weights ~1 keeps its 22 bits."""
    ei, ew, _ = synthetic.knn_graph(900, 40, seed=5)
    op = _op(ei, ew, 900)
    plan = _plan(op, waves=12, chunks=8)
    assert plan is not None and plan.rowid.shape[1:] == (12, 16) and plan.afr.shape[1:4] == (12, 8, 2)
    hdr = plan.hdr.numpy()
    assert hdr[:, 12:24].max() <= 16 and int(hdr[:, 12:24].sum()) == 900 and hdr[:, 24].max() <= 768
    dense = op.to_dense().numpy().astype(np.float64)
    assert np.abs(splitplan.plan_matrix(plan, 900, 900) - dense).max() <= 2.0 ** -21 * dense.max()
    rowscale = np.where(np.arange(900) % 3 == 0, 1e-9, np.where(np.arange(900) % 3 == 1, 1.0, 1e7))
    rows = np.repeat(np.arange(900), np.diff(op.rowptr.numpy()))
    val = (op.val.numpy().astype(np.float64) * rowscale[rows]).astype(np.float32)
    plan = splitplan.build_split_plan(op.rowptr.numpy(), op.col.numpy(), val, 900, 900, **LIM)
    _check_limits(plan, 900)
    dense = np.zeros((900, 900))
    np.add.at(dense, (rows, op.col.numpy()), val.astype(np.float64))
    got = splitplan.plan_matrix(plan, 900, 900)
    assert (np.abs(got - dense).max(1) <= 2.0 ** -21 * np.abs(dense).max(1)).all()       # row by row


def test_k_slots_keep_the_rows_of_a_transpose_read_on_different_banks():
    """A ``ds_read_b64_tr_b16`` serves lanes 0-31 and 32-63 in one LDS cycle each: 8 staged rows, each covering
    the 8 banks ``(s & 7) * 8 ..`` of its position s.  The planner deals a chunk's rows to the four 8-row sets by
    ``s & 7`` (``_bank_aware_slots``): on a geometric graph the reads must stay close to one cycle per lane group
    (columns in sorted order: ~1.8), and the plan still is the operator."""
    n = 6000
    ei, ew, _ = synthetic.knn_graph(n, 60, seed=3)
    op = _op(ei, ew, n)
    plan = _plan(op, waves=16, chunks=8)
    adr, hdr = _unpack(plan.adr.numpy()), plan.hdr.numpy()
    a = adr[hdr[:, 16:32] > 0]                                 # [wave, chunk, read, lane]
    s = (a // 512) * 8 + (a % 512) // 32
    cycles = []
    for half in range(2):
        rows = s[..., np.arange(half * 32, half * 32 + 32, 4)]          # the 8 rows one LDS cycle fetches
        per_bank = np.zeros(rows.shape[:-1], dtype=np.int64)
        for q in range(8):
            v = np.sort(np.where((rows & 7) == q, rows, -1), axis=-1)
            distinct = ((v[..., 1:] != v[..., :-1]) & (v[..., 1:] >= 0)).sum(-1) + (v[..., 0] >= 0)
            per_bank = np.maximum(per_bank, distinct)
        cycles.append(per_bank)
    mean_cycles = float(np.mean(cycles))
    assert mean_cycles < 1.35, mean_cycles
    dense = torch.sparse_coo_tensor(torch.stack([torch.repeat_interleave(torch.arange(n), op.rowptr[1:] - op.rowptr[:-1]),
                                                 op.col.long()]), op.val.double(), (n, n)).to_dense().numpy()
    got = splitplan.plan_matrix(plan, n, n)
    assert np.abs(got - dense).max() <= 2.0 ** -21 * np.abs(dense).max()
