"""The native split-hop planner (csrc/plan_split.hip: sgp_split_plan_deal / _fill) against its numpy restatement
(splitplan.build_split_plan_numpy): same arrays, byte for byte, on graphs that exercise every branch -- ragged rows,
empty rows, duplicate entries, rows past a wave's budget, a dealing order, tiles of empty rows, tiny and huge
weights.  CPU only: the planner is host code."""
import time

import numpy as np
import pytest
import torch

from sgp_amd import splitplan, synthetic
from sgp_amd.graph import ShiftOperator

LIM = dict(waves=16, chunks=7, max_union=768, rows_per_wave=16)


def csr_of(ei, ew, n):
    op = ShiftOperator.from_edges(ei, ew, n)
    return op.rowptr.numpy().astype(np.int64), op.col.numpy().astype(np.int64), op.val.numpy()


def same_plan(a, b):
    assert (a is None) == (b is None)
    if a is None:
        return
    for name in ("hdr", "rowid", "ucol", "adr", "rinv"):
        x, y = getattr(a, name).numpy(), getattr(b, name).numpy()
        assert x.shape == y.shape and np.array_equal(x, y), name
    assert np.array_equal(a.afr.view(torch.int16).numpy(), b.afr.view(torch.int16).numpy()), "afr"
    assert a.n_tiles == b.n_tiles and a.norm_inf == pytest.approx(b.norm_inf, rel=1e-12)
    for k, v in a.stats.items():
        assert v == pytest.approx(b.stats[k], rel=1e-12), k


def random_csr(n, n_cols, deg_lo, deg_hi, seed, dup=False, empty_block=None, scale=None):
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r in range(n):
        if empty_block is not None and empty_block[0] <= r < empty_block[1]:
            continue
        d = int(rng.integers(deg_lo, deg_hi + 1))
        base = int(r * n_cols / n)
        c = (base + rng.integers(-3 * deg_hi, 3 * deg_hi + 1, d)) % n_cols
        if not dup:
            c = np.unique(c)
        rows.append(np.full(c.size, r))
        cols.append(np.sort(c))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    val = rng.standard_normal(rows.size).astype(np.float32)
    if scale is not None:
        val *= np.exp2(rng.integers(scale[0], scale[1], rows.size)).astype(np.float32)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), cols.astype(np.int64), val


CASES = {
    "knn": lambda: csr_of(*synthetic.knn_graph(3000, 40, seed=3)[:2], 3000),
    "knn_dense": lambda: csr_of(*synthetic.knn_graph(2500, 100, seed=4)[:2], 2500),
    "ragged": lambda: random_csr(1500, 1500, 0, 60, 5),
    "duplicates": lambda: random_csr(900, 900, 5, 50, 6, dup=True),
    "empty_tile": lambda: random_csr(1400, 1400, 3, 30, 7, empty_block=(300, 700)),
    "rectangular": lambda: random_csr(800, 2000, 10, 40, 8),
    "scales": lambda: random_csr(700, 700, 10, 40, 9, scale=(-140, 100)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_native_planner_equals_numpy_planner(name):
    rowptr, col, val = CASES[name]()
    n_rows, n_cols = rowptr.size - 1, int(col.max()) + 1
    with np.errstate(over="ignore", under="ignore"):
        ref = splitplan.build_split_plan_numpy(rowptr, col, val, n_rows, n_cols, **LIM)
        for threads in (1, 3):
            same_plan(splitplan.build_split_plan(rowptr, col, val, n_rows, n_cols, threads=threads, **LIM), ref)
    assert ref is not None
    # the plan encodes the operator: dense check on the smallest case
    if name == "ragged":
        dense = np.zeros((n_rows, n_cols))
        np.add.at(dense, (np.repeat(np.arange(n_rows), np.diff(rowptr)), col), val.astype(np.float64))
        got = splitplan.plan_matrix(splitplan.build_split_plan(rowptr, col, val, n_rows, n_cols, **LIM), n_rows, n_cols)
        assert np.abs(got - dense).max() <= 2.0 ** -21 * np.abs(dense).max()


def test_native_planner_with_an_order_and_a_row_subset():
    rowptr, col, val = CASES["knn"]()
    n = rowptr.size - 1
    rng = np.random.default_rng(0)
    for order in (rng.permutation(n), np.sort(rng.choice(n, n // 3, replace=False))):
        same_plan(splitplan.build_split_plan(rowptr, col, val, n, n, order=order, **LIM),
                  splitplan.build_split_plan_numpy(rowptr, col, val, n, n, order=order, **LIM))


def test_native_planner_rows_beyond_the_budget_and_bad_input():
    rowptr, col, val = random_csr(400, 3000, 230, 400, 11)
    n = rowptr.size - 1
    assert splitplan.build_split_plan(rowptr, col, val, n, 3000, **LIM) is None
    assert splitplan.build_split_plan_numpy(rowptr, col, val, n, 3000, **LIM) is None
    passes = splitplan.build_split_passes(rowptr, col, val, n, 3000, **LIM)
    assert passes is not None and len(passes) >= 2 and all(p.accumulate == (i > 0) for i, p in enumerate(passes))
    dense = np.zeros((n, 3000))
    np.add.at(dense, (np.repeat(np.arange(n), np.diff(rowptr)), col), val.astype(np.float64))
    got = sum(splitplan.plan_matrix(p, n, 3000) for p in passes)
    assert np.abs(got - dense).max() <= 2.0 ** -21 * np.abs(dense).max()
    bad = val.copy()
    bad[5] = np.inf
    assert splitplan.build_split_plan(rowptr[:50], col, bad, 49, 3000, **LIM) is None
    with pytest.raises(ValueError):
        splitplan.build_split_plan(rowptr, col, val, n, 100, **LIM)
    with pytest.raises(RuntimeError, match="twice"):
        splitplan.build_split_plan(rowptr[:11], col % 60, val, 10, 60, order=np.array([1, 2, 2]), **LIM)


def test_native_planner_is_fast_on_a_target_shaped_graph():
    """Round-5 review: 22.7 s of numpy for the target graph (N = 100 000) in front of an 80 ms pass.  Here a fifth of it
    (N = 20 000, 100-NN, same structure): the native planner must stay under 2 s on this box's cores (measured: ~0.4 s
    on 8 cores; the numpy planner takes ~5 s)."""
    n = 20000
    ei, ew, _ = synthetic.knn_graph(n, 100, seed=1)
    rowptr, col, val = csr_of(ei, ew, n)
    t0 = time.time()
    plan = splitplan.build_split_plan(rowptr, col, val, n, n, **LIM)
    dt = time.time() - t0
    assert plan is not None and plan.stats["rows_per_wave"] > 12 and plan.stats["staged_per_row"] < 4.5
    assert dt < 2.0, f"native split planner took {dt:.2f} s for N = {n}"


def test_plan_cache_round_trip(tmp_path, monkeypatch):
    """SGP_AMD_CACHE: the second operator built from the same edges loads its plans instead of planning (split, tile and
    mix plans; a changed weight is another key; a truncated file is rebuilt)."""
    from sgp_amd import plancache
    monkeypatch.setenv("SGP_AMD_CACHE", str(tmp_path))
    n = 2600
    ei, ew, _ = synthetic.knn_graph(n, 30, seed=2)
    cpu = torch.device("cpu")
    before = dict(plancache.stats)
    op = ShiftOperator.from_edges(ei, ew, n)
    a = op.split_plan(cpu)
    ta, ma = op.tile_plan(64, cpu, tall=False), op.mix_plan(64, cpu, strict=False)
    assert plancache.stats["stores"] - before["stores"] == 3 and plancache.stats["hits"] == before["hits"]
    op2 = ShiftOperator.from_edges(ei, ew, n)
    t0 = time.time()
    b = op2.split_plan(cpu)
    tb, mb = op2.tile_plan(64, cpu, tall=False), op2.mix_plan(64, cpu, strict=False)
    dt = time.time() - t0
    assert plancache.stats["hits"] - before["hits"] == 3 and dt < 1.0
    same_plan(a, b)
    assert torch.equal(ta.ucol, tb.ucol) and torch.equal(ta.pipe["gw"], tb.pipe["gw"]) and ta.tile_rows == tb.tile_rows
    assert (ma is None) == (mb is None) and (ma is None or torch.equal(ma.gw, mb.gw))
    ew2 = ew.clone()
    ew2[0] *= 2
    ShiftOperator.from_edges(ei, ew2, n).split_plan(cpu)
    assert plancache.stats["hits"] - before["hits"] == 3            # another operator: a miss
    for f in tmp_path.iterdir():
        if f.name.startswith("split-"):
            f.write_bytes(f.read_bytes()[:100])
    same_plan(ShiftOperator.from_edges(ei, ew, n).split_plan(cpu), a)


WIDE = dict(waves=8, chunks=14, max_union=768, rows_per_wave=16)


@pytest.mark.parametrize("name", ["knn_dense", "ragged", "duplicates"])
def test_native_planner_equals_numpy_planner_in_the_wide_geometry(name):
    """The wide form of the split hop (8 waves x 14 chunks: csrc/spmm_split_wide.hip) takes the same plan format."""
    rowptr, col, val = CASES[name]()
    n_rows, n_cols = rowptr.size - 1, int(col.max()) + 1
    with np.errstate(over="ignore", under="ignore"):
        ref = splitplan.build_split_plan_numpy(rowptr, col, val, n_rows, n_cols, **WIDE)
        got = splitplan.build_split_plan(rowptr, col, val, n_rows, n_cols, **WIDE)
    same_plan(got, ref)
    assert tuple(got.afr.shape[1:3]) == (8, 14)


def test_long_row_operators_are_planned_for_the_wide_kernel():
    """Rows of ~400 entries (beyond the standard form's 224 columns per wave): the operator's plan is a list of WIDE
    passes, fewer than the standard geometry needs, and encodes the operator."""
    n = 1400
    ei, ew, _ = synthetic.threshold_graph(n, 330, seed=4)
    op = ShiftOperator.from_edges(ei, ew, n)
    rowptr, col, val = op.rowptr.numpy().astype(np.int64), op.col.numpy().astype(np.int64), op.val.numpy()
    assert op.max_degree() > 224
    plan = op.split_plan(torch.device("cpu"))
    assert plan is not None and all(tuple(p.afr.shape[1:3]) == (8, 14) for p in plan)
    std = splitplan.build_split_passes(rowptr, col, val, n, n, **LIM)
    assert len(plan) < len(std)
    dense = np.zeros((n, n))
    np.add.at(dense, (np.repeat(np.arange(n), np.diff(rowptr)), col), val.astype(np.float64))
    got = sum(splitplan.plan_matrix(p, n, n) for p in plan)
    assert np.abs(got - dense).max() <= 2.0 ** -21 * np.abs(dense).max()


def test_prepare_builds_only_what_the_default_dispatch_runs(monkeypatch):
    """``ShiftOperator.prepare``: where the split hop is the default, the split plan and the device CSR -- not the exact
    kernels' tile / mix plans (round-5 review: 12.7 + 3.3 s on the target graph for kernels behind a predicate that is 0 on
    every shipped configuration); small graphs and SGP_TUNE=hop=exact / exact_plans=eager get the staged kernels' plans."""
    monkeypatch.delenv("SGP_TUNE", raising=False)
    cpu = torch.device("cpu")
    ei, ew, _ = synthetic.knn_graph(2600, 30, seed=2)
    op = ShiftOperator.from_edges(ei, ew, 2600)
    made = op.prepare(64, cpu)
    assert made[0] == "split" and made[1].startswith("csr") and len(made) == 2
    assert ("split", "cpu") in op._plans and (True, "cpu") not in op._plans
    monkeypatch.setenv("SGP_TUNE", "exact_plans=eager")
    assert ShiftOperator.from_edges(ei, ew, 2600).prepare(64, cpu)[:2] == ["split", "tile"]
    monkeypatch.setenv("SGP_TUNE", "hop=exact")
    assert ShiftOperator.from_edges(ei, ew, 2600).prepare(64, cpu)[0] == "tile"
    monkeypatch.delenv("SGP_TUNE")
    ei2, ew2 = synthetic.sparse_traffic_graph(325, 2369, seed=1)
    assert ShiftOperator.from_edges(ei2, ew2, 325).prepare(128, cpu) == ["tile"]        # traffic-sized: the tall-tile VALU kernel
    assert ShiftOperator.from_edges(ei, ew, 2600).prepare(20, cpu) == ["csr"]             # a width no staged kernel serves


def test_plan_cache_holds_the_pass_lists_of_long_row_operators(tmp_path, monkeypatch):
    from sgp_amd import plancache
    monkeypatch.setenv("SGP_AMD_CACHE", str(tmp_path))
    n = 1400
    ei, ew, _ = synthetic.threshold_graph(n, 330, seed=4)
    cpu = torch.device("cpu")
    a = ShiftOperator.from_edges(ei, ew, n).split_plan(cpu)
    hits = plancache.stats["hits"]
    b = ShiftOperator.from_edges(ei, ew, n).split_plan(cpu)
    assert plancache.stats["hits"] == hits + 1 and len(a) == len(b) >= 2
    for p, q in zip(a, b):
        same_plan(p, q)
        assert p.accumulate == q.accumulate
