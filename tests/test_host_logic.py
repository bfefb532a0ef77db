"""Host-side logic of the product (graph operators, tile plans, weight init, API surface,
harness) against the oracle and the golden vectors.  CPU-only."""
import argparse
import inspect
import os

import numpy as np
import pytest
import torch

import sgp_amd
from conftest import GOLDEN, golden_files
from oracle import sgp_oracle as O
from sgp_amd import graph, synthetic
from sgp_amd.sgp_preprocessing import spatial_operators


def load(name):
    return np.load(os.path.join(GOLDEN, name))


FLAGS = [dict(), dict(bidirectional=True), dict(add_self_loops=True),
         dict(remove_self_loops=True), dict(undirected=True),
         dict(undirected=True, add_self_loops=True),
         dict(bidirectional=True, remove_self_loops=True)]


@pytest.mark.parametrize("flags", FLAGS)
def test_shift_operators_match_oracle(flags):
    z = load("g1_spatial_plain_k1.npz")
    n = z["x"].shape[1]
    for ew in (torch.from_numpy(z["edge_weight"]), None):
        ops = spatial_operators(torch.from_numpy(z["edge_index"]), ew, n, **flags)
        ref = O.shift_operators_dense(z["edge_index"], None if ew is None else ew, n, **flags)
        assert len(ops) == len(ref)
        for a, b in zip(ops, ref):
            assert torch.allclose(a.to_dense().double(), b, rtol=1e-6, atol=1e-7)
            rp = a.rowptr.long()
            assert rp[0] == 0 and rp[-1] == a.nnz() and bool((rp[1:] >= rp[:-1]).all())


def test_preprocess_adj_variants():
    z = load("g1_spatial_plain_k1.npz")
    n = z["x"].shape[1]
    ei = torch.from_numpy(z["edge_index"])
    ew = torch.from_numpy(z["edge_weight"])
    a = sgp_amd.preprocess_adj(ei, ew, num_nodes=n)                 # set_diag default True
    ref = O.normalize_dense(O.dense_adjacency(ei, ew, n), set_diag=True)
    assert torch.allclose(a.to_dense().double(), ref, atol=1e-7)
    b = sgp_amd.preprocess_adj(z["edge_index"], z["edge_weight"], num_nodes=n, set_diag=False)
    ref = O.normalize_dense(O.dense_adjacency(ei, ew, n))
    assert torch.allclose(b.to_dense().double(), ref, atol=1e-7)
    # a ready sparse object (row = target layout) is accepted, like SparseTensor at :83-84
    c = sgp_amd.preprocess_adj(b, set_diag=False)
    ref2 = O.normalize_dense(ref)
    assert torch.allclose(c.to_dense().double(), ref2, atol=1e-6)
    with pytest.raises(RuntimeError):
        sgp_amd.preprocess_adj([[0, 1], [1, 0]])
    with pytest.raises(AssertionError):
        spatial_operators(ei, ew, n, undirected=True, bidirectional=True)


def test_zero_degree_rows_and_isolated_nodes():
    ei = torch.tensor([[0, 1, 1], [1, 0, 0]])
    op = graph.ShiftOperator.from_edges(ei, torch.tensor([1., 2., 3.]), 4)
    d = op.to_dense()
    assert torch.allclose(d[0], torch.tensor([0., 1., 0., 0.]))
    assert float(d[2:].abs().sum()) == 0.0
    op = graph.ShiftOperator.from_edges(torch.zeros(2, 0, dtype=torch.long), None, 3)
    assert op.nnz() == 0 and op.rowptr.tolist() == [0, 0, 0, 0]


def _check_plan(op, plan):
    rowptr = op.rowptr.numpy().astype(np.int64)
    col, val = op.col.numpy(), op.val.numpy()
    trow, uptr, ucol = plan.trow.numpy(), plan.uptr.numpy(), plan.ucol.numpy()
    erow = plan.erow.numpy()
    ecol = plan.ecol.numpy().view(np.uint16)
    assert trow[0] == 0 and trow[-1] == op.num_nodes and (np.diff(trow) > 0).all()
    assert np.diff(trow).max() == plan.tile_rows and len(trow) == plan.n_tiles + 1
    assert np.diff(uptr).max() == plan.max_union
    assert (np.diff(erow) % 16 == 0).all() and np.diff(erow).max() == plan.max_row_edges
    for k in range(plan.n_tiles):
        u = ucol[uptr[k]:uptr[k + 1]]
        assert (np.diff(u) > 0).all()                              # sorted, distinct
        for r in range(trow[k], trow[k + 1]):
            d = rowptr[r + 1] - rowptr[r]
            e0 = erow[r]
            assert (u[ecol[e0:e0 + d]] == col[rowptr[r]:rowptr[r + 1]]).all()
            assert (plan.eval.numpy()[e0:e0 + d] == val[rowptr[r]:rowptr[r + 1]]).all()
            assert (plan.eval.numpy()[e0 + d:erow[r + 1]] == 0).all()
            assert (ecol[e0 + d:erow[r + 1]] == 0).all()


LIMITS = dict(max_union=512, max_tile_rows=128, max_row_edges=128)


def test_tile_plan_knn_graph():
    ei, ew, _ = synthetic.knn_graph(3000, 40, seed=5)
    op = graph.ShiftOperator.from_edges(ei, ew, 3000)
    plan = graph.build_tile_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), 3000, **LIMITS)
    assert plan is not None and plan.tile_rows == 64 and plan.max_row_edges == 48
    _check_plan(op, plan)


def test_tile_plan_splits_oversized_tiles_and_gives_up_on_random_graphs():
    ei, ew, _ = synthetic.knn_graph(3000, 40, seed=5)
    op = graph.ShiftOperator.from_edges(ei, ew, 3000)
    small = graph.build_tile_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), 3000,
                                  max_union=160, max_tile_rows=128, max_row_edges=128)
    assert small is not None and small.max_union <= 160
    assert np.diff(small.trow.numpy()).min() < small.tile_rows      # some tiles were halved
    _check_plan(op, small)
    ei, ew = synthetic.random_graph(4000, 100, seed=5)
    op = graph.ShiftOperator.from_edges(ei, ew, 4000)
    assert graph.build_tile_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), 4000,
                                 **LIMITS) is None
    # rows longer than the kernel's register budget -> generic kernel
    ei, ew = synthetic.random_graph(300, 200, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, 300)
    assert graph.build_tile_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), 300,
                                 **LIMITS) is None


def test_reordered_plan_for_graphs_without_locality_in_the_numbering():
    """A geometric k-NN graph whose node labels are scrambled: the row numbering has no locality
    (16-row tiles at best), the landmark order restores 64-row tiles, and the reordered plan --
    expressed in the ORIGINAL ids through ucol / rowmap -- is exactly the same operator."""
    n = 4000
    ei, ew, _ = synthetic.knn_graph(n, 24, seed=3)
    perm = np.random.default_rng(1).permutation(n)
    op = graph.ShiftOperator.from_edges(torch.from_numpy(perm[ei.numpy()]), ew, n)
    args = (op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n)
    plain = graph.build_tile_plan(*args, **LIMITS)
    assert plain is None or plain.tile_rows < 32
    order = graph.locality_order(op.rowptr.numpy(), op.col.numpy(), n)
    assert sorted(order.tolist()) == list(range(n))
    plan = graph.build_reordered_plan(*args, order, **LIMITS)
    assert plan is not None and plan.reordered and plan.tile_rows == 64 and plan.max_union <= 448
    ps = plan.pipe
    gptr, gw, gidx = ps["gptr"].numpy(), ps["gw"].numpy(), ps["gidx"].numpy()
    uptr, ucol, rowmap = ps["uptr"].numpy(), ps["ucol"].numpy(), ps["rowmap"].numpy()
    a = np.zeros((n, n))
    for tile in range(plan.n_tiles):
        for g in range(16):
            for qd in range(gptr[(tile * 16 + g) * 2], gptr[(tile * 16 + g) * 2 + 2]):
                for cls in range(4):
                    for sup in range(4):
                        c = ucol[uptr[tile] + gidx[qd, cls, sup] // 256]
                        for i in range(4):
                            if gw[qd, cls, sup, i] != 0:
                                a[rowmap[tile * 64 + g * 4 + i], c] += gw[qd, cls, sup, i]
    assert np.array_equal(a.astype(np.float32), op.to_dense().numpy())
    # every row is written exactly once
    rows = rowmap[rowmap >= 0]
    assert sorted(rows.tolist()) == list(range(n))


def test_tile_plan_small_sparse_graph():
    ei, ew = synthetic.sparse_traffic_graph(207, 1515, seed=3)
    op = graph.ShiftOperator.from_edges(ei, ew, 207)
    plan = graph.build_tile_plan(op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), 207, **LIMITS)
    assert plan is not None
    _check_plan(op, plan)


@pytest.mark.parametrize("n,e,rows,tiles", [(207, 1515, 207, 1), (325, 2369, 325, 1), (700, 4900, 64, None)])
def test_operator_picks_tall_tiles_for_small_sparse_graphs(n, e, rows, tiles):
    """Traffic-sized sparse graphs are planned as ONE tile (the VALU kernel stages the whole slab once
    per step); the 64-row plan stays available for the kernels that need it; a 700-node random graph
    has no tall tile that fits the stage."""
    ei, ew = synthetic.sparse_traffic_graph(n, e, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    limits = dict(max_union=448, max_tile_rows=64, max_row_edges=128)     # what hip.tiled_limits gives
    plan = op.tile_plan(64, torch.device("cpu"), limits=limits)
    assert plan is not None and plan.tile_rows == rows
    if tiles is not None:
        assert plan.n_tiles == tiles and plan.gw is None
        _check_plan(op, plan)
        # LDS budget of sgp_spmm_tiled_f32's tall form: staged rows (whole passes of 64) + 6 B per edge slot
        rpg = 4 if plan.tile_rows <= 256 else 6
        nb = 1 if plan.max_row_edges <= 16 else 2
        assert (plan.max_union + 63) // 64 * 64 * 256 + rpg * 64 * nb * 16 * 6 <= 160 * 1024
    std = op.tile_plan(64, torch.device("cpu"), limits=limits, tall=False)
    assert std.tile_rows <= 64 and std.gw is not None
    # the FIRST call of a fresh operator asking for the 64-row plan gets it too (round-2 advisor finding:
    # it used to return the tall plan it had just built)
    op2 = graph.ShiftOperator.from_edges(ei, ew, n)
    first = op2.tile_plan(64, torch.device("cpu"), limits=limits, tall=False)
    assert first.tile_rows <= 64 and first.gw is not None
    assert op2.tile_plan(64, torch.device("cpu"), limits=limits).tile_rows == rows


# ------------------------------------------------------------------ weights / API surface
@pytest.mark.parametrize("name", [f for f in golden_files("g4_seed") if "gesn" not in f])
def test_seed_reproduces_reference_weights(name):
    z = load(name)
    f, r, L, a, rho, dens, scale = z["cfg"]
    torch.manual_seed(int(z["seed"]))
    res = sgp_amd.Reservoir(input_size=int(f), hidden_size=int(r), num_layers=int(L),
                            leaking_rate=a, spectral_radius=rho, density=dens,
                            input_scaling=scale, alpha_decay=True)
    after = torch.rand(4)
    ref = O.layers_from_npz(z)
    for layer, g in zip(res.reservoir_layers, ref):
        assert torch.equal(layer.w_ih.data, g["w_ih"])
        assert torch.equal(layer.w_hh.data, g["w_hh"])
        assert torch.equal(layer.b_ih.data, g["b_ih"])
        assert float(layer.alpha) == g["alpha"]
    assert torch.equal(after, torch.from_numpy(z["rng_after"]))


def test_constructor_signatures_match_reference():
    # names are API: filter_args routes CLI/YAML values by __init__ parameter name
    # (tsl/utils/parser_utils.py:69-81); lists from lib/nn/encoders/*.py and reservoir.py:85-95
    def params(cls):
        return list(inspect.signature(cls.__init__).parameters)[1:]
    assert params(sgp_amd.SGPEncoder) == [
        "input_size", "reservoir_size", "reservoir_layers", "leaking_rate", "spectral_radius",
        "density", "input_scaling", "receptive_field", "bidirectional", "alpha_decay",
        "global_attr", "add_self_loops", "undirected", "reservoir_activation"]
    assert params(sgp_amd.SGPTemporalEncoder) == [
        "input_size", "reservoir_size", "reservoir_layers", "leaking_rate", "spectral_radius",
        "density", "input_scaling", "alpha_decay", "reservoir_activation"]
    assert params(sgp_amd.SGPSpatialEncoder) == [
        "receptive_field", "bidirectional", "undirected", "global_attr", "add_self_loops"]
    assert params(sgp_amd.Reservoir) == [
        "input_size", "hidden_size", "input_scaling", "num_layers", "leaking_rate",
        "spectral_radius", "density", "activation", "bias", "alpha_decay"]
    d = inspect.signature(sgp_amd.SGPTemporalEncoder.__init__).parameters
    assert (d["reservoir_size"].default, d["density"].default, d["leaking_rate"].default) == (32, 0.7, 0.9)
    d = inspect.signature(sgp_amd.Reservoir.__init__).parameters
    assert d["density"].default == 0.9 and d["bias"].default is True
    d = inspect.signature(sgp_amd.sgp_spatial_embedding).parameters
    assert list(d) == ["x", "num_nodes", "edge_index", "edge_weight", "k", "undirected",
                       "add_self_loops", "remove_self_loops", "bidirectional",
                       "one_hot_encoding", "dropout_rate"]


def test_cli_flags_on_plain_argparse():
    p = sgp_amd.SGPEncoder.add_model_specific_args(argparse.ArgumentParser())
    a = p.parse_args([])
    assert (a.reservoir_size, a.reservoir_layers, a.receptive_field) == (32, 1, 1)
    assert (a.spectral_radius, a.leaking_rate, a.density, a.input_scaling) == (0.9, 0.9, 0.7, 1.)
    assert (a.bidirectional, a.undirected, a.add_self_loops, a.alpha_decay, a.global_attr) == \
        (False,) * 5
    assert a.reservoir_activation == "tanh"
    a = p.parse_args(["--bidirectional", "--global-attr", "true", "--reservoir-size", "64"])
    assert a.bidirectional is True and a.global_attr is True and a.reservoir_size == 64

    class TT(argparse.ArgumentParser):                 # test_tube-like parser
        def opt_list(self, *args, options=None, tunable=False, **kw):
            self.add_argument(*args, **kw)
    a = sgp_amd.SGPTemporalEncoder.add_model_specific_args(TT()).parse_args([])
    assert a.receptive_field == 1 and a.reservoir_size == 32


def test_activation_errors_match_reference():
    with pytest.raises(AssertionError):
        sgp_amd.Reservoir(3, 8, activation="gelu")
    with pytest.raises(ValueError):                    # tsl/nn/utils/utils.py:44
        sgp_amd.Reservoir(3, 8, activation="identity")
    layer = sgp_amd.ReservoirLayer(3, 8, 0.9, 0.9, bias=False)
    assert layer.b_ih is not None                      # reservoir.py:47 quirk
    assert sgp_amd.ReservoirLayer(3, 8, 0.9, 0.9, bias=None).b_ih is None


def test_alpha_decay_schedule():
    res = sgp_amd.Reservoir(3, 8, num_layers=8, leaking_rate=1.0, alpha_decay=True)
    got = [float(l.alpha) for l in res.reservoir_layers]
    assert got == O.alpha_schedule(1.0, 8, True)
    assert np.allclose(got, [1., .9, .8, .7, .6, .5, .4, .3])


class FakeDataset:
    def __init__(self, data, u, ei, ew):
        self._t = {"data": data, "u": u}
        self.exogenous = {"u": u}
        self.edge_index, self.edge_weight = ei, ew
        self.calls = []

    def get_tensors(self, keys, preprocess=False, cat_dim=None):
        self.calls.append(("get_tensors", list(keys), preprocess, cat_dim))
        ts = [self._t[k] if self._t[k].dim() == 3 else
              self._t[k][:, None].expand(-1, self._t["data"].shape[1], -1) for k in keys]
        return torch.cat(ts, cat_dim), None

    def add_exogenous(self, name, value, add_to_input_map=True):
        self.calls.append(("add_exogenous", name, add_to_input_map))
        self._t[name] = value

    def set_input_map(self, m):
        self.calls.append(("set_input_map", m))
        self.input_map = m


class StubEncoder:
    """Host-logic stand-in so the harness can be exercised without a GPU."""

    def __init__(self, input_size, **kw):
        self.input_size = input_size

    def __call__(self, x, edge_index, edge_weight):
        assert x.shape[-1] == self.input_size
        return x.sum(-1, keepdim=True)


@pytest.mark.parametrize("name", golden_files("g5_"))
def test_encode_dataset_harness_logic(name, tmp_path):
    z = load(name)
    ds = FakeDataset(torch.from_numpy(z["data"]), torch.from_numpy(z["u"]),
                     torch.from_numpy(z["edge_index"]), torch.from_numpy(z["edge_weight"]))
    enc_exo, keep_raw = bool(z["encode_exogenous"]), bool(z["keep_raw"])
    path = tmp_path / "enc.pt"
    out = sgp_amd.encode_dataset(ds, StubEncoder, dict(input_size=3 if enc_exo else 1),
                                 encode_exogenous=enc_exo, keep_raw=keep_raw, save_path=str(path))
    assert out is ds and len(ds.calls) == int(z["n_calls"])
    assert ds.calls[0] == ("get_tensors", list(z["get_tensors_keys"]), True, -1)
    assert ds.calls[1] == ("add_exogenous", "encoded_x", False)
    assert ds.input_map["x"] == list(z["input_map_x"])
    assert ds.input_map.get("u", []) == list(z["input_map_u"])
    assert torch.equal(torch.load(path), ds._t["encoded_x"])


@pytest.mark.parametrize("name", golden_files("g6_"))
def test_spatial_support_matches_reference(name):
    """sgp_spatial_support (host-side supports, lib/sgp_preprocessing.py:108-160) against the
    reference's output, quirks included; oracle restatement checked on the same vectors."""
    z = load(name)
    n = int(z["n"])
    ei = torch.from_numpy(z["edge_index"])
    ew = torch.from_numpy(z["edge_weight"]) if bool(z["has_weight"]) else None
    kw = {k: (int(z[k]) if k == "k" else bool(z[k])) for k in
          ("k", "undirected", "add_self_loops", "remove_self_loops", "bidirectional", "global_attr")
          if k in z.files}
    ref = torch.from_numpy(z["supports"])
    got = sgp_amd.sgp_spatial_support(ei, ew, num_nodes=n, **kw)
    ora = O.spatial_support_dense(ei, ew, n, **kw)
    assert len(got) == len(ora) == ref.shape[0]
    for g, o, r in zip(got, ora, ref):
        gd = g if torch.is_tensor(g) else g.to_dense()
        assert torch.allclose(gd, r, rtol=1e-5, atol=1e-6)
        assert torch.allclose(o.float(), r, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ DynGESN host side
def test_graph_esn_seed_reproduces_reference_weights():
    """graph_reservoir.py:140 -- GraphESN draws every layer twice; the seed still matches."""
    z = load("g4_seed_gesn.npz")
    torch.manual_seed(int(z["seed"]))
    res = sgp_amd.GraphESN(input_size=3, hidden_size=16, num_layers=2, density=.8,
                           alpha_decay=True)
    after = torch.rand(4)
    for layer, g in zip(res.rnn_cells, O.layers_from_npz(z)):
        assert torch.equal(layer.w_ih.data, g["w_ih"])
        assert torch.equal(layer.w_hh.data, g["w_hh"])
        assert torch.equal(layer.b_ih.data, g["b_ih"])
        assert float(layer.alpha) == g["alpha"]
    assert torch.equal(after, torch.from_numpy(z["rng_after"]))


@pytest.mark.parametrize("name", golden_files("g3_gesn"))
def test_gesn_operator_matches_oracle(name):
    from sgp_amd.nn.encoders.dyn_gesn_encoder import gesn_operator
    z = load(name)
    n = z["x"].shape[1]
    op = gesn_operator(z["edge_index"], torch.from_numpy(z["edge_weight"]), n)
    ref = O.gesn_operator_dense(z["edge_index"], torch.from_numpy(z["edge_weight"]), n)
    assert torch.allclose(op.to_dense().double(), ref, rtol=1e-6, atol=1e-7)


def test_gesn_encoder_surface():
    import argparse
    from sgp_amd.nn.encoders.dyn_gesn_encoder import gesn_operator
    p = sgp_amd.GESNEncoder.add_model_specific_args(argparse.ArgumentParser())
    a = p.parse_args([])
    assert (a.reservoir_size, a.reservoir_layers, a.density, a.alpha_decay) == (32, 1, .7, False)
    with pytest.raises(TypeError):           # the reference divides None by the degree (:39)
        gesn_operator(torch.zeros(2, 3, dtype=torch.long), None, 4)
    with pytest.raises(NotImplementedError):
        sgp_amd.GESNLayer(3, 8, aggr="mean")
    enc = sgp_amd.GESNEncoder(3, 8, 2, .9, .9, .7, 1., True)
    assert [float(c.alpha) for c in enc.reservoir.rnn_cells] == [.9, .8]


def test_matmul_checks_the_node_count_before_touching_a_device():
    """ADVICE: an operand with the wrong node count is a ValueError (the reference raises a shape
    error), not an out-of-bounds read -- checked on the host, before the GPU is required."""
    from sgp_amd import graph
    op = graph.ShiftOperator.from_edges(torch.tensor([[0, 1, 2], [1, 2, 0]]), None, 3)
    with pytest.raises(ValueError):
        op @ torch.randn(2, 4, 8)
    with pytest.raises(TypeError):
        op @ 3.0


def test_encode_dataset_return_device_requires_the_gpu_and_describe_is_optional(tmp_path):
    from sgp_amd import hip
    ds = FakeDataset(torch.randn(6, 4, 1), torch.randn(6, 2), torch.tensor([[0, 1], [1, 2]]), None)
    path = tmp_path / "e.pt"
    sgp_amd.encode_dataset(ds, StubEncoder, dict(input_size=3), save_path=str(path))
    assert path.exists() and not (tmp_path / "e.pt.encoder.pt").exists()     # stub has no describe()
    if not (torch.cuda.is_available() and hip.load() is not None):
        with pytest.raises(RuntimeError):
            sgp_amd.encode_dataset(ds, StubEncoder, dict(input_size=3), return_device=True)


def test_encoder_describe_round_trips_without_a_gpu():
    enc = sgp_amd.SGPEncoder(input_size=2, reservoir_size=8, reservoir_layers=3, leaking_rate=.9,
                             spectral_radius=.9, density=.7, input_scaling=1., receptive_field=2,
                             bidirectional=True, alpha_decay=True, global_attr=True)
    d = enc.describe()
    twin = sgp_amd.SGPEncoder(**d["kwargs"])
    twin.load_state_dict(d["state_dict"])
    assert d["alphas"] == pytest.approx([0.9, 0.8, 0.7])
    for a, b in zip(enc.parameters(), twin.parameters()):
        assert torch.equal(a, b)
    assert twin.output_size == enc.output_size


def test_dropout_rate_is_validated_like_dropout_adj():
    with pytest.raises(ValueError):
        sgp_amd.sgp_spatial_embedding(torch.randn(1, 3, 2), 3, torch.tensor([[0, 1], [1, 2]]), dropout_rate=-0.1)


def test_equal_cost_tiles_cover_the_rows_and_respect_the_limits():
    """graph.equal_cost_tiles (opt-in planner step): boundaries cover every row once,
    tiles are whole 4-row groups of at most 64 rows with at most max_union distinct columns, and
    the spread of the per-tile cost shrinks against uniform 64-row tiles."""
    import numpy as np
    from sgp_amd import graph, synthetic
    n = 36000
    ei, ew, _ = synthetic.knn_graph(n, 24, seed=5)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    lim = dict(max_union=448, max_tile_rows=64, max_row_edges=4096)
    rp, col, val = op.rowptr.numpy(), op.col.numpy(), op.val.numpy()
    base = graph.build_tile_plan(rp, col, val, n, equalize=False, **lim)
    eq = graph.build_tile_plan(rp, col, val, n, equalize=True, **lim)
    assert base is not None and eq is not None and base.n_tiles >= 512
    trow = eq.trow.numpy()
    assert trow[0] == 0 and trow[-1] == n and (np.diff(trow) > 0).all()
    rows = np.diff(trow)
    assert rows.max() <= 64 and (rows[:-1] % 4 == 0).all()
    assert eq.pipe["max_union"] <= 448 + 4
    c0, c1 = np.asarray(base.pipe["phase_cost"], float), np.asarray(eq.pipe["phase_cost"], float)
    assert c1.std() / c1.mean() <= c0.std() / c0.mean()      # (36k nodes x 24-NN: 4.5 % -> 3.6 %; 100-NN target: 20 % -> 4 %)
    # the stream still reproduces the operator: every edge weight appears exactly once
    assert abs(float(eq.pipe["gw"].double().sum()) - float(op.val.double().sum())) < 1e-3 * n
    assert sorted(eq.pipe["rowmap"][eq.pipe["rowmap"] >= 0].tolist()) == list(range(n))


def test_sgp_tune_is_one_hook_read_alike_by_python_and_the_library(monkeypatch):
    """SGP_TUNE="key=value,..." is the one tuning variable: sgp_amd/tune.py and libsgp_amd.so parse it alike."""
    from sgp_amd import hip, tune
    lib = hip.load()
    monkeypatch.delenv("SGP_TUNE", raising=False)
    assert tune.get("hop", "split") == "split" and tune.get("mix_thr", 4, int) == 4
    assert lib.sgp_tune_value(b"spmm_chunk", 32) == 32
    monkeypatch.setenv("SGP_TUNE", "hop=exact, spmm_chunk=16,mix_min_share=0.4,abl=3072")
    assert tune.get("hop", "split") == "exact" and tune.get("spmm_chunk", 32, int) == 16
    assert tune.get("mix_min_share", 0.25, float) == 0.4 and tune.get("absent", 7, int) == 7
    assert lib.sgp_tune_value(b"spmm_chunk", 32) == 16 and lib.sgp_tune_value(b"abl", 0) == 3072
    assert lib.sgp_tune_value(b"chunk", 5) == 5                       # a suffix of a key is not the key
    with pytest.raises(ValueError):
        tune.get("hop", 1, int)


def test_small_graph_pipeline_pieces_follow_the_hops_share_of_the_chain(monkeypatch):
    """SGPEncoder._overlap_pieces (host logic of encode_device): 16 time pieces where the hops take as long as the
    reservoir chain (PEMS-BAY settings), 4 where they are a seventh of it (METR-LA settings), one piece for short
    sequences, for graphs beyond `overlap_tiles` node tiles, and when SGP_TUNE overrides the count."""
    monkeypatch.delenv("SGP_TUNE", raising=False)
    bay = sgp_amd.SGPEncoder(input_size=3, reservoir_size=128, reservoir_layers=1, leaking_rate=.8, spectral_radius=.9,
                             density=.7, input_scaling=1., receptive_field=4, bidirectional=True, alpha_decay=False,
                             global_attr=True)
    la = sgp_amd.SGPEncoder(input_size=3, reservoir_size=64, reservoir_layers=1, leaking_rate=.9, spectral_radius=.9,
                            density=.7, input_scaling=1., receptive_field=2, bidirectional=False, alpha_decay=False,
                            global_attr=False)
    assert bay._overlap_pieces(52116, 325) == 16 and la._overlap_pieces(34272, 207) == 4
    assert bay._overlap_pieces(1000, 325) == 8 and bay._overlap_pieces(100, 325) == 1   # >= 64 steps per piece: the next smaller count
    assert bay._overlap_pieces(52116, 16 * bay.overlap_tiles + 1) == 1
    monkeypatch.setenv("SGP_TUNE", "overlap_chunks=2")
    assert bay._overlap_pieces(52116, 325) == 2 and la._overlap_pieces(34272, 207) == 2


def test_unit_bound_mark_follows_the_tensor_version():
    """The a-priori bound 1 of the split-fp16 hop may be used for a carried state only while nobody edited it: the
    mark is tied to the tensor's version counter (round-5 advice: a Python attribute survived ``state.mul_(5)``)."""
    from sgp_amd import hip
    state = torch.zeros(2, 5, 4)
    assert not hip.is_unit_bounded(state) and not hip.is_unit_bounded(None)
    hip.mark_unit_bounded(state)
    assert hip.is_unit_bounded(state)
    state.mul_(5)
    assert not hip.is_unit_bounded(state)
    hip.mark_unit_bounded(state.zero_())
    assert hip.is_unit_bounded(state)


def test_overlap_pieces_walk_down_for_short_sequences():
    """A sequence too short for the chosen piece count takes the next smaller count, not one piece (round-5 advice)."""
    import sgp_amd
    enc = sgp_amd.SGPEncoder(input_size=3, reservoir_size=128, reservoir_layers=1, leaking_rate=0.8, spectral_radius=0.9,
                             density=0.7, input_scaling=1., receptive_field=4, bidirectional=True, alpha_decay=False,
                             global_attr=True)
    full = enc._overlap_pieces(52116, 325)
    assert full == 16
    assert enc._overlap_pieces(800, 325) == 8 and enc._overlap_pieces(300, 325) == 4 and enc._overlap_pieces(100, 325) == 1
    assert enc._overlap_pieces(52116, 100000) == 1                      # large graphs: no pieces
