"""The CPU oracle (oracle/sgp_oracle.py) against the golden vectors recorded
from the unmodified reference (oracle/make_golden.py).  CPU-only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_files
from oracle import sgp_oracle as O

RTOL = ATOL = 1e-5          # north_star: 1e-5 relative fp32 tolerance


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def close(a, b, rtol=RTOL, atol=ATOL):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape
    assert torch.allclose(a, b, rtol=rtol, atol=atol), \
        f"max abs {float((a - b).abs().max()):.3e}"
    assert O.rel_fro(a, b) <= 1e-5


@pytest.mark.parametrize("name", golden_files("g0_"))
def test_reservoir(name):
    z = load(name)
    layers = O.layers_from_npz(z)
    x = torch.from_numpy(z["x"])
    act = str(z["activation"])
    y = O.reservoir_forward(x, layers, act)
    close(y, z["y"])
    close(O.reservoir_forward(x, layers, act, return_last_state=True), z["y_last"])
    y64 = O.reservoir_forward(x, layers, act, dtype=torch.float64)
    close(y64, z["y64"], rtol=1e-10, atol=1e-10)
    # fp32 within 5e-6 of the fp64 evaluation (SURVEY.md 8d parity criterion)
    r64 = torch.from_numpy(z["y64"])
    assert float(((y.double() - r64).abs() / r64.abs().clamp_min(1)).max()) < 5e-6


@pytest.mark.parametrize("name", golden_files("g1_spatial"))
def test_spatial_encoder(name):
    z = load(name)
    x = torch.from_numpy(z["x"])
    for sparse in (False, True):
        y = O.spatial_encoder_forward(
            x, z["edge_index"], z["edge_weight"], int(z["k"]),
            bool(z["bidirectional"]), bool(z["undirected"]),
            bool(z["global_attr"]), bool(z["add_self_loops"]), sparse=sparse)
        close(y, z["y"])
    y64 = O.spatial_encoder_forward(
        x.double(), z["edge_index"], z["edge_weight"], int(z["k"]),
        bool(z["bidirectional"]), bool(z["undirected"]),
        bool(z["global_attr"]), bool(z["add_self_loops"]))
    close(y64, z["y64"], rtol=1e-6, atol=1e-7)   # edge weights are fp32 in both


def test_embedding_function_api():
    z = load("g1_embedding_noweight.npz")
    x = torch.from_numpy(z["x"])
    y = torch.cat(O.spatial_embedding(x, z["edge_index"], None, k=2), -1)
    close(y, z["y"])
    z = load("g1_embedding_removeloops.npz")
    y = torch.cat(O.spatial_embedding(x, z["edge_index"], z["edge_weight"], k=2,
                                      remove_self_loops=True,
                                      bidirectional=True), -1)
    close(y, z["y"])


@pytest.mark.parametrize("name", golden_files("g2_"))
def test_full_encoder(name):
    z = load(name)
    layers = O.layers_from_npz(z)
    x = torch.from_numpy(z["x"])
    for sparse in (False, True):
        y = O.sgp_encoder_forward(
            x, z["edge_index"], z["edge_weight"], layers,
            int(z["receptive_field"]), bool(z["bidirectional"]),
            bool(z["undirected"]), bool(z["global_attr"]),
            bool(z["add_self_loops"]), sparse=sparse)
        close(y, z["y"])


@pytest.mark.parametrize("name", golden_files("g3_gesn"))
def test_gesn(name):
    z = load(name)
    layers = O.layers_from_npz(z)
    act = str(z["activation"]) if "activation" in z else "tanh"
    y = O.gesn_forward(torch.from_numpy(z["x"]), z["edge_index"],
                       torch.from_numpy(z["edge_weight"]), layers, activation=act)
    close(y, z["y"])


@pytest.mark.parametrize("name", golden_files("g4_seed"))
def test_seed_to_weights(name):
    z = load(name)
    torch.manual_seed(int(z["seed"]))
    if name.startswith("g4_seed_gesn"):
        layers = O.init_reservoir(3, 16, num_layers=2, density=.8,
                                  alpha_decay=True, redraw=True)
    else:
        f, r, L, a, rho, dens, scale = z["cfg"]
        layers = O.init_reservoir(int(f), int(r), input_scaling=scale,
                                  num_layers=int(L), leaking_rate=a,
                                  spectral_radius=rho, density=dens,
                                  alpha_decay=True)
    after = torch.rand(4)
    ref = O.layers_from_npz(z)
    assert len(ref) == len(layers)
    for a_, b_ in zip(layers, ref):
        for k in ("w_ih", "w_hh", "b_ih"):
            assert torch.equal(a_[k], b_[k]), k
        assert a_["alpha"] == b_["alpha"]
    assert torch.equal(after, torch.from_numpy(z["rng_after"]))


def test_identity_activation_raises_like_reference():
    layers = O.init_reservoir(3, 8)
    with pytest.raises(ValueError):
        O.reservoir_forward(torch.zeros(2, 3, 3), layers, "identity")
    with pytest.raises(AssertionError):
        O.reservoir_forward(torch.zeros(2, 3, 3), layers, "gelu")


def test_properties():
    torch.manual_seed(0)
    n, t, d = 30, 5, 6
    x = torch.randn(t, n, d)
    ei = torch.randint(0, n, (2, 120))
    ew = torch.rand(120)
    # K=0 -> only x
    assert torch.equal(torch.cat(O.spatial_embedding(x, ei, ew, k=0), -1), x)
    # A = I -> every hop equals hop 0
    idx = torch.arange(n)
    outs = O.spatial_embedding(x, torch.stack([idx, idx]), None, k=3)
    for o in outs[1:]:
        close(o, x)
    # zero in-degree rows -> zero rows after one hop (sgp_preprocessing.py:102)
    ei2 = ei.clone()
    ei2[1][ei2[1] == 4] = 3
    out = O.spatial_embedding(x, ei2, ew, k=1)[1]
    assert float(out[:, 4].abs().max()) == 0.0
    # permutation equivariance over nodes
    perm = torch.randperm(n)
    inv = torch.empty_like(perm)
    inv[perm] = idx
    y = torch.cat(O.spatial_embedding(x, ei, ew, k=2), -1)
    yp = torch.cat(O.spatial_embedding(x[:, perm], inv[ei], ew, k=2), -1)
    close(yp, y[:, perm])


# ------------------------------------------------------------------ f1: IID sampling
@pytest.mark.parametrize("name", golden_files("g7_iid_"))
def test_iid_sampling(name):
    """oracle restatement of IIDDataset.sample == what the reference returned (indices, inputs,
    targets; RNG order; scaler applied after the gather)."""
    z = load(name)
    hz, delay, lag, n = [int(v) for v in z["cfg"]]
    emb, y, u = (torch.from_numpy(z[k]) for k in ("emb", "y", "u"))
    torch.manual_seed(int(z["seed"]))
    st, nd = O.iid_draw(emb.shape[0], emb.shape[1], hz, n)
    assert np.array_equal(st.numpy(), z["step_index"])
    assert np.array_equal(nd.numpy(), z["node_index"])
    assert np.array_equal(O.iid_gather_input(emb, "t n f", st, nd).numpy(), z["out_x"])
    ty = O.iid_gather_target(y, "t n f", O.iid_horizon_index(st, delay, hz, lag), nd)
    if bool(z["has_scaler"]):
        ty = (ty - torch.from_numpy(z["bias"])) / torch.from_numpy(z["scale"])
    assert np.array_equal(ty.numpy(), z["out_y"])
    if bool(z["has_exo"]):
        assert np.array_equal(O.iid_gather_input(u, "t f", st, nd).numpy(), z["out_u"])


# ------------------------------------------------------------------ f4: decoder input layer
@pytest.mark.parametrize("name", golden_files("g8_decoder_"))
def test_decoder_input_encoder(name):
    z = load(name)
    f, order, hidden = [int(v) for v in z["cfg"]]
    x, w, b = (torch.from_numpy(z[k]) for k in ("x", "weight", "bias"))
    y = O.decoder_input_encoder(x, w, b, order, str(z["activation"]))
    close(y, z["y"])
    y64 = O.decoder_input_encoder(x.double(), w.double(), b.double(), order, str(z["activation"]))
    close(y64, z["y64"], rtol=1e-10, atol=1e-10)
    # backward pass of the trained layer: oracle (written out) == the reference module's autograd
    gx, gw, gb = O.decoder_input_encoder_grads(x.double(), w.double(), b.double(), order, str(z["activation"]),
                                               torch.from_numpy(z["gy"]).double())
    sc = lambda a: float(np.abs(a).max())
    close(gx.float(), z["gx"], rtol=1e-5, atol=1e-5 * sc(z["gx"]))
    close(gw.float(), z["gw"], rtol=1e-5, atol=1e-5 * sc(z["gw"]))
    close(gb.float(), z["gb"], rtol=1e-5, atol=1e-5 * sc(z["gb"]))


# ------------------------------------------------------------------ f3: on-the-fly supports
@pytest.mark.parametrize("name", golden_files("g9_onthefly_"))
def test_onthefly_supports(name):
    z = load(name)
    kw = {k: (bool(z[k]) if z[k].dtype == bool else int(z[k])) for k in
          ("k", "undirected", "add_self_loops", "remove_self_loops", "bidirectional", "global_attr")
          if k in z.files}
    ei, ew, n = torch.from_numpy(z["edge_index"]), torch.from_numpy(z["edge_weight"]), int(z["n"])
    sup = O.spatial_support_dense(ei, ew, n, **kw)
    x = torch.from_numpy(z["x"])
    close(O.apply_supports_dense(x, sup), z["full"])
    close(O.apply_supports_dense(x, sup, torch.from_numpy(z["node_index"])), z["sub"])
