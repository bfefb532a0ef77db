"""Small graphs: the reservoir's time axis cut into pieces that run side by side (ReservoirLayer._run_time_parallel;
reference: the serial loop lib/nn/reservoir/reservoir.py:170-183).  A piece started a warm-up early from zero reaches the
true state only if the recurrence forgets; that is CHECKED on the device at every splice, and a rejected splice reruns
the sequential chain under a launch predicate -- the result is right either way."""
import os

import pytest
import torch

import sgp_amd
from oracle import sgp_oracle as O
from sgp_amd import hip

pytestmark = pytest.mark.gpu


def _layer(R, F=3, alpha=0.9, rho=0.9, act="tanh", seed=5):
    torch.manual_seed(seed)
    res = sgp_amd.Reservoir(F, R, leaking_rate=alpha, spectral_radius=rho, density=0.7, activation=act)
    return res, res.reservoir_layers[0]


def _sequential(layer, x, h0=None):
    saved = os.environ.get("SGP_TUNE")
    os.environ["SGP_TUNE"] = "time_parallel=0"
    try:
        out = torch.empty(x.shape[0], x.shape[1], layer.hidden_size, device=x.device)
        st = None if h0 is None else h0.clone()
        layer.run_sequence(x, out, st)
        assert layer.last_time_parallel is None
        return out, st
    finally:
        if saved is None:
            del os.environ["SGP_TUNE"]
        else:
            os.environ["SGP_TUNE"] = saved


@pytest.mark.parametrize("n,r,t", [(207, 64, 6000), (325, 128, 5000), (40, 48, 9000)])
def test_accepted_pieces_match_the_sequential_chain_and_the_fp64_oracle(n, r, t):
    res, layer = _layer(r)
    x = torch.randn(t, n, 3, device="cuda")
    plan = layer.time_parallel_plan(t, n, 3, x.device)
    assert plan is not None and plan[0] >= 2
    h0 = (torch.rand(n, r, device="cuda") - 0.5)
    out, st = torch.empty(t, n, r, device="cuda"), h0.clone()
    layer.run_sequence(x, out, st)
    info = layer.last_time_parallel
    assert info is not None and info["pieces"] == plan[0] and int(info["flag"]) == 1, float(info["gap"])
    seq, st_seq = _sequential(layer, x, h0)
    # at most the splice tolerance apart (contracting behind every cut), states included
    assert float((out - seq).abs().max()) <= 1e-6 and float((st - st_seq).abs().max()) <= 1e-6
    assert not torch.equal(out, seq)                      # (the pieces really ran: the low bits differ behind a cut)
    cut = plan[1]
    assert torch.equal(out[:cut], seq[:cut])              # piece 0 starts from the caller's state: identical bits
    # the encoder's criterion against the fp64 evaluation, as for the sequential kernel
    lay = [dict(w_ih=layer.w_ih.data, w_hh=layer.w_hh.data, b_ih=layer.b_ih.data, alpha=float(layer.alpha))]
    ref64 = O.reservoir_forward(x.cpu(), lay, h0=h0.cpu()[None], dtype=torch.float64)
    e_par, e_seq = float((out.cpu().double() - ref64).abs().max()), float((seq.cpu().double() - ref64).abs().max())
    assert e_par <= max(5e-6, 2 * e_seq), (e_par, e_seq)
    assert torch.allclose(out.cpu(), ref64.float(), rtol=1e-5, atol=1e-5)


def test_a_reservoir_that_does_not_forget_is_rejected_and_repaired():
    """tanh with spectral radius 3: trajectories from different starts never meet.  (The nominal rate is >= 1, so the
    plan is refused up front -- forced here through SGP_TUNE to show the device-side guard: flag 0, and the sequential
    launch behind the predicate leaves exactly the sequential result.)"""
    res, layer = _layer(64, rho=3.0)
    n, t = 207, 4000
    x = torch.randn(t, n, 3, device="cuda")
    assert layer.time_parallel_plan(t, n, 3, x.device) is None
    seq, st_seq = _sequential(layer, x, torch.zeros(n, 64, device="cuda"))
    os.environ["SGP_TUNE"] = "time_parallel_warm=256"
    try:
        assert layer.time_parallel_plan(t, n, 3, x.device) is not None
        out, st = torch.full((t, n, 64), float("nan"), device="cuda"), torch.zeros(n, 64, device="cuda")
        layer.run_sequence(x, out, st)
    finally:
        del os.environ["SGP_TUNE"]
    info = layer.last_time_parallel
    assert int(info["flag"]) == 0 and float(info["gap"]) > 1e-3
    assert torch.equal(out, seq) and torch.equal(st, st_seq)


def test_shapes_and_activations_outside_the_premises_keep_the_one_chain():
    n, t = 207, 6000
    dev = torch.device("cuda")
    for kw in (dict(act="relu"), dict(act="self_norm"), dict(alpha=1.3)):
        _, layer = _layer(64, **kw)
        assert layer.time_parallel_plan(t, n, 3, dev) is None, kw
    _, layer = _layer(64)
    assert layer.time_parallel_plan(500, n, 3, dev) is None            # shorter than two pieces of two warm-ups
    assert layer.time_parallel_plan(t, 100000, 3, dev) is None         # large graphs fill the chip with node tiles
    _, wide = _layer(256)
    assert wide.time_parallel_plan(t, n, 3, dev) is None               # not a shape of the piece kernel
    _, tiny = _layer(64)
    tiny.b_ih.data.mul_(1e-3)
    assert tiny.kernel_activation() == "tanh_rel" and tiny.time_parallel_plan(t, n, 3, dev) is None


def test_encoder_with_time_pieces_meets_the_oracle_through_the_default_call():
    """METR-LA-shaped encoder (config/traffic/sgp_la.yaml's reservoir: R = 64, L = 2, alpha decay) on 6000 steps: the
    default forward takes the time pieces layer by layer (layer 1 reads layer 0's slot), then the hops."""
    torch.manual_seed(3)
    n, t = 207, 6000
    from sgp_amd import synthetic
    ei, ew = synthetic.sparse_traffic_graph(n, 1515, seed=3)
    enc = sgp_amd.SGPEncoder(input_size=3, reservoir_size=64, reservoir_layers=2, leaking_rate=0.9, spectral_radius=0.9,
                             density=0.7, input_scaling=1., receptive_field=2, bidirectional=True, alpha_decay=True,
                             global_attr=True)
    enc.reservoir.fused = False
    x = torch.randn(t, n, 3)
    ops = enc.sgp_encoder.operators(n, ei, ew)
    y = enc.encode_device(x.cuda(), ops).cpu()
    assert all(l.last_time_parallel is not None and int(l.last_time_parallel["flag"]) == 1 for l in enc.reservoir.reservoir_layers)
    layers = [dict(w_ih=l.w_ih.data, w_hh=l.w_hh.data, b_ih=l.b_ih.data, alpha=float(l.alpha)) for l in enc.reservoir.reservoir_layers]
    ref = O.sgp_encoder_forward(x, ei, ew, layers, 2, bidirectional=True, global_attr=True, sparse=True)
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5), float((y - ref).abs().max())


def test_the_unit_bound_mark_survives_the_time_pieces():
    """The pieces hand the caller's carried state back through an in-place copy (a version bump): a state that was marked
    as inside [-1, 1] stays marked -- the next time chunk of a streamed encoding keeps its a-priori bound instead of
    measuring one."""
    res, layer = _layer(64)
    n, t = 207, 6000
    x = torch.randn(t, n, 3, device="cuda")
    st = hip.mark_unit_bounded(torch.zeros(n, 64, device="cuda"))
    out = torch.empty(t, n, 64, device="cuda")
    layer.run_sequence(x, out, st)
    assert layer.last_time_parallel is not None and hip.is_unit_bounded(st)
    raw = torch.zeros(n, 64, device="cuda")                    # an unmarked state stays unmarked
    layer.run_sequence(x, out, raw)
    assert not hip.is_unit_bounded(raw)
