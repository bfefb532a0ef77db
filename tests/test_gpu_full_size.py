"""The shapes ``bench.py`` times, verified at their OWN number of time steps (round-3 review: the target graph
was only checked at t = 3, C1 / C2 at truncated T): every time chunk of the hop kernels on the target line's
T = 1024, and the METR-LA / PEMS-BAY shaped configurations C1 / C2 at T = 34 272 / 52 116 (64-bit batch
offsets: C2's embedding is 86.7 GB) -- last steps and final reservoir state against the fp64 CPU oracle."""
import pytest
import torch

import sgp_amd
from oracle import sgp_oracle as O
from sgp_amd import graph, hip, synthetic
from test_gpu_parity import close, layers_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    hip.require_gpu()


def test_target_graph_hop_at_full_T_every_time_chunk():
    """N = 100 000, 100-NN, T = 1024, D = 64 inside a strided slot buffer: three steps of EVERY 64-step chunk
    (first, the 32-step boundary of the exact kernels' chunks, last) of the split-fp16 hop and of the
    exact-fp32 mixed kernel against the generic CSR kernel on the same operand."""
    n, d, t = 100000, 64, 1024
    ei, ew, _ = synthetic.knn_graph(n, 100, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, n)
    buf = torch.empty(t, n, 2 * d, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    for t0 in range(0, t, 64):                                   # (no 26 GB temporary)
        buf[t0:t0 + 64, :, :d] = torch.tanh(torch.randn(64, n, d, device="cuda", generator=g))
    x, y = buf[:, :, :d], buf[:, :, d:]
    steps = sorted({s for c in range(0, t, 64) for s in (c, c + 31, c + 32, c + 63)})
    idx = torch.tensor(steps, device="cuda")
    ref = torch.empty(len(steps), n, d, device="cuda")
    op.propagate(x[idx].contiguous(), ref, force="csr")
    for force, bound in (("split", 1.0), ("mix", None)):
        y.fill_(float("nan"))
        op.propagate(x, y, force=force, x_bound=bound)
        got = y[idx]
        assert torch.isfinite(y[::97]).all()
        close(got, ref, rtol=1e-5, atol=1e-5, fro=2e-6)
    # the default choice on this operand is the split kernel, with the bound measured when none is given
    op.propagate(x, y)
    assert op.last_kernel == "spmm_split"
    close(y[idx], ref, rtol=1e-5, atol=1e-5, fro=2e-6)


@pytest.mark.parametrize("cfg", ["c1", "c2"])
def test_baseline_configs_at_their_full_T(cfg):
    """C1 (N = 207, T = 34 272, R = 64, K = 2) and C2 (N = 325, T = 52 116, R = 128, K = 4, both directions,
    global block): the last 16 steps of the embedding and the final reservoir state against the fp64 oracle
    (the recurrence runs over ALL steps on the CPU; tanh reservoirs contract, so fp32 rounding does not
    accumulate: the criterion is the plain 1e-5)."""
    torch.manual_seed(42)
    if cfg == "c1":
        n, e, t, kw = 207, 1515, 34272, dict(reservoir_size=64, reservoir_layers=1, leaking_rate=.9,
                                             receptive_field=2, bidirectional=False,
                                             alpha_decay=False, global_attr=False)
    else:
        n, e, t, kw = 325, 2369, 52116, dict(reservoir_size=128, reservoir_layers=1, leaking_rate=.8,
                                             receptive_field=4, bidirectional=True,
                                             alpha_decay=True, global_attr=True)
    ei, ew = synthetic.sparse_traffic_graph(n, e, seed=1)
    enc = sgp_amd.SGPEncoder(input_size=3, spectral_radius=.9, density=.7, input_scaling=1., **kw)
    x = torch.randn(t, n, 3)
    ops = enc.sgp_encoder.operators(n, ei, ew)
    state = torch.zeros(1, n, kw["reservoir_size"], device="cuda")
    out = enc.encode_device(x.cuda(), ops, state=state)
    assert out.shape == (t, n, enc.output_size)
    tail = out[-16:].cpu()
    # fp64 oracle: reservoir over all T steps, propagation on the last 16.  (Few threads: the per-step operands are
    # small, and a 256-thread pool spends the time of ~50 000 steps in dispatch.)
    layers = layers_of(enc.reservoir)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 8))
    try:
        h = O.reservoir_forward(x, layers, dtype=torch.float64)
    finally:
        torch.set_num_threads(threads)
    cops = O.shift_operators_csr(ei, ew, n, bidirectional=kw["bidirectional"])
    cops = [a.to(torch.float64) for a in cops]
    blocks = [h[-16:]]
    for a in cops:
        z = h[-16:]
        for _ in range(kw["receptive_field"]):
            z = torch.stack([a @ z[b] for b in range(z.shape[0])])
            blocks.append(z)
    if kw["global_attr"]:
        blocks.append(torch.ones_like(h[-16:]) * h[-16:].mean(-2, keepdim=True))
    ref = torch.cat(blocks, -1)
    close(tail, ref.float())
    close(state[0].cpu(), h[-1].float())
    # and a few steps from the middle of the sequence (64-bit batch offsets of the hop kernels)
    mid = t // 2 + 12345 % 97
    close(out[mid, :, :kw["reservoir_size"]].cpu(), h[mid].float())
    z = h[mid:mid + 1]
    z = torch.stack([cops[0] @ z[0]])
    r = kw["reservoir_size"]
    close(out[mid:mid + 1, :, r:2 * r].cpu(), z.float())
