"""The narrow-reservoir kernel with three-piece bf16 products (sgp_amd/csrc/reservoir_bf3.h; reference:
lib/nn/reservoir/reservoir.py:77-81, :158-186) against the oracle in fp64 and fp32: every product is summed from six
bf16 piece products with fp32 accumulation, so the result must be as close to the fp64 evaluation as the CPU's own fp32
run is (tolerance below), for every activation, with the exact deal + tail, ragged last tiles, carried state and
strided output rows."""
import pytest
import torch

import oracle.sgp_oracle as O
import sgp_amd
from sgp_amd import hip

pytestmark = pytest.mark.gpu


def layers_of(res):
    return [dict(w_ih=l.w_ih.data.cpu(), w_hh=l.w_hh.data.cpu(), b_ih=l.b_ih.data.cpu(), alpha=float(l.alpha))
            for l in res.reservoir_layers]


def check(out, x, res, act, nodes):
    """max |error| against fp64 on a sample of nodes: no more than twice the CPU fp32 run's (+ 1e-6)."""
    idx = torch.as_tensor(nodes)
    ref64 = O.reservoir_forward(x[:, idx], layers_of(res), activation=act, dtype=torch.float64)
    ref32 = O.reservoir_forward(x[:, idx], layers_of(res), activation=act)
    got = out[:, idx.cuda()].cpu()
    assert torch.isfinite(got).all()
    e_gpu = float((got.double() - ref64).abs().max())
    e_cpu = float((ref32.double() - ref64).abs().max())
    assert e_gpu <= 2 * e_cpu + 1e-6, (e_gpu, e_cpu)
    assert O.rel_fro(got, ref32) <= 1e-5          # the parity bar of the encoder (north star: 1e-5 relative)
    return e_gpu, e_cpu


@pytest.mark.parametrize("n,f,r,act", [
    (20000, 64, 64, "tanh"),            # 1250 tiles: one or two tiles per wave, no tail
    (16 * 1024 * 2 + 16 * 37 + 5, 64, 64, "tanh"),   # exact deal (2 per SIMD) + split-J tail + ragged last tile
    (40007, 32, 32, "tanh"),            # JT = 2, ragged
    (30000, 16, 64, "relu"),
    (30000, 64, 32, "self_norm"),
    (9000, 32, 64, "tanh"),             # 563 tiles: single-tile workgroups
    (16384 * 5 + 16 * 3 + 1, 16, 32, "tanh"),   # five tiles per SIMD: waves with two tiles and with one
    (16384 * 6 + 16 * 200, 64, 64, "tanh"),     # the target line's shape of deal (6 per SIMD + tail)
])
def test_bf3_reservoir_matches_fp64_as_well_as_fp32_does(n, f, r, act):
    hip.require_gpu()
    torch.manual_seed(n % 89)
    t = 24
    res = sgp_amd.Reservoir(f, r, activation=act, spectral_radius=0.5 if act == "relu" else 0.9)
    x = torch.randn(t, n, f)
    out = torch.full((t, n, r), float("nan"), device="cuda")
    res.encode_into(x.cuda(), out)
    assert torch.isfinite(out).all()
    nodes = sorted({0, 1, 15, 16, 17, n // 3, n // 2, n - 17, n - 16, n - 2, n - 1} | set(range(4096, 4096 + 40)))
    check(out, x, res, act, nodes)


def test_bf3_reservoir_large_inputs_and_state_carry():
    """Inputs of magnitude 1e4 and 1e-4 in the same row (no scale is involved: bf16 keeps the exponent range of
    fp32), the state carried across two calls, the output written into a strided slot of a wider buffer."""
    hip.require_gpu()
    torch.manual_seed(3)
    n, f, r, t = 20000, 64, 64, 12
    res = sgp_amd.Reservoir(f, r, input_scaling=1e-3)
    x = torch.randn(t, n, f)
    x[:, :, ::2] *= 1e4
    x[:, :, 1::2] *= 1e-4
    xg = x.cuda()
    wide = torch.zeros(t, n, 3 * r + 4, device="cuda")
    res.encode_into(xg, wide[:, :, r:2 * r])
    assert float(wide[:, :, :r].abs().max()) == 0.0 and float(wide[:, :, 2 * r:].abs().max()) == 0.0
    out = wide[:, :, r:2 * r]
    check(out, x, res, "tanh", [0, 7, 4999, 12345, n - 1])
    state = torch.zeros(1, n, r, device="cuda")
    out2 = torch.empty(t, n, r, device="cuda")
    res.encode_into(xg[:5], out2[:5], state)
    res.encode_into(xg[5:], out2[5:], state)
    assert torch.equal(out2, out.contiguous())
    assert torch.equal(state[0], out2[-1])


def test_bf3_products_are_exact_for_bf16_representable_operands():
    """Operands with at most 8 significant bits and a short dot product: the six piece products are then the exact
    product, the fp32 accumulation is exact too, and the pre-activation equals the fp64 one -- what is left is the
    activation's own error (tanh: < 3e-7 absolute)."""
    hip.require_gpu()
    torch.manual_seed(5)
    n, f, r, t = 20000, 16, 32, 3
    res = sgp_amd.Reservoir(f, r)
    for l in res.reservoir_layers:
        l.w_ih.data = torch.randint(-8, 9, l.w_ih.shape).float() / 64
        l.w_hh.data = torch.randint(-8, 9, l.w_hh.shape).float() / 128
        l.b_ih.data = torch.randint(-8, 9, l.b_ih.shape).float() / 16
    x = torch.randint(-100, 101, (t, n, f)).float() / 32
    out = torch.empty(t, n, r, device="cuda")
    res.encode_into(x.cuda(), out)
    ref64 = O.reservoir_forward(x[:1, :256], layers_of(res), dtype=torch.float64)      # step 0: state 0, exact operands
    assert float((out[:1, :256].cpu().double() - ref64).abs().max()) < 4e-7


@pytest.mark.parametrize("t", [1, 2, 7])
def test_bf3_wide_reservoir_short_sequences_strided_rows_and_state(t):
    """R = 256 (fragments streamed through the LDS, reservoir_layer_stream_bf3): the state of a step is stored during
    the NEXT step's sub-blocks and once more after the last one -- sequences of 1 and 2 steps, output rows inside a
    wider buffer, ragged last tile, the state handed on through h_state."""
    hip.require_gpu()
    torch.manual_seed(t)
    n, f, r = 2048 * 16 + 16 * 5 + 3, 64, 256
    res = sgp_amd.Reservoir(f, r, spectral_radius=0.9)
    x = torch.randn(t + 3, n, f)
    xg = x.cuda()
    wide = torch.zeros(t + 3, n, r + 2 * 64, device="cuda")
    state = torch.zeros(1, n, r, device="cuda")
    res.encode_into(xg[:t], wide[:t, :, 64:64 + r], state)
    res.encode_into(xg[t:], wide[t:, :, 64:64 + r], state)
    assert float(wide[:, :, :64].abs().max()) == 0.0 and float(wide[:, :, 64 + r:].abs().max()) == 0.0
    out = wide[:, :, 64:64 + r]
    assert torch.equal(state[0], out[-1])
    check(out, x, res, "tanh", [0, 5, 16 * 1000 + 3, n - 20, n - 2, n - 1])


@pytest.mark.parametrize("n,f,r,act,t", [
    (207, 3, 64, "tanh", 300),          # METR-LA shape (C1): 13 node tiles, one output tile per wave
    (325, 3, 128, "tanh", 300),         # PEMS-BAY shape (C2): two output tiles per wave
    (325, 3, 128, "relu", 120),
    (207, 3, 64, "self_norm", 120),
    (500, 5, 100, "tanh", 60),          # padded units (R = 100 of 128) and features (F = 5 of 8)
    (77, 20, 50, "tanh", 60),           # F = 20: the 16-byte feature order of the wider input blocks; ragged last tile
    (8190, 3, 64, "tanh", 40),          # 512 node tiles: the largest problem the split-J form serves alone
    (1000, 32, 128, "tanh", 40),        # a full input k-block
])
def test_split_j_bf3_small_graphs(n, f, r, act, t):
    """The small-N form (reservoir_splitj_bf3.h: one node tile per workgroup, the output tiles split over its four
    waves, the state exchanged as bf16 pieces through LDS) at the traffic configs' shapes: as close to fp64 as the
    CPU's fp32 run, every activation, padded widths, and the state carried across calls."""
    hip.require_gpu()
    torch.manual_seed(n + r)
    res = sgp_amd.Reservoir(f, r, activation=act, spectral_radius=0.5 if act == "relu" else 0.9)
    x = torch.randn(t, n, f)
    out = torch.full((t, n, r), float("nan"), device="cuda")
    res.encode_into(x.cuda(), out)
    assert torch.isfinite(out).all()
    check(out, x, res, act, sorted({0, 1, 15, 16, n // 2, n - 2, n - 1}))
    state = torch.zeros(1, n, r, device="cuda")
    wide = torch.zeros(t, n, r + 8, device="cuda")
    res.encode_into(x[:t // 3].cuda(), wide[:t // 3, :, 4:4 + r], state)
    res.encode_into(x[t // 3:].cuda(), wide[t // 3:, :, 4:4 + r], state)
    assert torch.equal(wide[:, :, 4:4 + r], out)                       # same bits whatever the cut, strided rows
    assert float(wide[:, :, :4].abs().max()) == 0.0 and float(wide[:, :, 4 + r:].abs().max()) == 0.0
    assert torch.equal(state[0], out[-1])
