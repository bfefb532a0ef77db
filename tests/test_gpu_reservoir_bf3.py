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
    (3000, 64, 128, "tanh", 40),        # two input k-blocks (F = 64)
    (500, 40, 64, "relu", 40),          # two input k-blocks, the second one padded (F = 40 of 64)
    (8000, 64, 64, "tanh", 30),         # 500 node tiles, F = R = 64
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


def _per_unit_rel(got, ref):
    """relative Frobenius error of every reservoir unit's series (the max-abs check of `check` says nothing about
    units whose weights, and so whose states, are orders of magnitude below the others)"""
    num = (got.double() - ref.double()).pow(2).sum(dim=(0, 1)).sqrt()
    den = ref.double().pow(2).sum(dim=(0, 1)).sqrt()
    return num / den.clamp_min(1e-300)


@pytest.mark.parametrize("n,f,r", [(207, 3, 64), (325, 3, 128)])
def test_split_j_two_piece_fp16_state_weight_rows_of_mixed_magnitude(n, f, r):
    """The bounded-state loop of the split-J form (two fp16 pieces per recurrent operand, every row of W_hh under its own
    power-of-two scale, reservoir_splitj_bf3.h): rows of W_hh seven orders of magnitude apart and entries five orders
    apart INSIDE a row -- every unit's series is as close to fp64 as the CPU's fp32 run (relative, per unit)."""
    hip.require_gpu()
    torch.manual_seed(r)
    res = sgp_amd.Reservoir(f, r, spectral_radius=0.9)
    layer = res.reservoir_layers[0]
    assert float(layer.b_ih.abs().max()) >= 0.25           # the kernel's activation code is tanh (not tanh_rel): the fp16 loop
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        rows = 10.0 ** (torch.rand(r, 1, generator=g) * 7 - 6)          # 1e-6 .. 10
        inside = 10.0 ** (-5 * (torch.rand(r, r, generator=g) < 0.3).float() * torch.rand(r, r, generator=g))
        layer.w_hh.data = (layer.w_hh.data * rows * inside).contiguous()
        layer.w_hh.data *= 0.9 / float(torch.linalg.eigvals(layer.w_hh.data).abs().max())
    t = 200
    x = torch.randn(t, n, f)
    out = torch.full((t, n, r), float("nan"), device="cuda")
    res.encode_into(x.cuda(), out)
    idx = torch.as_tensor(sorted({0, 1, 15, 16, n // 2, n - 2, n - 1}))
    ref64 = O.reservoir_forward(x[:, idx], layers_of(res), dtype=torch.float64)
    ref32 = O.reservoir_forward(x[:, idx], layers_of(res))
    got = out[:, idx.cuda()].cpu()
    e_gpu, e_cpu = _per_unit_rel(got, ref64), _per_unit_rel(ref32, ref64)
    assert torch.isfinite(got).all()
    # (the kernel's tanh is accurate to ~1e-7 absolute, sgp_amd.h: units whose state is tiny are held to that)
    floor = 2e-7 / ref64.abs().amax(dim=(0, 1)).clamp_min(1e-30)
    assert bool((e_gpu <= 4 * e_cpu + floor).all()), (float((e_gpu / (4 * e_cpu + floor)).max()), int((e_gpu / (4 * e_cpu + floor)).argmax()))
    assert float(e_gpu.max()) <= 1e-5


def test_split_j_two_piece_fp16_state_leaves_to_the_three_piece_loop_outside_the_unit_interval():
    """A caller's initial state outside [-1, 1] (the fp16 pieces are scaled for |h| <= 1): the workgroups holding such
    nodes run the three-piece bf16 loop (decided in the kernel, per workgroup), the others the fp16 loop -- both as
    close to fp64 as the CPU; a NaN in the initial state stays with its node."""
    hip.require_gpu()
    torch.manual_seed(9)
    n, f, r, t = 325, 3, 128, 80
    res = sgp_amd.Reservoir(f, r, spectral_radius=0.9, leaking_rate=0.7)
    x = torch.randn(t, n, f)
    h0 = torch.rand(1, n, r) * 2 - 1
    h0[0, 40] *= 50.0                                       # node tile 2
    h0[0, 300, 7] = -3.0                                    # node tile 18
    h0[0, 17, 5] = 1.0                                      # exactly on the bound: stays with the fp16 loop
    state = h0.clone().cuda()
    out = torch.full((t, n, r), float("nan"), device="cuda")
    res.encode_into(x.cuda(), out, state)
    idx = torch.as_tensor([0, 17, 32, 40, 47, 48, 299, 300, 324])
    ref64 = O.reservoir_forward(x[:, idx], layers_of(res), h0=h0[:, idx], dtype=torch.float64)
    ref32 = O.reservoir_forward(x[:, idx], layers_of(res), h0=h0[:, idx])
    got = out[:, idx.cuda()].cpu()
    e_gpu = float((got.double() - ref64).abs().max())
    e_cpu = float((ref32.double() - ref64).abs().max())
    assert e_gpu <= 2 * e_cpu + 1e-6, (e_gpu, e_cpu)
    assert torch.equal(state[0], out[-1])
    h0[0, 100, 3] = float("nan")
    state = h0.clone().cuda()
    res.encode_into(x.cuda(), out, state)
    assert bool(torch.isnan(out[:, 100]).any()) and bool(torch.isfinite(out[:, :100]).all()) and bool(torch.isfinite(out[:, 101:]).all())


def test_split_j_two_piece_fp16_loop_is_the_one_that_runs():
    """SGP_TUNE=res_h16=0 (three bf16 pieces for the bounded state too) gives different low bits than the default: the
    fp16 loop is not silently skipped; both stay within the CPU fp32 run's distance from fp64."""
    import os
    import subprocess
    import sys
    hip.require_gpu()
    code = ("import torch, sgp_amd, hashlib, sys\n"
            "torch.manual_seed(4)\n"
            "res = sgp_amd.Reservoir(3, 128, spectral_radius=0.9)\n"
            "x = torch.randn(64, 325, 3)\n"
            "out = torch.empty(64, 325, 128, device='cuda')\n"
            "res.encode_into(x.cuda(), out)\n"
            "print('HASH', hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for tune in ("res_h16=0", "res_h16=1"):
        env = dict(os.environ, SGP_TUNE=tune, PYTHONPATH=root)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        hashes.append([l for l in p.stdout.splitlines() if l.startswith("HASH")][0])
    assert hashes[0] != hashes[1]


@pytest.mark.parametrize("n,f,r", [(16384 * 6 + 16 * 200, 64, 64), (20000, 32, 32), (16 * 1024 * 2 + 16 * 37 + 5, 64, 64),
                                   (2048 * 16 + 16 * 5 + 3, 64, 256), (40000, 32, 256)])    # R = 256: the streamed form (>= 2048 node tiles)
def test_large_n_two_piece_fp16_state_under_the_launch_predicate(n, f, r):
    """The large-N form of the bounded-state loop (reservoir_layer_bf3<.., H16>: recurrent products from two fp16 pieces,
    the row scale folded into bias and input fragments, sgp_amd.h): chosen ON THE DEVICE -- alone when the recurrence
    starts from zero, under the word "some initial state lies outside [-1, 1]" == 0 when a state is handed in, with
    the three-piece instance under == 1 behind it.  Rows of W_hh five orders of magnitude apart; a state inside the
    interval, one with a single entry of 1.5, one with a NaN: each as close to fp64 as the CPU's fp32 run."""
    hip.require_gpu()
    torch.manual_seed(n % 97)
    t = 20
    res = sgp_amd.Reservoir(f, r, spectral_radius=0.9, leaking_rate=0.8)
    layer = res.reservoir_layers[0]
    assert float(layer.b_ih.abs().max()) >= 0.25           # activation code tanh: the fp16 loop is eligible
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        layer.w_hh.data = (layer.w_hh.data * 10.0 ** (torch.rand(r, 1, generator=g) * 5 - 4)).contiguous()
        layer.w_hh.data *= 0.9 / float(torch.linalg.eigvals(layer.w_hh.data).abs().max())
    x = torch.randn(t, n, f)
    xg = x.cuda()
    nodes = sorted({0, 1, 15, 16, 17, n // 3, n // 2, n - 17, n - 16, n - 2, n - 1} | set(range(4096, 4096 + 24)))
    idx = torch.as_tensor(nodes)

    def run(h0):
        state = None if h0 is None else h0.clone().cuda()
        out = torch.full((t, n, r), float("nan"), device="cuda")
        res.encode_into(xg, out, state)
        if state is not None:
            assert torch.equal(state[0], out[-1]) or bool(torch.isnan(state).any())
        return out

    def check_against_fp64(out, h0, skip=()):
        keep = [k for k, v in enumerate(nodes) if v not in skip]
        sel = idx[keep]
        kw = {} if h0 is None else dict(h0=h0[:, sel])
        ref64 = O.reservoir_forward(x[:, sel], layers_of(res), dtype=torch.float64, **kw)
        ref32 = O.reservoir_forward(x[:, sel], layers_of(res), **kw)
        got = out[:, sel.cuda()].cpu()
        assert torch.isfinite(got).all()
        e_gpu = float((got.double() - ref64).abs().max())
        e_cpu = float((ref32.double() - ref64).abs().max())
        assert e_gpu <= 2 * e_cpu + 1e-6, (e_gpu, e_cpu)
        assert O.rel_fro(got, ref32) <= 1e-5

    zero = run(None)
    check_against_fp64(zero, None)
    h_in = torch.rand(1, n, r) * 2 - 1
    inside = run(h_in)
    check_against_fp64(inside, h_in)
    h_out = h_in.clone()
    h_out[0, n // 2, 3] = 1.5                                # one entry: the whole launch takes the three-piece instance
    outside = run(h_out)
    check_against_fp64(outside, h_out)
    # (same data except one node: the two instances differ in the low bits elsewhere -- the predicate switched kernels)
    far = [v for v in nodes if v != n // 2]
    assert not torch.equal(outside[:, far], inside[:, far])
    assert float((outside[:, far] - inside[:, far]).abs().max()) < 1e-5
    h_nan = h_in.clone()
    h_nan[0, 17, 0] = float("nan")
    with_nan = run(h_nan)
    assert bool(torch.isnan(with_nan[:, 17]).any())
    check_against_fp64(with_nan, h_nan, skip=(17,))


@pytest.mark.parametrize("n,f,r", [(325, 3, 128), (20000, 64, 64), (2048 * 16 + 40, 64, 256)])
def test_leaking_rate_outside_the_unit_interval_keeps_three_pieces(n, f, r):
    """A leaking rate of 1.7 (the reference takes any float, reservoir.py:109-123): |h| reaches 1.7 / 0.3 = 5.7 -- outside the
    range the two-piece fp16 state is scaled for.  The library keeps three bf16 pieces there: as close to fp64 as the
    CPU's fp32 run."""
    hip.require_gpu()
    torch.manual_seed(r + 1)
    res = sgp_amd.Reservoir(f, r, spectral_radius=0.5, leaking_rate=1.7)
    t = 60
    x = torch.randn(t, n, f)
    out = torch.full((t, n, r), float("nan"), device="cuda")
    res.encode_into(x.cuda(), out)
    assert float(out.abs().max()) > 1.5                     # the state does leave the unit interval
    check(out, x, res, "tanh", sorted({0, 1, 15, 16, n // 2, n - 2, n - 1}))
