"""Leaky echo-state reservoir with the reference's Python surface
(``lib/nn/reservoir/reservoir.py``) and a HIP time loop underneath.

Weights are drawn on the host from the global torch RNG in exactly the
reference's order (``reservoir.py:54-75``) so a seed reproduces the reference's
parameters; the recurrence itself runs in ``sgp_reservoir_f32`` (one launch per
layer, the whole sequence on the device).
"""
import numpy as np
import torch
import torch.nn as nn

from ... import hip, tune
from .init import draw_reservoir_weights

_ACTIVATIONS = ['tanh', 'relu', 'self_norm', 'identity']


def _check_activation(activation):
    # reservoir.py:37 asserts membership, then :41 resolves the name through
    # tsl.nn.utils.get_functional_activation (tsl/nn/utils/utils.py:34-44),
    # which knows 'linear' but not 'identity' and raises ValueError for it.
    assert activation in _ACTIVATIONS
    if activation == 'identity':
        raise ValueError(f"Activation '{activation}' not valid.")


class ReservoirLayer(nn.Module):
    def __init__(self,
                 input_size,
                 hidden_size,
                 spectral_radius,
                 leaking_rate,
                 bias=True,
                 density=1.,
                 in_scaling=1.,
                 bias_scale=1.,
                 activation='tanh'):
        super(ReservoirLayer, self).__init__()
        self.w_ih_scale = in_scaling
        self.b_scale = bias_scale
        self.density = density
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.alpha = leaking_rate
        self.spectral_radius = spectral_radius
        _check_activation(activation)
        self.activation_name = activation

        self.w_ih = nn.Parameter(torch.empty(hidden_size, input_size), requires_grad=False)
        self.w_hh = nn.Parameter(torch.empty(hidden_size, hidden_size), requires_grad=False)
        if bias is not None:      # reservoir.py:47 -- bias=False still creates it
            self.b_ih = nn.Parameter(torch.empty(hidden_size), requires_grad=False)
        else:
            self.register_parameter('b_ih', None)
        self.reset_parameters()

    def reset_parameters(self):
        """Re-draw the layer from the global torch RNG (``init.draw_reservoir_weights`` keeps the
        reference's draw order, so a seed reproduces the reference's parameters)."""
        w_in, bias, w_rec = draw_reservoir_weights(
            self.input_size, self.hidden_size, density=self.density,
            spectral_radius=self.spectral_radius, in_scaling=self.w_ih_scale,
            bias_scale=self.b_scale, with_bias=self.b_ih is not None)
        self.w_ih.data.copy_(w_in)
        self.w_hh.data.copy_(w_rec)
        if bias is not None:
            self.b_ih.data.copy_(bias)
        self._act_key = None

    def _device_weights(self, device):
        b = self.b_ih if self.b_ih is not None else torch.zeros(self.hidden_size)
        return tuple(w.detach().to(device=device, dtype=torch.float32).contiguous()
                     for w in (self.w_ih, self.w_hh, b))

    def kernel_activation(self):
        """Activation code for the device kernels.  Their default tanh is accurate to 3e-7 ABSOLUTE (6 instructions per
        value) -- as good as fp32 for states of order 1, which the reference's bias ~U(-1, 1) guarantees.  A layer whose
        bias is tiny (scaled down by the caller: max |b| < 0.25) may hold states far below 1, which the reference's tanh
        resolves to fp32's RELATIVE accuracy: such layers run with ``tanh_rel`` (odd polynomial below 0.25, +4-15 % time)."""
        if self.activation_name != "tanh":
            return self.activation_name
        b = self.b_ih
        if b is None:
            return "tanh_rel"
        if not b.is_cuda:
            # host weights (the default: the reference's modules live on the CPU): looked at on every call -- a few
            # hundred floats, and `.data` edits do not bump a version counter
            return "tanh_rel" if float(b.detach().abs().max()) < 0.25 else "tanh"
        # a module moved to the GPU: the read is a device-to-host sync on the current stream -- once per layer and time
        # piece in front of the chain's launches, where the host then waits for the previous piece and enqueues the
        # hops late.  Decided once per (storage, version) of the bias instead; weights set through copy_ / load_state_dict
        # / reset_parameters / .to() change that key, an edit through ``.data`` of a GPU-resident bias needs
        # ``layer._act_key = None``.
        key = (b.data_ptr(), b._version)
        if getattr(self, "_act_key", None) != key:
            self._act_key, self._act_name = key, ("tanh_rel" if float(b.detach().abs().max()) < 0.25 else "tanh")
        return self._act_name

    def run_sequence(self, x, out, h_state=None):
        """x[T, M, F] -> out[T, M, R] on the device (strided views allowed)."""
        w_ih, w_hh, b = self._device_weights(x.device)
        act = self.kernel_activation()
        self.last_time_parallel = None
        plan = self.time_parallel_plan(x.shape[0], x.shape[1], x.shape[2], x.device, act) if x.is_cuda else None
        if plan is not None:
            try:
                return self._run_time_parallel(x, out, h_state, plan, (w_ih, w_hh, b), act)
            except NotImplementedError:
                pass                                      # (a shape the piece kernel does not serve: one chain)
        return hip.reservoir_layer(x, w_ih, w_hh, b, self.alpha, act, out, h_state)

    # ---- small graphs: time pieces side by side (reservoir.py:170-183 is a serial chain of T steps; at N = 207 / 325 it
    # occupies 13 / 21 of the chip's 256 compute units)
    # Steps a piece is started early, from zero (None: from the leaky recurrence's nominal contraction rate (1 - a) + a rho,
    # enough for 1e-7 -- 192 steps at rho = a = 0.9; measured there: the gap is at its floor after 64).
    time_parallel_warm = None
    # A splice is accepted when the warmed-up state is this close (max-abs) to the true one.  Two fp32 evaluations of the
    # same contractive recurrence from different starts do not meet exactly: their gap settles at the rounding floor,
    # measured 3.0e-7 .. 4.4e-7 on the C1 / C2 shapes for every warm-up from 64 to 512 steps (profiles/r6/
    # time_parallel_probe.log) -- the distance of either from the fp64 trajectory is 5e-7.  1e-6 = 2.5x that floor,
    # a tenth of the encoder's 1e-5.
    time_parallel_tol = 1e-6
    time_parallel_wgs_per_cu = 2  # piece workgroups per compute unit (SGP_TUNE=time_parallel_wgs; measured C1 2.78 -> 2.65 ms, C2 unchanged)

    def time_parallel_plan(self, T, N, F, device, act=None):
        """``(pieces, steps per piece, warm-up steps)`` when this layer's sequence may be cut into time pieces that run
        side by side, else None.  Premises: a CONTRACTIVE recurrence -- tanh with a leaking rate in [0, 1]: a piece
        started ``warm`` steps early from zero then arrives at the true state (the echo-state property the reference's
        spectral-radius rescaling is there to give, reservoir.py:74-75) -- and a shape the piece kernel serves.  Whether a
        given reservoir really forgets fast enough is NOT assumed: every splice is checked on the device and a miss reruns
        the sequential chain (``_run_time_parallel``).  relu / identity / self_norm, leaking rates outside [0, 1] and
        layers with tiny biases (``tanh_rel``: states far below the absolute tolerance) never take this path."""
        act = act or self.kernel_activation()
        if act != "tanh" or not (0.0 <= float(self.alpha) <= 1.0) or tune.get("time_parallel", 1, int) == 0:
            return None
        R = self.hidden_size
        tiles = (N + 15) // 16
        if not (32 < R <= 128 and F <= 64) or tiles < 1:
            return None
        warm = tune.get("time_parallel_warm", self.time_parallel_warm or 0, int)
        if warm <= 0:
            rate = (1.0 - float(self.alpha)) + float(self.alpha) * float(self.spectral_radius)
            if not (0.0 < rate < 0.999):
                return None                               # nominally not contractive (or so slowly that no warm-up pays)
            warm = min(4096, -(-int(np.ceil(np.log(1e-7) / np.log(rate))) // 64) * 64)
        cus = torch.cuda.get_device_properties(device).multi_processor_count if torch.cuda.is_available() else 256
        per_cu = tune.get("time_parallel_wgs", self.time_parallel_wgs_per_cu, int)
        pieces = min(per_cu * cus // tiles, T // (2 * warm))   # workgroups per compute unit; a piece >= 2 warm-ups long
        if pieces < 2:
            return None
        steps = -(-T // pieces)
        pieces = -(-T // steps)                           # (the last piece may be shorter, never empty)
        return (pieces, steps, warm) if pieces >= 2 else None

    def _run_time_parallel(self, x, out, h_state, plan, weights, act):
        """Four launches on the caller's stream, no host round trip: (1) warm-ups -- pieces 1 .. P-1 run ``warm`` steps in
        front of their cut from a zero state and leave only their final state; (2) all P pieces from those states
        (piece 0: the caller's); (3) the splice test -- every piece's end state against its successor's warmed-up start,
        max-abs <= tol (NaN fails) -> a device flag; (4) the sequential layer under ``flag == 0``: it runs, and repairs
        every row, only if a splice was rejected.  An accepted result differs from the sequential one by at most tol at
        a cut, contracting from there -- two orders below the encoder's 1e-5."""
        P, S, W = plan
        T, N, _ = x.shape
        R = self.hidden_size
        w_ih, w_hh, b = weights
        init = torch.zeros(P, N, R, dtype=torch.float32, device=x.device)
        if h_state is not None:
            init[0].copy_(h_state)
        hip.reservoir_pieces(x[S - W:], w_ih, w_hh, b, self.alpha, act, out, init[1:], W, W, S * x.stride(0), 0,
                             no_store=True)
        end = init.clone()
        hip.reservoir_pieces(x, w_ih, w_hh, b, self.alpha, act, out, end, S, T - (P - 1) * S, S * x.stride(0),
                             S * out.stride(0))
        gap = (end[:-1] - init[1:]).abs().amax()
        tol = tune.get("time_parallel_tol", self.time_parallel_tol, float)
        flag = (gap <= tol).to(torch.int32).reshape(1)
        hip.reservoir_pieces(x, w_ih, w_hh, b, self.alpha, act, out, h_state, T, T, 0, 0, pred=(flag, 0))
        if h_state is not None:
            was_bounded = hip.is_unit_bounded(h_state)
            h_state.copy_(torch.where(flag.bool(), end[P - 1], h_state))
            if was_bounded:                               # (a tanh state from a state inside [-1, 1]: the copy is an in-place
                hip.mark_unit_bounded(h_state)            # edit only to the version counter the mark is tied to)
        self.last_time_parallel = dict(pieces=P, steps=S, warm=W, flag=flag, gap=gap)
        return out

    def forward(self, x, h):
        """One step (reservoir.py:77-81): a length-1 sequence with initial state h."""
        dev = x.device
        xg = x.reshape(1, -1, x.shape[-1]).float()
        hg = h.reshape(-1, self.hidden_size).float().contiguous().clone()
        if not xg.is_cuda:
            hip.require_gpu()
            xg, hg = xg.cuda(), hg.cuda()
        out = torch.empty(1, xg.shape[1], self.hidden_size, device=xg.device)
        self.run_sequence(xg.contiguous(), out, hg)
        return out[0].reshape(*x.shape[:-1], self.hidden_size).to(dev)


class Reservoir(nn.Module):
    def __init__(self,
                 input_size,
                 hidden_size,
                 input_scaling=1.,
                 num_layers=1,
                 leaking_rate=0.9,
                 spectral_radius=0.9,
                 density=0.9,
                 activation='tanh',
                 bias=True,
                 alpha_decay=False):
        super(Reservoir, self).__init__()
        self.mode = activation
        self.input_size = input_size
        self.input_scaling = input_scaling
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.leaking_rate = leaking_rate
        self.spectral_radius = spectral_radius
        self.density = density
        self.bias = bias
        self.alpha_decay = alpha_decay

        layers = []
        alpha = leaking_rate
        for i in range(num_layers):
            layers.append(
                ReservoirLayer(
                    input_size=input_size if i == 0 else hidden_size,
                    hidden_size=hidden_size,
                    in_scaling=input_scaling,
                    density=density,
                    activation=activation,
                    spectral_radius=spectral_radius,
                    leaking_rate=alpha))
            if self.alpha_decay:        # reservoir.py:122-123
                alpha = np.clip(alpha - 0.1, 0.1, 1.)
        self.reservoir_layers = nn.ModuleList(layers)

    def reset_parameters(self):
        for layer in self.reservoir_layers:
            layer.reset_parameters()

    @property
    def output_size(self):
        return len(self.reservoir_layers) * self.hidden_size

    def encode_into(self, x, out, h_state=None, col_sums=None):
        """Device path: x[T, M, F] (CUDA) -> out[T, M, L*R] view, layer-major features
        (reservoir.py:181-183).  Layer l > 0 consumes layer l-1's slot of ``out`` -- at
        step s its input is layer l-1's NEW state of step s, as in reservoir.py:174-176.
        ``h_state``: optional [L, M, R] carried across time chunks.  ``col_sums`` [T, L*R]: filled
        with the sums over the M rows of every step's states (the global_attr block is their mean);
        the fused kernel produces them from its registers, the other kernels by a pass over ``out``."""
        R = self.hidden_size
        L = len(self.reservoir_layers)
        if self.fused and L > 1 and self._fusable(x):
            # all layers in one launch, pipelined as a wavefront (reservoir.py:170-180 steps every
            # layer inside one time step)
            weights = [layer._device_weights(x.device) for layer in self.reservoir_layers]
            hip.reservoir_stack(x, weights, [layer.alpha for layer in self.reservoir_layers],
                                "tanh_rel" if any(l.kernel_activation() == "tanh_rel" for l in self.reservoir_layers)
                                else self.reservoir_layers[0].activation_name, out[:, :, :L * R], h_state,
                                col_sums=col_sums)
            return out
        src = x
        for i, layer in enumerate(self.reservoir_layers):
            dst = out[:, :, i * R:(i + 1) * R]
            layer.run_sequence(src, dst, None if h_state is None else h_state[i])
            src = dst
        if col_sums is not None:
            col_sums.copy_(hip.node_sums(out[:, :, :L * R]))
        return out

    fused = True                    # set False to force one launch per layer

    def time_parallel(self, T, N, device):
        """True when every layer of a [T, N, .] sequence runs as time pieces side by side (``ReservoirLayer.
        time_parallel_plan``): the chain then fills the chip by itself and is an order of magnitude shorter, so the
        encoder does not pipeline it against the hops."""
        if self.fused and len(self.reservoir_layers) > 1 and self.hidden_size * len(self.reservoir_layers) <= 256 and \
                hip.reservoir_fused_supported(self.input_size, self.hidden_size, len(self.reservoir_layers)):
            return False                                   # (the stacked kernel serves narrow multi-layer reservoirs)
        return all(l.time_parallel_plan(T, N, self.input_size if i == 0 else self.hidden_size, device) is not None
                   for i, l in enumerate(self.reservoir_layers))

    def produces_col_sums(self, x):
        """True when ``encode_into(..., col_sums=)`` gets the sums from the kernel's registers (the
        fused stacked kernel) rather than from a second pass over the states."""
        return bool(self.fused and len(self.reservoir_layers) > 1 and x.is_cuda and self._fusable(x))

    def _fusable(self, x):
        """Narrow stacked reservoirs (R * L <= 256: the shipped sgp_pv.yaml has 16 x 8) whose
        layer-by-layer form is a chain of L x T latency-bound steps; wide layers at large N are
        throughput-bound and keep the per-layer kernel."""
        R, L = self.hidden_size, len(self.reservoir_layers)
        if R * L > 256 or not hip.reservoir_fused_supported(self.input_size, R, L):
            return False
        return R <= 32 or x.shape[1] <= 16384

    def forward(self, x, h0=None, return_last_state=False):
        # x : b s n f   (reservoir.py:158-186)
        batch_size, steps, nodes, _ = x.size()
        dev = x.device
        xg = x.float()
        if not xg.is_cuda:
            hip.require_gpu()
            xg = xg.cuda()
        # 'b s n f -> s (b n) f'
        xs = xg.permute(1, 0, 2, 3).reshape(steps, batch_size * nodes, -1).contiguous()
        L, R = len(self.reservoir_layers), self.hidden_size
        state = None
        if h0 is not None:
            state = h0.to(xs.device, torch.float32).reshape(L, batch_size * nodes, R).contiguous().clone()
        out = torch.empty(steps, batch_size * nodes, L * R, device=xs.device)
        self.encode_into(xs, out, state)
        # 's (b n) (l f) -> b s n (l f)'
        out = out.reshape(steps, batch_size, nodes, L * R).permute(1, 0, 2, 3)
        if return_last_state:
            return out[:, -1].to(dev)
        return out.contiguous().to(dev)

    def forward_prealloc(self, x, h0=None, return_last_state=False):
        """The reference's ``forward_prealloc`` (reservoir.py:131-156) is dead code that
        reads a not-yet-written slot as the previous state; the name is kept and mapped to
        the correct recurrence."""
        if x.dim() == 3:
            return self.forward(x[None], h0, return_last_state)[0]
        return self.forward(x, h0, return_last_state)
