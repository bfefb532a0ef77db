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

from ... import hip
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
        return hip.reservoir_layer(x, w_ih, w_hh, b, self.alpha, self.kernel_activation(),
                                   out, h_state)

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
