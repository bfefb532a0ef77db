"""DynGESN -- the graph echo-state baseline (``lib/nn/reservoir/graph_reservoir.py``) with the
reference's Python surface and two HIP launches per (time step, layer) underneath.

    h' = (1 - a) h + a * act( x W_ih^T + b + A_hat (h W_hh^T) )          (graph_reservoir.py:85-93)

The graph product sits inside the recurrence, so -- unlike SGP -- nothing can be hoisted out of
the time loop except the first layer's input term (one GEMM per 256 steps).
"""
import numpy as np
import torch
import torch.nn as nn

from ... import hip
from ...graph import ShiftOperator
from .reservoir import ReservoirLayer


class GESNLayer(ReservoirLayer):
    """Same parameters and RNG order as ``ReservoirLayer`` (graph_reservoir.py:56-78 repeats
    reservoir.py:54-75); ``aggr`` is accepted for signature parity, only 'add' exists here."""

    def __init__(self,
                 input_size,
                 hidden_size,
                 spectral_radius=0.9,
                 leaking_rate=0.9,
                 bias=True,
                 density=0.9,
                 in_scaling=1.,
                 bias_scale=1.,
                 activation='tanh',
                 aggr='add'):
        if aggr != 'add':
            raise NotImplementedError("GESNLayer: only aggr='add' is implemented")
        super(GESNLayer, self).__init__(input_size, hidden_size, spectral_radius, leaking_rate,
                                        bias=bias, density=density, in_scaling=in_scaling,
                                        bias_scale=bias_scale, activation=activation)
        self.aggr = aggr

    def step(self, p, h, op_dev, z, h_out, out_rows, w_hh):
        """One update given the input term ``p = x W_ih^T + b`` ([N, R])."""
        hip.gemm_nt(h, w_hh, None, z)
        return hip.gesn_update(*op_dev, z, p, h, self.alpha, self.activation_name, h_out, out_rows)

    def forward(self, x, h, edge_index, edge_weight=None):
        """x[..., N, F], h[..., N, R] -> h'[..., N, R]; ``edge_index`` is a normalised operator
        (``ShiftOperator`` / SparseTensor-like, rows = targets) or a raw edge list."""
        op = _as_operator(edge_index, edge_weight, x.shape[-2])
        dev_in = x.device
        xg = x.reshape(-1, x.shape[-2], x.shape[-1]).float()
        hg = h.reshape(-1, h.shape[-2], self.hidden_size).float()
        if not xg.is_cuda:
            hip.require_gpu()
            xg, hg = xg.cuda(), hg.cuda()
        dev = xg.device
        w_ih, w_hh, b = self._device_weights(dev)
        op_dev = op.device_csr(dev)
        n, r = xg.shape[1], self.hidden_size
        out = torch.empty(xg.shape[0], n, r, device=dev)
        p, z = torch.empty(n, r, device=dev), torch.empty(n, r, device=dev)
        for i in range(xg.shape[0]):
            hip.gemm_nt(xg[i].contiguous(), w_ih, b, p)
            self.step(p, hg[i].contiguous(), op_dev, z, out[i], out[i], w_hh)
        return out.reshape(*x.shape[:-1], r).to(dev_in)


def _as_operator(edge_index, edge_weight, num_nodes):
    if isinstance(edge_index, ShiftOperator):
        return edge_index
    if hasattr(edge_index, 'coo') and not torch.is_tensor(edge_index):
        row, col, val = edge_index.coo()
        if val is None:
            val = torch.ones(row.numel())
        return _raw_operator(row.long().cpu(), col.long().cpu(), val.float().cpu(), num_nodes)
    ei = torch.as_tensor(edge_index).long().cpu()
    ew = (torch.ones(ei.shape[1]) if edge_weight is None
          else torch.as_tensor(edge_weight).float().cpu())
    return _raw_operator(ei[1], ei[0], ew, num_nodes)     # flow source -> target


def _raw_operator(row, col, val, n):
    """CSR of the given weights as they are (duplicates summed, no normalisation)."""
    key = row * n + col
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    v = torch.zeros(uniq.numel(), dtype=torch.float32).index_add_(0, inv, val)
    r, c = uniq // n, uniq % n
    rowptr = torch.zeros(n + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return ShiftOperator(rowptr, c, v, n)


class GraphESN(nn.Module):
    """``graph_reservoir.py:96-146`` on top of tsl's ``_GraphRNN`` loop
    (``tsl/nn/blocks/encoders/gcrnn.py:44-93``, ``_cat_states_layers = True``)."""
    _cat_states_layers = True

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
        super(GraphESN, self).__init__()
        self.mode = activation
        self.input_size = input_size
        self.input_scaling = input_scaling
        self.hidden_size = hidden_size
        self.n_layers = num_layers
        self.leaking_rate = leaking_rate
        self.spectral_radius = spectral_radius
        self.density = density
        self.bias = bias
        self.alpha_decay = alpha_decay

        layers = []
        alpha = leaking_rate
        for i in range(num_layers):
            layers.append(
                GESNLayer(
                    input_size=input_size if i == 0 else hidden_size,
                    hidden_size=hidden_size,
                    in_scaling=input_scaling,
                    density=density,
                    activation=activation,
                    spectral_radius=spectral_radius,
                    leaking_rate=alpha))
            if self.alpha_decay:
                alpha = np.clip(alpha - 0.1, 0.1, 1.)
        self.rnn_cells = nn.ModuleList(layers)
        self.reset_parameters()        # graph_reservoir.py:140 draws every layer a second time

    def reset_parameters(self):
        for layer in self.rnn_cells:
            layer.reset_parameters()

    def encode_into(self, x, op, out, h=None):
        """x[T, N, F] (CUDA) -> out[T, N, L*R]; ``h`` [L, N, R] initial states, updated in
        place to the final ones.  One C call: ``sgp_gesn_f32`` issues the 2 launches per
        (step, layer) itself."""
        T, n, _ = x.shape
        r, L = self.hidden_size, self.n_layers
        dev = x.device
        assert out.shape == (T, n, L * r) and out.stride(2) == 1
        weights = [c._device_weights(dev) for c in self.rnn_cells]
        if h is None:
            h = torch.zeros(L, n, r, device=dev)
        if x.stride(2) != 1:
            x = x.contiguous()
        hip.gesn_sequence(*op.device_csr(dev), x, weights, [c.alpha for c in self.rnn_cells],
                          self.rnn_cells[0].activation_name, out, h)
        return out, h

    def forward(self, x, edge_index, edge_weight=None, h=None):
        """x[B, T, N, F] -> (out[B, T, N, L*R], h[L, B, N, R]) as ``_GraphRNN.forward``."""
        dev_in = x.device
        B, T, n, _ = x.shape
        op = _as_operator(edge_index, edge_weight, n)
        xg = x.float()
        if not xg.is_cuda:
            hip.require_gpu()
            xg = xg.cuda()
        dev = xg.device
        L, r = self.n_layers, self.hidden_size
        out = torch.empty(B, T, n, L * r, device=dev)
        h_last = torch.empty(L, B, n, r, device=dev)
        for bi in range(B):
            h0 = None
            if h is not None:
                h0 = torch.stack([hi[bi] for hi in h]).to(dev, torch.float32).contiguous().clone()
            _, hl = self.encode_into(xg[bi], op, out[bi], h0)
            h_last[:, bi] = hl
        return out.to(dev_in), h_last.to(dev_in)
