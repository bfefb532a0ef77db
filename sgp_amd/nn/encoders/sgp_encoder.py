import torch
from torch import nn

from ... import hip
from ..reservoir import Reservoir
from ._args import add_reservoir_args, add_spatial_args
from .sgp_spatial_encoder import SGPSpatialEncoder


class SGPEncoder(nn.Module):
    """SGP's training-free spatiotemporal encoder, ``lib/nn/encoders/sgp_encoder.py:9-51``:
    reservoir over time, then K-hop graph-shift propagation over nodes.

    The device path allocates the final ``[T, N, D_out]`` tensor once; the reservoir
    writes its states straight into block 0, every hop reads one block and writes the next
    and the global-mean block is filled last -- the ``stack``/``rearrange``/``cat`` copies of
    the reference (reservoir.py:178-183, sgp_spatial_encoder.py:35) never happen.
    """

    def __init__(self,
                 input_size,
                 reservoir_size,
                 reservoir_layers,
                 leaking_rate,
                 spectral_radius,
                 density,
                 input_scaling,
                 receptive_field,
                 bidirectional,
                 alpha_decay,
                 global_attr,
                 add_self_loops=False,
                 undirected=False,
                 reservoir_activation='tanh'):
        super(SGPEncoder, self).__init__()
        self.reservoir = Reservoir(input_size=input_size,
                                   hidden_size=reservoir_size,
                                   input_scaling=input_scaling,
                                   num_layers=reservoir_layers,
                                   leaking_rate=leaking_rate,
                                   spectral_radius=spectral_radius,
                                   density=density,
                                   activation=reservoir_activation,
                                   alpha_decay=alpha_decay)
        self.sgp_encoder = SGPSpatialEncoder(
            receptive_field=receptive_field,
            bidirectional=bidirectional,
            undirected=undirected,
            add_self_loops=add_self_loops,
            global_attr=global_attr)

    @property
    def output_size(self):
        return self.sgp_encoder.num_blocks() * self.reservoir.output_size

    def encode_device(self, x, ops, out=None):
        """x[T, N, F] CUDA float32 -> out[T, N, D_out] on the same device."""
        T, N, _ = x.shape
        d_h = self.reservoir.output_size
        if out is None:
            out = torch.empty(T, N, self.output_size, dtype=torch.float32, device=x.device)
        self.reservoir.encode_into(x, out[:, :, :d_h])
        self.sgp_encoder.encode_into(out, d_h, ops)
        return out

    # Device-memory budget for one pass (bytes); None = 80 % of what is free right now.  Host
    # inputs whose input + embedding exceed it are encoded in time chunks (see encode_streamed).
    max_device_bytes = None

    def _budget(self):
        if self.max_device_bytes is not None:
            return int(self.max_device_bytes)
        free, _ = torch.cuda.mem_get_info()
        return int(0.8 * free)

    def encode_streamed(self, x, ops, t_chunk):
        """Host tensor x[T, N, F] -> host tensor [T, N, D_out], ``t_chunk`` steps at a time.
        The recurrence is carried across chunks in a device-resident state ``[L, N, R]``
        (``sgp_reservoir_f32``'s h_state); the propagation is independent per time step, so
        the result is bit-identical to a single pass.  This is how embeddings larger than the
        288 GB of HBM (BASELINE config C5: 629 GB) or than the free memory are produced."""
        hip.require_gpu()
        T, N, _ = x.shape
        dev = torch.device("cuda", torch.cuda.current_device())
        L, R = len(self.reservoir.reservoir_layers), self.reservoir.hidden_size
        d_h = L * R
        state = torch.zeros(L, N, R, dtype=torch.float32, device=dev)
        out = torch.empty(T, N, self.output_size, dtype=torch.float32)
        buf = torch.empty(min(t_chunk, T), N, self.output_size, dtype=torch.float32, device=dev)
        for t0 in range(0, T, t_chunk):
            t1 = min(T, t0 + t_chunk)
            xc = x[t0:t1].to(dev, torch.float32, non_blocking=True).contiguous()
            oc = buf[:t1 - t0]
            self.reservoir.encode_into(xc, oc[:, :, :d_h], state)
            self.sgp_encoder.encode_into(oc, d_h, ops)
            out[t0:t1].copy_(oc)
        return out

    def forward(self, x, edge_index, edge_weight):
        # x : [t n f]
        dev = x.device
        ops = self.sgp_encoder.operators(x.size(-2), edge_index, edge_weight)
        xg = x.float()
        if not xg.is_cuda:
            hip.require_gpu()
            T, N, F = xg.shape
            per_step = N * (F + self.output_size) * 4
            budget = self._budget()
            if T * per_step > budget:
                t_chunk = max(1, budget // per_step)
                return self.encode_streamed(xg, ops, t_chunk)
            xg = xg.cuda()
        if xg.stride(2) != 1:
            xg = xg.contiguous()
        return self.encode_device(xg, ops).to(dev)

    @staticmethod
    def add_model_specific_args(parser):
        add_reservoir_args(parser)
        add_spatial_args(parser)
        return parser
