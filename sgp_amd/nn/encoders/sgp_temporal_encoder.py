import torch
from torch import nn, Tensor

from ... import hip
from ..reservoir import Reservoir
from ._args import add_reservoir_args, add_spatial_args


class SGPTemporalEncoder(nn.Module):
    """lib/nn/encoders/sgp_temporal_encoder.py:8-34: reservoir only, graph ignored."""

    def __init__(self, input_size,
                 reservoir_size=32,
                 reservoir_layers=1,
                 leaking_rate=0.9,
                 spectral_radius=0.9,
                 density=0.7,
                 input_scaling=1.,
                 alpha_decay=False,
                 reservoir_activation='tanh'):
        super(SGPTemporalEncoder, self).__init__()
        self.reservoir = Reservoir(input_size=input_size,
                                   hidden_size=reservoir_size,
                                   input_scaling=input_scaling,
                                   num_layers=reservoir_layers,
                                   leaking_rate=leaking_rate,
                                   spectral_radius=spectral_radius,
                                   density=density,
                                   activation=reservoir_activation,
                                   alpha_decay=alpha_decay)

    def forward(self, x: Tensor, *args, **kwargs):
        # x : [t n f]
        dev = x.device
        xg = x.float()
        if not xg.is_cuda:
            hip.require_gpu()
            xg = xg.cuda()
        if xg.stride(2) != 1:
            xg = xg.contiguous()
        out = torch.empty(xg.shape[0], xg.shape[1], self.reservoir.output_size,
                          dtype=torch.float32, device=xg.device)
        self.reservoir.encode_into(xg, out)
        return out.to(dev)

    @staticmethod
    def add_model_specific_args(parser):
        add_reservoir_args(parser)
        add_spatial_args(parser)   # "for sgp spatial preprocessing", sgp_temporal_encoder.py:53-63
        return parser
