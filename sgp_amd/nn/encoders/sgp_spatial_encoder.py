import torch
from torch import nn

from ... import hip
from ...sgp_preprocessing import propagate_into, spatial_operators
from ._args import add_spatial_args


class SGPSpatialEncoder(nn.Module):
    """lib/nn/encoders/sgp_spatial_encoder.py:8-35.  Output feature order:
    ``[x | A x | ... | A^K x | (A_b x | ... | A_b^K x) | (mean_n x)]``."""

    def __init__(self,
                 receptive_field,
                 bidirectional,
                 undirected,
                 global_attr,
                 add_self_loops=False):
        super(SGPSpatialEncoder, self).__init__()
        self.receptive_field = receptive_field
        self.bidirectional = bidirectional
        self.undirected = undirected
        self.add_self_loops = add_self_loops
        self.global_attr = global_attr

    def num_blocks(self):
        dirs = 2 if self.bidirectional else 1
        return 1 + dirs * self.receptive_field + (1 if self.global_attr else 0)

    def operators(self, num_nodes, edge_index, edge_weight):
        return spatial_operators(edge_index, edge_weight, num_nodes,
                                 undirected=self.undirected,
                                 add_self_loops=self.add_self_loops,
                                 bidirectional=self.bidirectional)

    def encode_into(self, out, feat, ops, timeline=None, col_sums=None, x_bound=None):
        """Device path.  ``out[B, N, P * feat]`` with slot 0 already filled: run the hops
        and the global-mean block in place (sgp_spatial_encoder.py:22-35 without the
        ``torch.cat``).  ``col_sums`` [B, feat]: sums over the nodes of slot 0 when the producer
        already has them (``Reservoir.encode_into``): the global block is then written without
        reading slot 0 again.  ``x_bound``: see ``propagate_into``."""
        propagate_into(out, feat, ops, self.receptive_field, timeline, x_bound=x_bound)
        if self.global_attr:          # :32-34
            p = self.num_blocks() - 1
            slot = out[:, :, p * feat:(p + 1) * feat]
            if col_sums is not None:
                hip.bcast_rows(col_sums, 1.0 / out.shape[1], slot)
            else:
                hip.node_mean_bcast(out[:, :, :feat], slot)
        return out

    def forward(self, x, edge_index, edge_weight):
        num_nodes = x.size(-2)
        ops = self.operators(num_nodes, edge_index, edge_weight)
        dev = x.device
        xg = x.float()
        squeeze = xg.dim() == 2
        if squeeze:
            xg = xg[None]
        if not xg.is_cuda:
            hip.require_gpu()
            xg = xg.cuda()
        B, N, F = xg.shape
        out = torch.empty(B, N, self.num_blocks() * F, dtype=torch.float32, device=xg.device)
        hip.copy_rows(xg if xg.stride(2) == 1 else xg.contiguous(), out[:, :, :F])
        self.encode_into(out, F, ops)
        if squeeze:
            out = out[0]
        return out.to(dev)

    @staticmethod
    def add_model_specific_args(parser):
        add_spatial_args(parser)
        return parser
