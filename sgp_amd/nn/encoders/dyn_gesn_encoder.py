"""``lib/nn/encoders/dyn_gesn_encoder.py`` -- the DynGESN baseline encoder."""
import torch
from torch import nn

from ... import hip
from ...graph import ShiftOperator
from ..reservoir.graph_reservoir import GraphESN, _as_operator
from ._args import add_reservoir_args


def gesn_operator(edge_index, edge_weight, num_nodes) -> ShiftOperator:
    """dyn_gesn_encoder.py:37-43: unit self loops appended for nodes ``0 .. max(edge_index)``
    (torch_geometric ``add_self_loops`` with ``num_nodes=None``; existing loops are kept, so
    they add up), weights divided by the weighted in-degree of their target
    (``tsl.ops.connectivity.normalize(dim=1)``), rows = targets."""
    if edge_weight is None:
        # tsl's normalize divides ``edge_weights / degree`` (connectivity.py:227)
        raise TypeError("unsupported operand type(s) for /: 'NoneType' and 'Tensor'")
    ei = torch.as_tensor(edge_index).long().cpu()
    ew = torch.as_tensor(edge_weight).float().cpu()
    n_loops = int(ei.max()) + 1 if ei.numel() else 0
    loop = torch.arange(n_loops)
    row = torch.cat([ei[1], loop])
    col = torch.cat([ei[0], loop])
    val = torch.cat([ew, torch.ones(n_loops)])
    return ShiftOperator.from_coo(row, col, val, num_nodes)


class GESNEncoder(nn.Module):
    def __init__(self,
                 input_size,
                 reservoir_size,
                 reservoir_layers,
                 leaking_rate,
                 spectral_radius,
                 density,
                 input_scaling,
                 alpha_decay,
                 reservoir_activation='tanh'
                 ):
        super(GESNEncoder, self).__init__()
        self.reservoir = GraphESN(input_size=input_size,
                                  hidden_size=reservoir_size,
                                  input_scaling=input_scaling,
                                  num_layers=reservoir_layers,
                                  leaking_rate=leaking_rate,
                                  spectral_radius=spectral_radius,
                                  density=density,
                                  activation=reservoir_activation,
                                  alpha_decay=alpha_decay)

    def forward(self, x, edge_index, edge_weight=None):
        # x : [t n f]
        if isinstance(edge_index, ShiftOperator) or (hasattr(edge_index, 'coo')
                                                     and not torch.is_tensor(edge_index)):
            # the reference hands a SparseTensor through without normalising it (:39)
            op = _as_operator(edge_index, None, x.size(-2))
        else:
            op = gesn_operator(edge_index, edge_weight, x.size(-2))
        dev_in = x.device
        xg = x.float()
        if not xg.is_cuda:
            hip.require_gpu()
            xg = xg.cuda()
        T, n, _ = xg.shape
        out = torch.empty(T, n, self.reservoir.n_layers * self.reservoir.hidden_size,
                          device=xg.device)
        self.reservoir.encode_into(xg, op, out)
        return out.to(dev_in)

    @staticmethod
    def add_model_specific_args(parser):
        add_reservoir_args(parser)
        return parser
