"""Graph-shift propagation with the reference's call surface
(``lib/sgp_preprocessing.py``): ``preprocess_adj``, ``sgp_spatial_embedding`` and the
legacy one-call helpers, computing on the MI355X through ``libsgp_amd.so``."""
from typing import List, Optional, Union

import numpy as np
import torch
from torch import Tensor

from . import hip
from .graph import ShiftOperator
from .nn.reservoir import Reservoir


def ensure_list(value):
    # tsl/utils/python_utils.py:5-10
    if hasattr(value, '__iter__') and not isinstance(value, str):
        return list(value)
    return [value]


def _is_sparse_like(obj):
    return isinstance(obj, ShiftOperator) or (hasattr(obj, 'csr') and hasattr(obj, 'coo')
                                              and not torch.is_tensor(obj))


def _operator_from_sparse(adj, set_diag, remove_diag, gcn_norm) -> ShiftOperator:
    """A ready sparse object in row = target layout (sgp_preprocessing.py:83-84)."""
    row, col, val = adj.coo()
    n = adj.size(0)
    if val is None:
        val = torch.ones(row.numel(), dtype=torch.float32)
    return ShiftOperator.from_coo(row.long().cpu(), col.long().cpu(), val.float().cpu(), n,
                                  gcn_norm=gcn_norm, set_diag=set_diag, remove_diag=remove_diag)


def preprocess_adj(edge_index, edge_weight=None, num_nodes: Optional[int] = None,
                   gcn_norm: bool = False, set_diag: bool = True,
                   remove_diag: bool = False) -> ShiftOperator:
    """lib/sgp_preprocessing.py:67-105.  Returns a :class:`ShiftOperator` (CSR on the
    host; ``op @ x`` runs on the GPU) in place of a ``torch_sparse.SparseTensor``."""
    if isinstance(edge_index, (np.ndarray, Tensor)):
        return ShiftOperator.from_edges(edge_index, edge_weight, num_nodes, gcn_norm=gcn_norm,
                                        set_diag=set_diag, remove_diag=remove_diag)
    if _is_sparse_like(edge_index):
        return _operator_from_sparse(edge_index, set_diag, remove_diag, gcn_norm)
    raise RuntimeError("Edge index must be (edge_index, edge_weight) tuple "
                       "or SparseTensor.")


def spatial_operators(edge_index, edge_weight, num_nodes, undirected=False,
                      add_self_loops=False, remove_self_loops=False, bidirectional=False):
    """Forward (and backward) operators of sgp_spatial_embedding (:182-192, :205-216)."""
    if undirected:
        assert bidirectional is False
    if _is_sparse_like(edge_index):
        if undirected or bidirectional:
            raise NotImplementedError("undirected/bidirectional need an edge list")
        return [_operator_from_sparse(edge_index, add_self_loops, remove_self_loops, False)]
    if not isinstance(edge_index, (np.ndarray, Tensor)):
        raise RuntimeError("Edge index must be (edge_index, edge_weight) tuple "
                           "or SparseTensor.")
    ops = [ShiftOperator.from_edges(edge_index, edge_weight, num_nodes, gcn_norm=undirected,
                                    set_diag=add_self_loops, remove_diag=remove_self_loops,
                                    undirected=undirected)]
    if bidirectional:
        ops.append(ShiftOperator.from_edges(edge_index, edge_weight, num_nodes, gcn_norm=False,
                                            set_diag=add_self_loops,
                                            remove_diag=remove_self_loops, transpose=True))
    return ops


def propagate_into(out, feat, ops, k, timeline=None, x_bound=None):
    """Fill hop slots of ``out[B, N, (1 + len(ops) * k) * feat]`` in place: slot 0 must
    already hold x; slot 1 + d*k + (h-1) receives ops[d]^h x.  No concatenation and no
    temporaries: every hop reads one slot and writes the next.  ``timeline``: a list that
    receives one (start, end) pair of ``hip.Event`` per hop launch, recorded on the stream the
    hop runs on (bench.py's roofline timing).  ``x_bound`` >= max |slot 0| where the caller knows it
    (bounded reservoir activations); the split-fp16 hop derives its per-column scales from it, every hop hands the
    next one per-column bounds (``op.next_bound`` = bound x the operator's infinity norm, on the device), and an
    unknown bound is measured by the first hop of a direction."""
    for d, op in enumerate(ops):
        src = out[:, :, 0:feat]
        bound = x_bound
        for h in range(k):
            s = 1 + d * k + h
            dst = out[:, :, s * feat:(s + 1) * feat]
            if timeline is not None:
                from . import hip
                a, b = hip.Event(), hip.Event()
                a.record()
            op.propagate(src, dst, x_bound=bound)
            bound = getattr(op, "next_bound", None)
            if timeline is not None:
                b.record()
                timeline.append((a, b))
            src = dst
    return out


def sgp_spatial_embedding(x,
                          num_nodes,
                          edge_index,
                          edge_weight=None,
                          k=2,
                          undirected=False,
                          add_self_loops=False,
                          remove_self_loops=False,
                          bidirectional=False,
                          one_hot_encoding=False,
                          dropout_rate=0.):
    """lib/sgp_preprocessing.py:163-218: ``[x, A x, ..., A^k x (, A_b x, ..., A_b^k x)]``
    as a list of views into one fused ``[B, N, P * F]`` buffer."""
    if dropout_rate < 0. or dropout_rate > 1.:
        raise ValueError(f"Dropout probability has to be between 0 and 1 (got {dropout_rate})")
    if dropout_rate != 0.:
        # torch_geometric.utils.dropout_adj (sgp_preprocessing.py:177-179): one bernoulli draw
        # over E entries of value 1 - p on the host RNG keeps the surviving edges; the backward
        # operator is built from the same thinned list (:203-204)
        if _is_sparse_like(edge_index):
            raise NotImplementedError("dropout_rate > 0 needs an edge_index tensor")
        ei = torch.as_tensor(edge_index).cpu()
        keep = torch.bernoulli(torch.full((ei.shape[1],), 1. - dropout_rate, dtype=torch.float)).to(torch.bool)
        edge_index = ei[:, keep]
        edge_weight = None if edge_weight is None else torch.as_tensor(edge_weight).cpu()[keep]
    ops = spatial_operators(edge_index, edge_weight, num_nodes, undirected=undirected,
                            add_self_loops=add_self_loops,
                            remove_self_loops=remove_self_loops,
                            bidirectional=bidirectional)
    dev = x.device
    xg = x.float() if x.dtype != torch.float32 else x
    squeeze = xg.dim() == 2
    if squeeze:
        xg = xg[None]
    if not xg.is_cuda:
        hip.require_gpu()
        xg = xg.cuda()
    if one_hot_encoding:                              # :194-197
        ids = torch.eye(num_nodes, dtype=xg.dtype, device=xg.device)
        xg = torch.cat([xg, ids.unsqueeze(0).expand(xg.size(0), -1, -1)], dim=-1)
    B, N, F = xg.shape
    P = 1 + len(ops) * k
    out = torch.empty(B, N, P * F, dtype=torch.float32, device=xg.device)
    hip.copy_rows(xg if xg.stride(2) == 1 else xg.contiguous(), out[:, :, :F])
    propagate_into(out, F, ops, k)
    if dev != out.device:
        out = out.to(dev)
    res = [out[:, :, i * F:(i + 1) * F] for i in range(P)]
    if squeeze:
        res = [r[0] for r in res]
    return res


def _row_normalise(a, gcn_norm):
    import scipy.sparse as sp
    deg = np.asarray(a.sum(1)).ravel().astype(np.float32)
    with np.errstate(divide="ignore"):
        d = np.power(deg, -0.5 if gcn_norm else -1.0)
    d[np.isinf(d)] = 0
    out = sp.diags(d) @ a
    if gcn_norm:
        out = out @ sp.diags(d)
    return out.tocsr()


def _operator_from_scipy(m):
    m = m.tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return ShiftOperator(torch.from_numpy(m.indptr.astype(np.int64)),
                         torch.from_numpy(m.indices.astype(np.int64)),
                         torch.from_numpy(m.data.astype(np.float32)), m.shape[0])


def sgp_spatial_support(edge_index, edge_weight=None, num_nodes=None, k=2, undirected=False,
                        add_self_loops=False, remove_self_loops=False, bidirectional=False,
                        global_attr=False):
    """lib/sgp_preprocessing.py:108-160: explicit sparse supports for on-the-fly propagation
    (``sgp_preprocessing: True``).  One-off graph preparation, done on the host with scipy
    (SpSpGEMM); every support is a :class:`ShiftOperator` whose ``@`` runs on the GPU.  The
    reference's quirks are kept so that a decoder trained against them sees the same inputs:
    every support after the first is ``A_hat @ A_hat`` (:143-145), and the ``bidirectional``
    recursion is handed the un-transposed adjacency (:147-154), i.e. its supports are the
    row-normalised forward operator again.  ``global_attr`` appends a dense 1/N matrix."""
    import scipy.sparse as sp
    if _is_sparse_like(edge_index):
        row, col, val = edge_index.coo()
        n = edge_index.size(0)
        val = np.ones(row.numel(), np.float32) if val is None else val.float().numpy()
        adj = sp.csr_matrix((val, (row.numpy(), col.numpy())), shape=(n, n))
    elif sp.issparse(edge_index):
        adj = edge_index.tocsr().astype(np.float32)
        n = adj.shape[0]
    else:
        ei = torch.as_tensor(edge_index).long().cpu().numpy()
        n = int(num_nodes) if num_nodes is not None else int(ei.max()) + 1
        w = np.ones(ei.shape[1], np.float32) if edge_weight is None else \
            torch.as_tensor(edge_weight).float().cpu().numpy()
        adj = sp.csr_matrix((w, (ei[1], ei[0])), shape=(n, n))       # "transpose", :117-119
    adj.sum_duplicates()
    if undirected:
        adj = (adj + adj.T).tocsr()
    if add_self_loops:
        adj = adj.tolil(); adj.setdiag(1.0); adj = adj.tocsr()
    elif remove_self_loops:
        adj = adj.tolil(); adj.setdiag(0.0); adj = adj.tocsr(); adj.eliminate_zeros()
    adj_0 = _row_normalise(adj, gcn_norm=undirected)
    support = [_operator_from_scipy(adj_0)]
    if k > 1:
        sq = _operator_from_scipy(adj_0 @ adj_0)
        support += [sq for _ in range(k - 1)]
    if bidirectional:
        support += sgp_spatial_support(adj, k=k)
    if global_attr:
        support.append(torch.full((n, n), 1.0 / n))
    return support


def reservoir_preprocessing_(data, hidden_size: int,
                             preprocess_exogenous: Union[bool, List] = False,
                             num_layers=1, leaking_rate=0.9, spectral_radius=0.9,
                             density=0.9, activation='tanh', bias=True, cuda=False):
    """lib/sgp_preprocessing.py:40-64 (legacy helper; always computes on the GPU)."""
    reservoir = Reservoir(input_size=data.size(-1), hidden_size=hidden_size,
                          num_layers=num_layers, leaking_rate=leaking_rate,
                          spectral_radius=spectral_radius, density=density,
                          activation=activation, bias=bias)
    return reservoir(data[None])[0].to(data.device)


def preprocess_dataset(dataset, preprocess_exogenous, reservoir_kwargs, sgp_kwargs):
    """lib/sgp_preprocessing.py:15-37 (legacy one-call API, uncalled in the reference)."""
    if isinstance(preprocess_exogenous, bool):
        preprocess_exogenous = dataset.exogenous.keys() if preprocess_exogenous else []
    preprocess_exogenous = ensure_list(preprocess_exogenous)
    data, _ = dataset.get_tensors(['data'] + preprocess_exogenous, preprocess=True, cat_dim=-1)
    res = reservoir_preprocessing_(data, **reservoir_kwargs)
    res = sgp_spatial_embedding(res, num_nodes=data.size(1), edge_index=dataset.edge_index,
                                edge_weight=dataset.edge_weight, **sgp_kwargs)
    dataset.add_exogenous('processed_x', torch.cat(res, -1), add_to_input_map=False)
    dataset.set_input_map({'x': ['processed_x']})
