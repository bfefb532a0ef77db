"""On-the-fly spatial supports on the GPU -- the compute of "next" row f3.

With ``sgp_preprocessing: True`` the reference does not precompute the embedding: its loaders
apply the explicit supports of ``sgp_spatial_support`` to every sample,

* ``SGPLoader.collate`` (``lib/dataloader/sgp_dataloader.py:57-61``):
  ``sample[key] = torch.cat([x] + [adj @ x for adj in support], dim=-1)``
* ``IIDDataset._populate_input_frame`` (``lib/datasets/iid_dataset.py:111-114``) for a node subset:
  ``torch.cat([tens.index_select(1, node_index)] +
  [adj.index_select(0, node_index) @ tens for adj in self.sgp_support], dim=-1)``

``apply_supports`` is that expression with the products on the MI355X (``sgp_spmm_csr_f32`` on
the supports' CSR, rectangular for a node subset) and every block written straight into its slot
of the result (no ``torch.cat``).  The supports come from ``sgp_amd.sgp_spatial_support`` (the
reference's quirks included); the dense support of ``global_attr`` (1/N everywhere) is the scaled
column sum of x broadcast to every row (``sgp_node_sums`` + ``sgp_bcast_rows``).
The DataLoader / Batch plumbing around these lines is tsl's and is not rebuilt here.
"""
import torch

from .. import hip
from ..graph import ShiftOperator


def _rect_dense(a, x3, dst):
    """Rows of a dense support selected by ``node_index``: CSR of the [rows, N] block."""
    r, c = a.nonzero(as_tuple=True)
    counts = torch.bincount(r, minlength=a.shape[0])
    rowptr = torch.zeros(a.shape[0] + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(counts, 0)
    ShiftOperator(rowptr, c, a[r, c], a.shape[0], num_cols=a.shape[1]).propagate_rect(x3, dst)


_CONSTANT_CACHE = {}          # id(adj) -> (weakref to adj, adj._version, value); a handful of entries at most


def _constant_value(adj):
    """The single value of a dense support whose entries are all equal (the reference's global_attr
    support: 1 / N everywhere), else None.  Looked at once per support object AND version: the N x N matrix
    is not copied to the host and compared on every batch, a support modified in place is looked at again,
    and the cache holds the matrices weakly (it never keeps an N x N tensor alive)."""
    import weakref
    key = id(adj)
    version = getattr(adj, "_version", None)
    hit = _CONSTANT_CACHE.get(key)
    if hit is not None and hit[0]() is adj and hit[1] == version:
        return hit[2]
    a = torch.as_tensor(adj)
    value = None
    if a.numel():
        c = a.flatten()[0]
        if bool((a == c).all()):
            value = float(c)
    for k in [k for k, v in _CONSTANT_CACHE.items() if v[0]() is None]:      # entries whose tensor is gone
        del _CONSTANT_CACHE[k]
    if len(_CONSTANT_CACHE) >= 8:
        _CONSTANT_CACHE.clear()
    try:
        _CONSTANT_CACHE[key] = (weakref.ref(adj), version, value)
    except TypeError:                                   # (objects that cannot be weakly referenced: not cached)
        pass
    return value


def apply_supports(x, support, node_index=None):
    """x[..., N, F] -> [..., N (or len(node_index)), (1 + len(support)) * F]."""
    hip.require_gpu()
    dev_in = x.device
    xg = x.float()
    if not xg.is_cuda:
        xg = xg.cuda()
    lead = xg.shape[:-2]
    n, f = xg.shape[-2:]
    x3 = xg.reshape(-1, n, f)
    if x3.stride(2) != 1:
        x3 = x3.contiguous()
    idx = None if node_index is None else torch.as_tensor(node_index, dtype=torch.long)
    rows = n if idx is None else idx.numel()
    out = torch.empty(x3.shape[0], rows, (1 + len(support)) * f, dtype=torch.float32, device=xg.device)
    if idx is None:
        hip.copy_rows(x3, out[:, :, :f])
    else:
        hip.gather_nodes(x3, idx.to(xg.device, torch.int32), out[:, :, :f])
    for s, adj in enumerate(support, start=1):
        dst = out[:, :, s * f:(s + 1) * f]
        if isinstance(adj, ShiftOperator):
            op = adj if idx is None else adj.index_select(0, idx)
            op.propagate_rect(x3, dst)
        else:
            # dense support.  The reference has exactly one (global_attr: 1/N everywhere,
            # sgp_preprocessing.py:157-158): a constant matrix times x is the scaled column sum of
            # x in every row -- sgp_node_sums / sgp_bcast_rows, no N x N product.  Any other dense
            # matrix goes through the CSR kernel like the sparse supports.
            const = _constant_value(adj)          # decided once per support object, not per batch
            if const is not None:
                hip.bcast_rows(hip.node_sums(x3), const, dst)
            else:
                a = torch.as_tensor(adj, dtype=torch.float32).cpu()
                _rect_dense(a if idx is None else a[idx.cpu()], x3, dst)
    out = out.reshape(*lead, rows, out.shape[-1])
    return out if dev_in == out.device else out.to(dev_in)
