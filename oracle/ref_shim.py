"""Container-only import shim for the *unmodified* reference hot-path files.

TEST INFRASTRUCTURE -- never imported by the product (``sgp_amd``), never run
on the GPU box (``/root/reference`` does not exist there).  It exists so that
``oracle/make_golden.py`` can execute the reference's own Python
(``lib/nn/reservoir/reservoir.py``, ``lib/sgp_preprocessing.py``,
``lib/nn/encoders/*.py``, ``lib/utils.py``) and record input/output vectors
under ``tests/golden/``.

The reference depends on third-party packages that are absent from this image
(``torch_sparse``, ``torch_geometric``, ``torch_scatter``, ``test_tube``,
``pytorch_lightning``).  Only their *documented semantics* are restated here,
in pure torch, with no code taken from them:

* ``torch_sparse.SparseTensor`` (pairs with pyg=2.0 / pytorch=1.9 in the
  reference's ``conda_env.yml:9-11``; the wheel of that era is 0.6.12):
  COO triplets sorted by (row, col), duplicates kept (no coalescing);
  ``value=None`` means implicit ones; ``set_diag()`` replaces the diagonal
  with ones; ``remove_diag()`` drops it; ``sum(dim)`` reduces values;
  ``dense(N,1) * sp`` scales rows, ``sp * dense(1,N)`` scales columns;
  ``sp @ dense[..., N, F]`` is a batched sum-SpMM.
* ``torch_geometric.utils``: ``dropout_adj`` (identity for p == 0),
  ``to_undirected`` (both directions concatenated, duplicates summed),
  ``add_self_loops`` (append unit loops for ``max_index + 1`` nodes).

Parity at that third-party boundary is therefore "unpinned" by any upstream
binary; the dense-matrix cross-check in ``tests/test_oracle.py`` is the
tie-breaker ("Oracle").
"""
import argparse
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("SGP_REFERENCE_ROOT", "/root/reference")


# --------------------------------------------------------------------------
# torch_sparse
# --------------------------------------------------------------------------
class SparseTensor:
    def __init__(self, row=None, rowptr=None, col=None, value=None,
                 sparse_sizes=None, is_sorted=False):
        assert row is not None and col is not None
        row = row.to(torch.long)
        col = col.to(torch.long)
        if sparse_sizes is None or sparse_sizes[0] is None:
            m = int(row.max()) + 1 if row.numel() else 0
            n = int(col.max()) + 1 if col.numel() else 0
            sparse_sizes = (m, n)
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        if not is_sorted and row.numel():
            key = row * self._sizes[1] + col
            perm = torch.argsort(key, stable=True)
            row, col = row[perm], col[perm]
            if value is not None:
                value = value[perm]
        self._row, self._col, self._value = row, col, value

    # -- accessors ---------------------------------------------------------
    def coo(self):
        return self._row, self._col, self._value

    def csr(self):
        counts = torch.bincount(self._row, minlength=self._sizes[0])
        rowptr = torch.zeros(self._sizes[0] + 1, dtype=torch.long)
        rowptr[1:] = torch.cumsum(counts, 0)
        return rowptr, self._col, self._value

    def size(self, dim):
        return self._sizes[dim]

    def sizes(self):
        return list(self._sizes)

    def sparse_sizes(self):
        return self._sizes

    def has_value(self):
        return self._value is not None

    def nnz(self):
        return self._row.numel()

    def _vals(self, dtype=torch.float32):
        if self._value is None:
            return torch.ones(self._row.numel(), dtype=dtype)
        return self._value

    # -- structure ops -----------------------------------------------------
    def t(self):
        return SparseTensor(row=self._col, col=self._row, value=self._value,
                            sparse_sizes=(self._sizes[1], self._sizes[0]))

    def remove_diag(self):
        keep = self._row != self._col
        v = None if self._value is None else self._value[keep]
        return SparseTensor(row=self._row[keep], col=self._col[keep], value=v,
                            sparse_sizes=self._sizes, is_sorted=True)

    def set_diag(self, values=None):
        base = self.remove_diag()
        n = min(self._sizes)
        idx = torch.arange(n, dtype=torch.long)
        if self._value is None:
            v = None
        else:
            dv = torch.ones(n, dtype=self._value.dtype) if values is None \
                else values
            v = torch.cat([base._value, dv])
        return SparseTensor(row=torch.cat([base._row, idx]),
                            col=torch.cat([base._col, idx]), value=v,
                            sparse_sizes=self._sizes)

    def sum(self, dim=None):
        v = self._value
        if dim is None:
            return self._vals().sum()
        index = self._row if dim in (1, -1) else self._col
        size = self._sizes[0] if dim in (1, -1) else self._sizes[1]
        if v is None:
            return torch.bincount(index, minlength=size)
        out = torch.zeros(size, dtype=v.dtype)
        out.index_add_(0, index, v)
        return out

    # -- arithmetic --------------------------------------------------------
    def _scale(self, other):
        assert torch.is_tensor(other) and other.dim() == 2
        if other.size(0) == self._sizes[0] and other.size(1) == 1:
            s = other[:, 0][self._row]
        elif other.size(0) == 1 and other.size(1) == self._sizes[1]:
            s = other[0][self._col]
        else:
            raise ValueError("unsupported broadcast in SparseTensor shim")
        v = s if self._value is None else s * self._value
        return SparseTensor(row=self._row, col=self._col, value=v,
                            sparse_sizes=self._sizes, is_sorted=True)

    def __mul__(self, other):
        return self._scale(other)

    __rmul__ = __mul__

    def mul(self, other):
        return self._scale(other)

    def __add__(self, other):
        assert isinstance(other, SparseTensor)
        a = self.to_dense_matrix() + other.to_dense_matrix()
        # torch_sparse's add keeps the union pattern
        pat = (self.to_dense_pattern() + other.to_dense_pattern()) > 0
        r, c = pat.nonzero(as_tuple=True)
        return SparseTensor(row=r, col=c, value=a[r, c],
                            sparse_sizes=self._sizes)

    def to_dense_matrix(self, dtype=torch.float32):
        a = torch.zeros(self._sizes, dtype=dtype)
        a.index_put_((self._row, self._col), self._vals(dtype).to(dtype),
                     accumulate=True)
        return a

    def to_dense_pattern(self):
        a = torch.zeros(self._sizes, dtype=torch.float32)
        a[self._row, self._col] = 1.0
        return a

    def to_dense(self):
        return self.to_dense_matrix()

    def index_select(self, dim, idx):
        """torch_sparse ``SparseTensor.index_select``: keep the listed rows (dim 0) / columns
        (dim 1) in the order given, repeats allowed."""
        idx = torch.as_tensor(idx, dtype=torch.long)
        m, n = self._sizes
        a, b = (self._row, self._col) if dim == 0 else (self._col, self._row)
        size = m if dim == 0 else n
        order = torch.argsort(a, stable=True)
        counts = torch.bincount(a, minlength=size)
        start = torch.cumsum(counts, 0) - counts
        take = torch.cat([order[start[i]:start[i] + counts[i]] for i in idx.tolist()]) \
            if idx.numel() else torch.zeros(0, dtype=torch.long)
        new_a = torch.repeat_interleave(torch.arange(idx.numel()), counts[idx])
        new_b = b[take]
        val = None if self._value is None else self._value[take]
        if dim == 0:
            return SparseTensor(row=new_a, col=new_b, value=val, sparse_sizes=(idx.numel(), n))
        return SparseTensor(row=new_b, col=new_a, value=val, sparse_sizes=(m, idx.numel()))

    def __matmul__(self, other):
        return matmul(self, other)


def matmul(src, other, reduce="sum"):
    assert reduce in ("sum", "add")
    if isinstance(other, SparseTensor):
        # SpSpGEMM: pattern = structural product, values summed
        a = src.to_dense_matrix() @ other.to_dense_matrix()
        pat = (src.to_dense_pattern() @ other.to_dense_pattern()) > 0
        r, c = pat.nonzero(as_tuple=True)
        return SparseTensor(row=r, col=c, value=a[r, c],
                            sparse_sizes=(src.size(0), other.size(1)))
    row, col, val = src.coo()
    val = src._vals(other.dtype)
    lead = other.shape[:-2]
    n, f = other.shape[-2:]
    x = other.reshape(-1, n, f)
    out = torch.zeros(x.size(0), src.size(0), f, dtype=other.dtype)
    # one row at a time, edges in (row, col) order: the accumulation order
    # of torch_sparse's CPU spmm_sum
    out.index_add_(1, row, x[:, col, :] * val.view(1, -1, 1))
    return out.reshape(*lead, src.size(0), f)


# --------------------------------------------------------------------------
# torch_geometric
# --------------------------------------------------------------------------
def _maybe_num_nodes(edge_index, num_nodes=None):
    if num_nodes is not None:
        return num_nodes
    return int(edge_index.max()) + 1 if edge_index.numel() else 0


def dropout_adj(edge_index, edge_attr=None, p=0.5, force_undirected=False,
                num_nodes=None, training=True):
    if p < 0. or p > 1.:
        raise ValueError(f"Dropout probability has to be between 0 and 1 "
                         f"(got {p}")
    if not training or p == 0.0:
        return edge_index, edge_attr
    raise NotImplementedError("shim: dropout_adj with p > 0 consumes RNG; "
                              "no reference caller enables it")


def to_undirected(edge_index, edge_attr=None, num_nodes=None, reduce="add"):
    if isinstance(edge_attr, int):
        num_nodes, edge_attr = edge_attr, None
    n = _maybe_num_nodes(edge_index, num_nodes)
    row, col = edge_index
    row, col = torch.cat([row, col]), torch.cat([col, row])
    key = row * n + col
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    out_index = torch.stack([uniq // n, uniq % n])
    if edge_attr is None:
        return out_index, None
    attr = torch.cat([edge_attr, edge_attr])
    out = torch.zeros(uniq.numel(), dtype=attr.dtype)
    out.index_add_(0, inv, attr)
    return out_index, out


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    n = _maybe_num_nodes(edge_index, num_nodes)
    loop = torch.arange(n, dtype=torch.long)
    loop = loop.unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        fv = 1. if fill_value is None else fill_value
        la = torch.full((n,), fv, dtype=edge_attr.dtype)
        edge_attr = torch.cat([edge_attr, la])
    return torch.cat([edge_index, loop], dim=1), edge_attr


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", **kwargs):
        super().__init__()
        self.aggr = aggr

    def propagate(self, edge_index, size=None, **kwargs):
        if isinstance(edge_index, SparseTensor):
            return self.message_and_aggregate(edge_index, kwargs["x"])
        raise NotImplementedError("shim: only the SparseTensor path is used")


# --------------------------------------------------------------------------
# test_tube
# --------------------------------------------------------------------------
class HyperOptArgumentParser(argparse.ArgumentParser):
    def __init__(self, *args, strategy="grid_search", **kwargs):
        super().__init__(*args, **kwargs)

    def opt_list(self, *args, options=None, tunable=False, **kwargs):
        self.add_argument(*args, **kwargs)

    def opt_range(self, *args, low=None, high=None, nb_samples=None,
                  log_base=None, tunable=False, **kwargs):
        self.add_argument(*args, **kwargs)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _bare_package(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Make ``import lib...`` resolve to the unmodified reference files."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}; "
                           "the shim only works in the build container")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    ts = _module("torch_sparse", SparseTensor=SparseTensor, matmul=matmul)
    ts.__version__ = "shim"
    tg = _module("torch_geometric")
    tg.__path__ = []
    _module("torch_geometric.typing", Adj=object, OptTensor=object,
            OptPairTensor=object, Size=object)
    _module("torch_geometric.utils", dropout_adj=dropout_adj,
            to_undirected=to_undirected, add_self_loops=add_self_loops)
    _module("torch_geometric.utils.num_nodes",
            maybe_num_nodes=_maybe_num_nodes)
    _module("torch_geometric.nn", MessagePassing=MessagePassing)
    _module("test_tube", HyperOptArgumentParser=HyperOptArgumentParser)

    import tsl  # real: global_scope + lazy loaders only

    r = os.path.join(REFERENCE_ROOT, "tsl")
    _module("tsl.data", SpatioTemporalDataset=object)
    _bare_package("tsl.utils", os.path.join(r, "utils"))
    _bare_package("tsl.ops", os.path.join(r, "ops"))
    nn_pkg = _bare_package("tsl.nn", os.path.join(r, "nn"))
    _bare_package("tsl.nn.blocks", os.path.join(r, "nn", "blocks"))
    _bare_package("tsl.nn.blocks.encoders",
                  os.path.join(r, "nn", "blocks", "encoders"))

    # tsl/nn/functional.py needs torch_scatter; the only symbol the hot path
    # pulls from it (via tsl/nn/ops/ops.py -> tsl/nn/utils/utils.py) is
    # expand_then_cat, which the hot path never calls.
    def expand_then_cat(tensors, dim=-1):
        raise NotImplementedError("shim: expand_then_cat is off the hot path")

    _module("tsl.nn.functional", expand_then_cat=expand_then_cat)
    import tsl.nn.utils as nn_utils  # real tsl/nn/utils/utils.py
    nn_pkg.utils = nn_utils
    tsl.nn = nn_pkg
    _installed = True


def load_reference():
    """Return a namespace with the reference's hot-path symbols."""
    install()
    from lib.nn.reservoir import Reservoir, ReservoirLayer, GraphESN, GESNLayer
    from lib.sgp_preprocessing import (sgp_spatial_embedding, preprocess_adj,
                                       sgp_spatial_support)
    from lib.nn.encoders import (SGPEncoder, SGPTemporalEncoder,
                                 SGPSpatialEncoder, GESNEncoder)
    from lib.utils import encode_dataset, self_normalizing_activation
    return types.SimpleNamespace(**locals())
