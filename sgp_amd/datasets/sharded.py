"""Embeddings that do not live in one host tensor: time / node shards on disk (outputs larger than host RAM --
SURVEY.md 8b "for outputs > host RAM the build's harness writes shards"; ``run_largescale_sgp.py:208-212``
logs 629 GB for configuration C5) and node shards on devices ("GPU-resident, node-sharded
embedding ... trivially sharded").

* ``ShardedEmbedding``: index over shard files ``dict(t0, steps, rows | None, embedding[steps, rows, D])`` as
  ``SGPEncoder.encode_to_shards`` (one GPU: time shards of all nodes) and ``multigpu.encode_multi_gpu(...,
  shard_dir=)`` (N GPUs: time x node-block shards) write them; loads any time range back, in the original node
  order.
* ``ShardedIIDSampler``: the reference's ``IIDDataset.sample`` (``lib/datasets/iid_dataset.py:57-99``) on a
  NODE-SHARDED embedding.  The index sequence is drawn once, exactly as the reference draws it; every shard
  gathers the rows of the nodes it owns with ``sgp_gather_rows_f32`` on its own device and the pieces meet in
  the batch -- inside one process (shards on one or several devices) or, with ``group=``, across ranks that
  each hold one shard (one ``all_reduce`` of the [n, D] batch: the other ranks contribute exact zeros).
"""
import glob
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .. import hip
from .iid_dataset import IIDSampler, _Entry


class ShardedEmbedding:
    def __init__(self, paths: Sequence[str], n_steps: int, n_nodes: int, d_out: int):
        self.paths = list(paths)
        self.shape = (int(n_steps), int(n_nodes), int(d_out))
        self._index = None

    @classmethod
    def from_dir(cls, shard_dir):
        meta = torch.load(os.path.join(shard_dir, "index.pt"))
        paths = [os.path.join(shard_dir, os.path.basename(p)) for p in meta["paths"]]
        return cls(paths, *meta["shape"])

    @staticmethod
    def write_index(shard_dir, paths, shape):
        torch.save(dict(paths=[os.path.basename(p) for p in paths], shape=tuple(int(v) for v in shape)),
                   os.path.join(shard_dir, "index.pt"))

    def index(self):
        """[(t0, steps, path)] -- read from the file names' shard headers once."""
        if self._index is None:
            out = []
            for p in self.paths:
                s = torch.load(p, mmap=True)                 # (header only: the embedding stays on disk)
                out.append((int(s["t0"]), int(s["steps"]), p))
            self._index = sorted(out)
        return self._index

    def load_steps(self, t0, t1):
        """Host tensor [t1 - t0, N, D] in the original node order."""
        T, N, D = self.shape
        t0, t1 = max(0, int(t0)), min(T, int(t1))
        out = torch.empty(t1 - t0, N, D, dtype=torch.float32)
        for s0, steps, path in self.index():
            lo, hi = max(t0, s0), min(t1, s0 + steps)
            if lo >= hi:
                continue
            s = torch.load(path)
            piece = s["embedding"][lo - s0:hi - s0]
            rows = s.get("rows")
            if rows is None:
                out[lo - t0:hi - t0] = piece
            else:
                out[lo - t0:hi - t0].index_copy_(1, rows.long(), piece)
        return out

    def __len__(self):
        return self.shape[0]


class ShardedIIDSampler(IIDSampler):
    """``add_input`` / ``add_target`` take, for node-level tensors, a LIST of node shards
    ``[(node_ids, tensor[T, len(node_ids), f]), ...]`` (device-resident, any devices; with ``group=`` only the
    calling rank's shard) instead of one tensor; graph-level tensors (pattern ``t f``) stay whole.  ``sample``
    returns the same batch as ``IIDSampler`` on the unsharded tensor."""

    def __init__(self, n_steps, n_nodes, horizon, delay=0, horizon_lag=1, device=None, group=None):
        super().__init__(n_steps, n_nodes, horizon, delay, horizon_lag, device)
        self.group = group
        self._owner = {}                                    # key -> (shard of node [N], local row of node [N])

    def _register(self, store, key, shards, pattern, scaler, preprocess):
        if "n" not in pattern.split():
            raise ValueError("node shards need a pattern with a node axis")
        hip.require_gpu()
        owner = torch.full((self.n_nodes,), -1, dtype=torch.long)
        local = torch.zeros(self.n_nodes, dtype=torch.long)
        tensors = []
        for k, (ids, t) in enumerate(shards):
            ids = torch.as_tensor(ids, dtype=torch.long).cpu()
            if t.dim() != 3 or t.shape[0] != self.n_steps or t.shape[1] != ids.numel():
                raise ValueError("shard tensor must be [n_steps, len(node_ids), f]")
            if (owner[ids] >= 0).any():
                raise ValueError("a node belongs to two shards")
            owner[ids] = k
            local[ids] = torch.arange(ids.numel())
            t = t.to(torch.float32) if t.is_cuda else t.to(self.device, torch.float32)
            tensors.append(t if t.stride(-1) == 1 else t.contiguous())
        if self.group is None and (owner < 0).any():
            raise ValueError("the shards do not cover every node")
        e = _Entry(_ShardList(tensors), pattern, scaler, preprocess)
        store[key] = e
        self._owner[id(e)] = (owner, local)

    def add_input_shards(self, key, shards, pattern="t n f", scaler=None, preprocess=True):
        self._register(self.inputs, key, shards, pattern, scaler, preprocess)

    def add_target_shards(self, key, shards, pattern="t n f", scaler=None, preprocess=True):
        self._register(self.targets, key, shards, pattern, scaler, preprocess)

    def _rows(self, e, steps, nodes):
        if not isinstance(e.tensor, list):
            return super()._rows(e, steps, nodes)
        owner, local = self._owner[id(e)]
        nodes_cpu = nodes.long().cpu()
        own, loc = owner[nodes_cpu], local[nodes_cpu]
        f = e.tensor[0].shape[-1]
        out = torch.zeros(steps.numel(), f, dtype=torch.float32, device=self.device)
        for k, t in enumerate(e.tensor):
            sel = (own == k).nonzero().flatten()
            if sel.numel() == 0:
                continue
            st = steps[sel.to(steps.device)].to(t.device, torch.int32).contiguous()
            nd = loc[sel].to(t.device, torch.int32).contiguous()
            rows = hip.gather_rows(t, st, nd)               # sgp_gather_rows_f32 on the shard's own device
            out.index_copy_(0, sel.to(self.device), rows.to(self.device))
        if self.group is not None:
            import torch.distributed as dist
            if dist.get_backend(self.group) == "gloo":
                c = out.cpu()
                dist.all_reduce(c, group=self.group)
                out.copy_(c)
            else:
                dist.all_reduce(out, group=self.group)
        return out


class _ShardList(list):
    """A list of shard tensors that answers ``dim()`` like the [T, N, f] tensor it stands for."""

    def dim(self):
        return 3
