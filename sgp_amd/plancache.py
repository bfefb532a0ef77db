"""On-disk cache of the host-side hop plans (split / tile / mix), keyed by a hash of the operator's CSR arrays, the
kernel limits and a format version: the second ``encode_dataset`` of an experiment sweep (reference: one call per run,
lib/utils.py:27-32) pays no planning at all.  Off unless a directory is named -- ``SGP_AMD_CACHE=/path`` or
``sgp_amd.plancache.set_dir(path)``; files are ``torch.save`` pickles of the plan objects (CPU tensors), so the
directory must be one the user trusts, like any checkpoint directory."""
import hashlib
import os
import tempfile

import torch

VERSION = 6          # bump when a plan format or a planner's output changes
_dir = None
stats = {"hits": 0, "misses": 0, "stores": 0}


def set_dir(path):
    """Directory of the cache (None: back to the SGP_AMD_CACHE environment variable / off)."""
    global _dir
    _dir = path


def directory():
    return _dir or os.environ.get("SGP_AMD_CACHE") or None


def operator_hash(op):
    h = getattr(op, "_csr_hash", None)
    if h is None:
        m = hashlib.blake2b(digest_size=16)
        m.update(f"{op.num_nodes}:{op.num_cols}:{op.nnz()}".encode())
        for t in (op.rowptr, op.col, op.val):
            m.update(t.contiguous().numpy().tobytes())
        h = op._csr_hash = m.hexdigest()
    return h


def _path(op, kind, params):
    d = directory()
    if d is None:
        return None
    tag = hashlib.blake2b(repr((VERSION, kind, params)).encode(), digest_size=8).hexdigest()
    return os.path.join(d, f"{kind}-{operator_hash(op)}-{tag}.pt")


_MISS = object()


def fetch(op, kind, params, build):
    """The plan ``build()`` returns (CPU tensors inside, or None), from the cache when a file for (operator, kind,
    params) exists; a fresh build is stored.  A file that does not load (truncated, another version of the package)
    counts as a miss and is replaced."""
    path = _path(op, kind, params)
    if path is None:
        return build()
    if os.path.exists(path):
        try:
            got = torch.load(path, map_location="cpu", weights_only=False)
            if isinstance(got, dict) and got.get("version") == VERSION and got.get("params") == repr(params):
                stats["hits"] += 1
                return got["plan"]
        except Exception:
            pass
    stats["misses"] += 1
    plan = build()
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=os.path.dirname(path), suffix=".tmp")
        os.close(fd)
        torch.save({"version": VERSION, "params": repr(params), "plan": plan}, tmp)
        os.replace(tmp, path)                              # atomic: concurrent ranks never see half a file
        stats["stores"] += 1
    except OSError:
        pass                                               # (a read-only or full cache directory: plan anyway)
    return plan
