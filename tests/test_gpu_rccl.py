"""The two halo-exchange forms over RCCL with DATA on the wire, on the one GPU the pool's leases have: a one-rank
``nccl`` group whose local block treats a set of its OWN rows as halo rows, so that ``all_to_all_single`` (packed form,
explicit split sizes) and ``all_gather`` (full-shard form) move real rows through RCCL and the hop reads them from the
receive buffer.  (Uneven split sizes across peers need more than one rank: covered over gloo by
tests/test_partition.py and by the multi-rank bench tests that share the GPU; RCCL has never seen two ranks here.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from sgp_amd import partition, synthetic, multigpu
from sgp_amd.graph import ShiftOperator
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = %r
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=multigpu.dist_timeout())
dev = torch.device("cuda", 0)
n, T, D, K = 3000, 24, 64, 2
ei, ew, _ = synthetic.knn_graph(n, 30, seed=7)
op = ShiftOperator.from_edges(ei, ew, n)
x = torch.tanh(torch.randn(T, n, D, device=dev))
ref = torch.empty(T, n, (1 + K) * D, device=dev); ref[..., :D] = x
for h in range(K):
    op.propagate(ref[..., h * D:(h + 1) * D], ref[..., (h + 1) * D:(h + 2) * D], force="csr")
rp, col, val = op.rowptr.long(), op.col.long(), op.val
for form in ("packed", "gather"):
    if form == "packed":
        # every third node doubles as a "halo" row: columns pointing at it read the RECEIVED copy
        halo = torch.arange(0, n, 3)
        is_h = torch.zeros(n, dtype=torch.bool); is_h[halo] = True
        pos = torch.cumsum(is_h, 0) - 1
        local = torch.where(is_h[col], n + pos[col], col)
        blk = partition.LocalBlock(ShiftOperator(rp, local, val, n, num_cols=n + halo.numel()), 0, n, halo,
                                   [int(halo.numel())], halo.int(), [int(halo.numel())])
    else:
        # full-shard form: odd columns read the gathered buffer (column n + i = row i of rank 0's shard)
        local = torch.where(col %% 2 == 1, n + col, col)
        blk = partition.LocalBlock(ShiftOperator(rp, local, val, n, num_cols=2 * n), 0, n, torch.arange(n),
                                   [n], torch.zeros(0, dtype=torch.int32), [0], gather_rows=n)
    sp = partition.PartitionedSpatial([blk], K, False, n, force_collectives=True, n_chunks=3)
    sp.norm_inf = [op.norm_inf()]
    out = torch.zeros(T, n, (1 + K) * D, device=dev); out[..., :D] = x
    sp.encode_into(out, D, x_bound=1.0)
    torch.cuda.synchronize()
    err = float((out - ref).abs().max())
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5), (form, err)
    print("rccl-ok", form, blk.n_halo, "%%.2e" %% err, blk.op.resolved_kernel())
dist.destroy_process_group()
'''


def test_both_exchange_forms_move_rows_through_rccl_on_one_rank():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, str(29700 + os.getpid() % 200))], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert r.stdout.count("rccl-ok") == 2, r.stdout
