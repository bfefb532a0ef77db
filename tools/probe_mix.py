"""Mixed dense / sparse hop kernel vs spmm_res on the target graph: parity against the CSR kernel and
time per launch.  python tools/probe_mix.py [N] [T] [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgp_amd import graph, hip, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ei, ew, _ = synthetic.knn_graph(N, 100, seed=1)
op = graph.ShiftOperator.from_edges(ei, ew, N)
dev = torch.device("cuda")
t0 = time.time(); op.tile_plan(64, dev, tall=False); t1 = time.time()
mp = op.mix_plan(64, dev, strict=False); t2 = time.time()
print(f"plans: base {t1 - t0:.1f}s mix {t2 - t1:.1f}s  dense_share {mp.dense_share:.3f} max_dense {mp.max_dense} "
      f"max_range {mp.max_range_steps} max_quads {mp.max_tile_quads} cost {mp.mean_phase_cost:.1f} thr {mp.thr}", flush=True)
x = torch.randn(T, N, 64, device=dev)
y = torch.empty_like(x)
ref = torch.empty(min(T, 8), N, 64, device=dev)
op.propagate(x[:ref.shape[0]], ref, force="csr")
bytes_hop = 2 * N * T * 64 * 4 + op.nnz() * 8 + (N + 1) * 4
for force in sys.argv[4:] or ("res", "mix"):
    y.fill_(float("nan"))
    op.propagate(x, y, force=force)
    torch.cuda.synchronize()
    err = (y[:ref.shape[0]] - ref).abs().max().item()
    e0, e1 = hip.Event(), hip.Event()
    best = 1e9
    for _ in range(reps):
        e0.record(); op.propagate(x, y, force=force); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_ms(e1))
    print(f"{force}: {best:.3f} ms  {bytes_hop / best / 1e6:.0f} GB/s  frac {bytes_hop / best / 1e6 / 8000:.4f}  max|err| vs csr {err:.2e} "
          f"nan {int(torch.isnan(y).sum())}", flush=True)
