"""SpMM hop on the traffic-graph shapes (C1 / C2): every kernel that accepts the plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgp_amd import graph, hip, synthetic


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    return a.elapsed_ms(b) / n


for (N, E, D, T) in [(325, 2369, 128, 16384), (207, 1515, 64, 34272)]:
    ei, ew = synthetic.sparse_traffic_graph(N, E, seed=1)
    op = graph.ShiftOperator.from_edges(ei, ew, N, gcn_norm=True) if False else graph.ShiftOperator.from_edges(ei, ew, N)
    x = torch.randn(T, N, D, device="cuda"); y = torch.empty_like(x); y0 = torch.empty_like(x)
    bytes_hop = 2 * N * T * D * 4 + op.nnz() * 8 + (N + 1) * 4
    plan = op.tile_plan(D, x.device)
    print(f"N={N} E={E} D={D} T={T}: tiles {plan.n_tiles} fill {plan.pipe['fill'] if plan.pipe else None}", flush=True)
    op.propagate(x, y0, force="csr")
    for force in ("tiled", "res", "mix", "csr"):
        try:
            ms = timeit(lambda: op.propagate(x, y, force=force))
            err = float((y - y0).abs().max())
            print(f"  {force:6s}: {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}  max|diff vs csr| {err:.2g}", flush=True)
        except Exception as e:
            print(f"  {force:6s}: {type(e).__name__}: {str(e)[:100]}", flush=True)
