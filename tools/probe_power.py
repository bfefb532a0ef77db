"""Is the hop power-bound?  Same kernel, same plan, same instruction stream on three inputs: N(0,1) values,
all zeros (no toggling in the LDS / matrix datapaths) and a constant.  A DVFS-limited kernel runs faster on
the quiet inputs (MI355X_MICROARCH.md, DVFS give-back); a pipe- or latency-bound one does not care.
python tools/probe_power.py [T] [kernels...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgp_amd import graph, hip, synthetic

N = 100000
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ei, ew, _ = synthetic.knn_graph(N, 100, seed=1)
op = graph.ShiftOperator.from_edges(ei, ew, N)
dev = torch.device("cuda")
y = torch.empty(T, N, 64, device=dev)
bytes_hop = 2 * N * T * 64 * 4 + op.nnz() * 8 + (N + 1) * 4
for force in sys.argv[2:] or ("res", "mix"):
    for name, make in (("randn", lambda: torch.randn(T, N, 64, device=dev)),
                       ("zeros", lambda: torch.zeros(T, N, 64, device=dev)),
                       ("const", lambda: torch.full((T, N, 64), 0.5, device=dev)),
                       ("randn", lambda: torch.randn(T, N, 64, device=dev))):
        x = make()
        op.propagate(x, y, force=force)
        torch.cuda.synchronize()
        times = []
        for _ in range(6):
            e0, e1 = hip.Event(), hip.Event()
            e0.record(); op.propagate(x, y, force=force); e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_ms(e1))
        best = min(times)
        print(f"{force} {name}: best {best:.3f} ms  median {sorted(times)[3]:.3f}  frac {bytes_hop / best / 1e6 / 8000:.4f}", flush=True)
        del x
