import torch, sys
sys.path.insert(0, '/root/repo')
from sgp_amd import graph, hip, synthetic
from bench_probe import timeit
N, T, D = 100000, 256, 64
ei, ew, _ = synthetic.knn_graph(N, 100)
op = graph.ShiftOperator.from_edges(ei, ew, N)
x = torch.randn(T, N, D, device="cuda"); y = torch.empty_like(x)
ms = timeit(lambda: op.propagate(x, y, force="mfma"))
print("mfma ms", ms)
