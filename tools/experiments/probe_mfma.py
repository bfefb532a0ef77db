import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sgp_amd import graph, hip, synthetic
from probe_kernels import timeit
N, T, D = 100000, 256, 64
ei, ew, _ = synthetic.knn_graph(N, 100)
op = graph.ShiftOperator.from_edges(ei, ew, N)
x = torch.randn(T, N, D, device="cuda"); y = torch.empty_like(x)
ms = timeit(lambda: op.propagate(x, y, force="mfma"))
print("mfma ms", ms)
