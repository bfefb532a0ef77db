"""Short, profiler-friendly run of the two hot kernels on the headline shapes
(N = 100k, 100-NN, D = 64).  Used under rocprofv3 (--kernel-trace --stats, or --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgp_amd  # noqa: E402
from sgp_amd import graph, hip, synthetic  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "spmm"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    N, D = 100000, 64
    if what in ("spmm", "both"):
        ei, ew, _ = synthetic.knn_graph(N, 100)
        op = graph.ShiftOperator.from_edges(ei, ew, N)
        x = torch.tanh(torch.randn(T, N, D, device="cuda"))
        y = torch.empty_like(x)
        force = os.environ.get("SGP_FORCE", "split")
        for _ in range(3):
            op.propagate(x, y, force=force, x_bound=1.0 if force == "split" else None)
        torch.cuda.synchronize()
    if what in ("res", "both"):
        torch.manual_seed(0)
        res = sgp_amd.Reservoir(64, 64)
        xin = torch.randn(T, N, 64, device="cuda")
        out = torch.empty(T, N, 64, device="cuda")
        for _ in range(3):
            res.encode_into(xin, out)
        torch.cuda.synchronize()
    if what == "res256":                       # C5's layer: F = 128, R = 256 (weights streamed through LDS)
        torch.manual_seed(0)
        res = sgp_amd.Reservoir(128, 256)
        xin = torch.randn(T, N, 128, device="cuda")
        out = torch.empty(T, N, 256, device="cuda")
        for _ in range(3):
            res.encode_into(xin, out)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
