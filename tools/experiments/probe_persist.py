"""Persistent / XCD-lockstep launch modes of spmm_pipe on the target graph: time per hop and
bit-equality with the default launch."""
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


def main():
    N, D = int(os.environ.get("SGP_PROBE_N", 100000)), 64
    T = int(os.environ.get("SGP_PROBE_T", 512))
    ei, ew, _ = synthetic.knn_graph(N, 100)
    op = graph.ShiftOperator.from_edges(ei, ew, N)
    x = torch.randn(T, N, D, device="cuda")
    y = torch.empty_like(x)
    y0 = torch.empty_like(x)
    lib = hip.load()
    bytes_hop = 2 * N * T * D * 4 + op.nnz() * 8 + (N + 1) * 4
    lib.sgp_spmm_pipe_tune(0, 128)
    op.propagate(x, y0, force="pipe")
    modes = os.environ.get("SGP_PROBE_MODES", "0:128,1:32,1:64,1:128,1:256,2:32,2:64,2:128,2:256,2:16")
    for m in modes.split(","):
        persist, unit = (int(v) for v in m.split(":"))
        lib.sgp_spmm_pipe_tune(persist, unit)
        y.zero_()
        ms = timeit(lambda: op.propagate(x, y, force="pipe"))
        same = bool(torch.equal(y, y0))
        print(f"persist={persist} unit={unit:4d}: {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}  equal={same}", flush=True)
    lib.sgp_spmm_pipe_tune(0, 128)


if __name__ == "__main__":
    main()
