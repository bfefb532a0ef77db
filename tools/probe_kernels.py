"""Scratch GPU probe (first contact): timings of the kernels on the headline shapes."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgp_amd
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
    only_spmm = len(sys.argv) > 1 and sys.argv[1] == "spmm"
    N, T, D, K = int(os.environ.get("SGP_PROBE_N", 100000)), 256, 64, 4
    ei, ew, _ = synthetic.knn_graph(N, int(os.environ.get("SGP_PROBE_K", 100)))
    op = graph.ShiftOperator.from_edges(ei, ew, N)
    x = torch.randn(T, N, D, device="cuda")
    y = torch.empty_like(x)
    bytes_hop = 2 * N * T * D * 4 + op.nnz() * 8 + (N + 1) * 4
    plan = op.tile_plan(D, torch.device("cuda"))
    print("group fill", plan.group_fill, "pipe fill", plan.pipe["fill"], flush=True)
    N_T = int(os.environ.get("SGP_PROBE_T", T))
    if N_T != T:
        T = N_T
        x = torch.randn(T, N, D, device="cuda"); y = torch.empty_like(x)
        bytes_hop = 2 * N * T * D * 4 + op.nnz() * 8 + (N + 1) * 4
    for force in os.environ.get("SGP_PROBE", "split,mix,res,tiled,csr").split(","):
        ms = timeit(lambda: op.propagate(x, y, force=force))
        print(f"spmm {force}: {ms:.2f} ms  {bytes_hop / ms / 1e6:.1f} GB/s  frac {bytes_hop / ms / 1e6 / 8000:.3f}", flush=True)
    if only_spmm:
        return
    eir, ewr = synthetic.random_graph(N, 100)
    opr = graph.ShiftOperator.from_edges(eir, ewr, N)
    ms = timeit(lambda: opr.propagate(x, y))
    print(f"spmm random-graph csr: {ms:.2f} ms frac {bytes_hop / ms / 1e6 / 8000:.3f}", flush=True)
    # reservoir
    for (F, R) in [(64, 64), (3, 64), (128, 256)]:
        n = N if R < 256 else 20000
        res = sgp_amd.Reservoir(F, R)
        xin = torch.randn(T, n, F, device="cuda")
        out = torch.empty(T, n, R, device="cuda")
        ms = timeit(lambda: res.encode_into(xin, out), n=2)
        fl = n * T * 2 * R * (F + R)
        print(f"reservoir F={F} R={R} N={n}: {ms:.2f} ms {fl / ms / 1e9:.1f} TF/s", flush=True)
    for (n, F, R, t) in [(207, 3, 64, 2016), (325, 3, 128, 2016)]:
        res = sgp_amd.Reservoir(F, R)
        xin = torch.randn(t, n, F, device="cuda")
        out = torch.empty(t, n, R, device="cuda")
        ms = timeit(lambda: res.encode_into(xin, out), n=2)
        print(f"reservoir small N={n} R={R} T={t}: {ms:.2f} ms  {ms / t * 1e3:.2f} us/step", flush=True)


if __name__ == "__main__":
    main()
