"""spmm_res (register-resident stream) against spmm_pipe on the target graph: time per hop and
bit-equality."""
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
    plan = op.tile_plan(D, x.device)
    print("tiles", plan.n_tiles, "max_union", plan.pipe["max_union"], "max_range_steps", plan.pipe["max_range_steps"],
          "fill", round(plan.pipe["fill"], 4), flush=True)
    ms = timeit(lambda: op.propagate(x, y0, force="pipe"))
    print(f"pipe: {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}", flush=True)
    for cfg in (int(c) for c in os.environ.get("SGP_PROBE_CFGS", "0,1").split(",")):
        lib.sgp_spmm_res_tune(cfg)
        y.zero_()
        ms = timeit(lambda: op.propagate(x, y, force="res"))
        same = bool(torch.equal(y, y0))
        err = (y - y0).abs().max().item()
        print(f"res cfg={cfg}: {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}  equal={same} max|diff|={err:.3g}", flush=True)


if __name__ == "__main__":
    main()
