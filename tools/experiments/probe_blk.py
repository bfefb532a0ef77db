"""spmm_blk (row-block kernel) against spmm_res on the target graph: time per hop, agreement with
the generic CSR kernel."""
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
    bytes_hop = 2 * N * T * D * 4 + op.nnz() * 8 + (N + 1) * 4
    kernels = os.environ.get("SGP_PROBE_KERNELS", "res,blk").split(",")
    if "blk" in kernels:
        bp = op.block_plan(D, x.device)
        print("blk tiles", bp.n_tiles, "max_union", bp.max_union, "max_steps", bp.max_steps, "fill", round(bp.fill, 4), flush=True)
    else:
        tp = op.tile_plan(D, x.device)
        print("tiles", tp.n_tiles, "fill", round(tp.pipe["fill"], 4), flush=True)
        if "tri" in kernels:
            tt = op.tri_plan(D, x.device)
            print("tri tiles", tt.n_tiles, "fill", round(tt.pipe["fill"], 4), "max_union", tt.pipe["max_union"], flush=True)
    if os.environ.get("SGP_PROBE_CHECK", "1") == "1":
        op.propagate(x[:4], y0[:4], force="csr")
        for kname in kernels:
            op.propagate(x[:4], y[:4], force=kname)
            err = (y[:4] - y0[:4]).abs().max().item()
            print(f"{kname} vs csr: max|diff| = {err:.3g}  allclose(1e-5) = {torch.allclose(y[:4], y0[:4], rtol=1e-5, atol=1e-5)}", flush=True)
    for force in kernels:
        for cfg in ([0] if force != "blk" else [int(c) for c in os.environ.get("SGP_PROBE_CFGS", "0,1,2").split(",")]):
            hip.load().sgp_spmm_blk_tune(cfg)
            ms = timeit(lambda: op.propagate(x, y, force=force))
            print(f"{force} cfg={cfg}: {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}", flush=True)


if __name__ == "__main__":
    main()
