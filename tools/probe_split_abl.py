"""Ablations of the split-fp16 hop on the target graph (needs an ablation build: tools/build_variant.sh abl -DSGP_ABLATION,
SGP_AMD_LIB=tools/variants/abl/libsgp_amd.so) (SGP_TUNE=split_abl=.. is read once per process: one process per mode)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from sgp_amd import synthetic, hip
    from sgp_amd.graph import ShiftOperator
    n, T, D = 100000, int(os.environ.get("T", "128")), 64
    dev = torch.device("cuda:0")
    ei, ew, _ = synthetic.knn_graph(n, 100, seed=1)
    op = ShiftOperator.from_edges(ei, ew, n)
    plan = op.split_plan(dev)
    out = torch.empty(T, n, 5 * D, device=dev)
    out[..., :D] = torch.tanh(torch.randn(T, n, D, device=dev))
    x, y = out[..., :D], out[..., D:2 * D]
    tc = int(os.environ.get("TC", "0"))
    fn = lambda: hip.spmm_split(plan, x, y, 1.0, t_chunk=tc)
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"SGP_TUNE={os.environ.get('SGP_TUNE', '')} T={T} tc={tc}: {best:.3f} ms  ({best * 1024 / T:.2f} per 1024 steps)", flush=True)
else:
    for m in sys.argv[1:] or ["0", "1", "2", "4", "8", "3", "9", "13", "6", "14"]:
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SGP_TUNE=",".join(v for v in (os.environ.get("SGP_TUNE", ""), "split_abl=" + m) if v)))
