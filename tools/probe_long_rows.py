"""Long-row operators (the full PV-US / CER-En graph shapes): ms per hop of the split kernel under four plans -- per-group
column segments (product: splitplan.build_split_passes) and tile-level slabs (tools/experiments/slab_passes.py), each in
the standard (16 waves x 224 columns) and the wide (8 x 448) form.  T steps, D = 128, scaled to the workload's 8868 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
import torch
from sgp_amd import hip, splitplan, synthetic
from sgp_amd.graph import ShiftOperator
import slab_passes

dev = torch.device("cuda:0")
T, D = int(os.environ.get("T", "1024")), 128
for n, deg in ((5016, 740), (6435, 495)):
    ei, ew, _ = synthetic.threshold_graph(n, deg, seed=1)
    op = ShiftOperator.from_edges(ei, ew, n)
    args = (op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n, n)
    x = torch.tanh(torch.randn(T, n, D, device=dev))
    ref = torch.empty_like(x)
    op.propagate(x, ref, force="csr")
    for form, lim in (("standard", hip.split_limits()), ("wide", hip.split_limits(True))):
        for kind, build in (("group segments", splitplan.build_split_passes), ("tile slabs", slab_passes.build_split_slab_passes)):
            plans = [p.to(dev) for p in build(*args, **lim)]
            y = torch.empty_like(x)
            hip.spmm_split(plans, x, y, 1.0); torch.cuda.synchronize()
            err = float((y - ref).abs().max())
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); hip.spmm_split(plans, x, y, 1.0); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            tiles = sum(p.n_tiles for p in plans)
            print(f"N={n} ~{deg}/row {form:8s} {kind:14s}: {len(plans):2d} passes {tiles:4d} tile-passes  {best:7.2f} ms per {T} steps "
                  f"= {best * 8868 / T:6.1f} ms per hop   max|y - csr| {err:.1e}", flush=True)
