"""Cost of the device-side choice of the split-fp16 hop on the target graph: the operand statistics + prepare kernel, and the
exact kernel's launch when its predicate says "skip" (every workgroup exits at its first instruction)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgp_amd import synthetic, hip
from sgp_amd.graph import ShiftOperator

n, T, D = 100000, int(os.environ.get("T", "1024")), 64
dev = torch.device("cuda:0")
ei, ew, _ = synthetic.knn_graph(n, 100, seed=1)
op = ShiftOperator.from_edges(ei, ew, n)
out = torch.empty(T, n, 5 * D, device=dev)
out[..., :D] = torch.tanh(torch.randn(T, n, D, device=dev))
x, y = out[..., :D], out[..., D:2 * D]


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


plan = op.split_plan(dev)
prof = hip.split_profile(x, None, 1.0, op.norm_inf())
print("flag", int(prof.flag.item()))
print("split kernel alone           %.3f ms" % timed(lambda: hip.spmm_split(plan, x, y, prof)))
print("statistics + prepare         %.3f ms" % timed(lambda: hip.split_profile(x, None, 1.0, op.norm_inf())))
print("propagate (default dispatch) %.3f ms" % timed(lambda: op.propagate(x, y, x_bound=1.0)))
mplan = op.mix_plan(D, dev)
def skipped():
    hip.spmm_mix(mplan, x, y, None, n, pred=(prof.flag, 0))
print("exact kernel, predicate off  %.3f ms" % timed(skipped))
os.environ["SGP_TUNE"] = "split_guard=0"
print("propagate, no statistics     %.3f ms" % timed(lambda: op.propagate(x, y, x_bound=1.0)))
