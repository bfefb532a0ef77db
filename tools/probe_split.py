"""Split-fp16 hop: correctness against the generic CSR kernel / fp64 and hop time on the target graph.
python tools/probe_split.py [N] [T] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgp_amd import synthetic, hip
from sgp_amd.graph import ShiftOperator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
D, P = 64, 5
dev = torch.device("cuda:0")
t0 = time.time()
ei, ew, _ = synthetic.knn_graph(n, k, seed=1)
op = ShiftOperator.from_edges(ei, ew, n)
print(f"graph {time.time() - t0:.1f}s", flush=True)
t0 = time.time()
plan = op.split_plan(dev)
print(f"split plan {time.time() - t0:.1f}s", plan.stats if plan else None, flush=True)
torch.manual_seed(0)
out = torch.empty(T, n, P * D, device=dev)
out[..., :D] = torch.tanh(torch.randn(T, n, D, device=dev))
x, y = out[..., :D], out[..., D:2 * D]
yref = torch.empty(min(T, 4), n, D, device=dev)
op.propagate(x[:yref.shape[0]], yref, force="csr")
op.propagate(x, y, force="split", x_bound=1.0)
torch.cuda.synchronize()
d = (y[:yref.shape[0]] - yref).abs().max().item()
print(f"split vs csr kernel: max|diff| {d:.3e}  (scale {yref.abs().max().item():.3f})")
d2 = (y[T - 1] - (op.propagate(x[T - 1:], yref[:1], force='csr'))[0]).abs().max().item()
print(f"last step: {d2:.3e}")
# fp64 check of one step on the host
A = torch.sparse_csr_tensor(op.rowptr.long(), op.col.long(), op.val.double(), (n, n))
ref64 = (A @ x[0].double().cpu())
e_split = (y[0].cpu().double() - ref64).abs().max().item()
e_csr = (yref[0].cpu().double() - ref64).abs().max().item()
print(f"vs fp64: split {e_split:.3e}  exact-fp32 csr kernel {e_csr:.3e}")

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(reps):
        ev[0].record(); fn(); ev[1].record(); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]))
    return best
bytes_hop = 2 * n * T * D * 4 + op.nnz() * 8 + (n + 1) * 4
for name, kw in (("split", dict(force="split", x_bound=1.0)), ("mix", dict(force="mix"))):
    try:
        ms = timeit(lambda: op.propagate(x, y, **kw))
        print(f"{name}: {ms:.3f} ms per hop of {T} steps = {bytes_hop / ms / 1e9:.0f} GB/s = {bytes_hop / ms / 8e9:.3f} of 8 TB/s"
              f"  ({ms * 1024 / T:.2f} ms per 1024 steps)")
    except Exception as e:
        print(name, "failed:", e)
for tc in (8, 16, 32, 64, 128):
    ms = timeit(lambda: hip.spmm_split(plan, x, y, 1.0, t_chunk=tc))
    print(f"split t_chunk {tc}: {ms:.3f} ms  {bytes_hop / ms / 8e9:.3f}")
