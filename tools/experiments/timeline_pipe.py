"""Debug: per-wave s_memtime timeline of one workgroup of spmm_pipe (ablation build, SGP_PIPE_ABL=32)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGP_PIPE_ABL", "32")
import numpy as np, torch
from sgp_amd import graph, hip, synthetic
N, T, D = int(os.environ.get("SGP_PROBE_N", 100000)), int(os.environ.get("SGP_PROBE_T", 64)), 64
ei, ew, _ = synthetic.knn_graph(N, 100)
op = graph.ShiftOperator.from_edges(ei, ew, N)
x = torch.randn(T, N, D, device="cuda"); y = torch.empty_like(x)
for _ in range(2):
    op.propagate(x, y, force="pipe")
torch.cuda.synchronize()
buf = np.zeros(4 * 16 * 8, dtype=np.uint32)
lib = hip.load()
lib.sgp_spmm_pipe_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
st = buf.reshape(4, 16, 8).astype(np.int64)
base = st[0, :, 0].min()
st = st - base
plan = op.tile_plan(D, torch.device("cuda"))
g = plan.pipe["gptr"].cpu().numpy().astype(np.int64)
tile = 777 % plan.n_tiles
h = np.diff(g[tile * 32: tile * 32 + 33]).reshape(16, 2)
print("tile", tile, "quads A/B per wave:", h.tolist(), "usplit", int(plan.pipe["usplit"][tile]), "U", int(plan.uptr[tile+1]-plan.uptr[tile]))
names = ["preX", "X", "dmaB", "endA", "vm0", "Y", "endB", "fold"]
for ts in range(4):
    print("step", ts)
    for w in range(16):
        print("  w%2d" % w, " ".join("%s=%6d" % (n, v) for n, v in zip(names, st[ts, w])))
    print("  step span: first preX %d .. last fold %d ; next" % (st[ts, :, 0].min(), st[ts, :, 7].max()))
