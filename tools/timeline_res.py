"""Debug: per-wave s_memtime timeline of one workgroup of spmm_res (ablation build:
`make EXTRA=-DSGP_ABLATION`, SGP_TUNE=abl=128; +1 = no staging DMA, +4 = staging from L2-resident rows)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SGP_TUNE", "abl=128")
import numpy as np, torch
from sgp_amd import graph, hip, synthetic
N, T, D = int(os.environ.get("SGP_PROBE_N", 100000)), int(os.environ.get("SGP_PROBE_T", 64)), 64
ei, ew, _ = synthetic.knn_graph(N, 100)
op = graph.ShiftOperator.from_edges(ei, ew, N)
x = torch.randn(T, N, D, device="cuda"); y = torch.empty_like(x)
for _ in range(2):
    op.propagate(x, y, force="res")
torch.cuda.synchronize()
buf = np.zeros(4 * 16 * 10, dtype=np.uint32)
ctypes.CDLL(hip.LIB_PATH).sgp_spmm_res_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
st = buf.reshape(4, 16, 10).astype(np.int64)
st = st - st[0, :, 0].min()
plan = op.tile_plan(D, torch.device("cuda"))
tile = 777 % plan.n_tiles
gs = plan.pipe["gsup"].cpu().numpy().astype(np.int64)[tile * 32: tile * 32 + 32].reshape(16, 2)
print("tile", tile, "super-steps A/B per wave:", gs.tolist())
names = ["top", "barA", "dma1", "endA", "vm0A", "barB", "dma2", "endB", "vm0B"]
for ts in range(4):
    print("step", ts)
    for w in range(16):
        print("  w%2d" % w, " ".join("%s=%6d" % (n, v) for n, v in zip(names, st[ts, w, :9])))
d = st[:, :, :9]
print("mean over waves/steps (cycles): top->barA %.0f | barA->endA %.0f | endA->vm0A %.0f | vm0A->barB %.0f | barB->endB %.0f | endB->vm0B %.0f | step %.0f" % (
    (d[:, :, 1] - d[:, :, 0]).mean(), (d[:, :, 3] - d[:, :, 1]).mean(), (d[:, :, 4] - d[:, :, 3]).mean(),
    (d[:, :, 5] - d[:, :, 4]).mean(), (d[:, :, 7] - d[:, :, 5]).mean(), (d[:, :, 8] - d[:, :, 7]).mean(),
    (d[1:, :, 0] - d[:-1, :, 0]).mean()))
print("phase A: first wave done %s, last wave done %s (after barA of the step)" % (
    (d[:, :, 3].min(1) - d[:, :, 1].max(1)).tolist(), (d[:, :, 3].max(1) - d[:, :, 1].max(1)).tolist()))
print("phase B: first wave done %s, last wave done %s (after barB)" % (
    (d[:, :, 7].min(1) - d[:, :, 5].max(1)).tolist(), (d[:, :, 7].max(1) - d[:, :, 5].max(1)).tolist()))
