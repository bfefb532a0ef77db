"""Debug: per-wave s_memtime timeline of one workgroup of spmm_mix (ablation build:
tools/build_variant.sh abl -DSGP_ABLATION, SGP_AMD_LIB=tools/variants/abl/libsgp_amd.so, SGP_TUNE=abl=128;
+1 = no staging DMA)."""
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
    op.propagate(x, y, force="mix")
torch.cuda.synchronize()
buf = np.zeros(4 * 16 * 12, dtype=np.uint32)
ctypes.CDLL(hip.LIB_PATH).sgp_spmm_mix_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
st = buf.reshape(4, 16, 12).astype(np.int64)
st = st - st[0, :, 0].min()
mp = op.mix_plan(D, torch.device("cuda"), strict=False)
tile = 777 % mp.n_tiles
gs = mp.gsup.cpu().numpy().astype(np.int64)[tile * 32: tile * 32 + 32].reshape(16, 2)
dn = np.diff(mp.dptr.cpu().numpy().astype(np.int64))[tile * 8: tile * 8 + 8].reshape(4, 2)
print("tile", tile, "sparse super-steps A/B per wave:", gs.tolist(), "dense instructions A/B per block:", dn.tolist())
order = [0, 1, 2, 9, 10, 3, 4, 5, 6, 11, 7, 8]
names = ["top", "barA", "dma1", "emit", "dnsA", "endA", "vm0A", "barB", "dma2", "dnsB", "endB", "vm0B"]
for ts in range(4):
    print("step", ts)
    for w in range(16):
        print("  w%2d" % w, " ".join("%s=%6d" % (n, st[ts, w, i]) for n, i in zip(names, order)))
d = st[:, :, order]
seg = (d[:, :, 1:] - d[:, :, :-1]).mean((0, 1))
print("mean segment (cycles): " + " | ".join("%s->%s %.0f" % (names[i], names[i + 1], seg[i]) for i in range(len(names) - 1)),
      "| step %.0f" % (d[1:, :, 0] - d[:-1, :, 0]).mean())
for w4 in range(4):
    ws = [w4, w4 + 4, w4 + 8, w4 + 12]
    print("SIMD class", w4, "phase A done after barA:", (d[1, ws, 5] - d[1, :, 1].max()).tolist(),
          "phase B done after barB:", (d[1, ws, 10] - d[1, :, 7].max()).tolist())
