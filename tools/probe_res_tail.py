"""Reservoir layer: exact deal + split-J tail (default) against the even deal (SGP_TUNE=res_tail=0), and the
split-J kernel alone on mid-size graphs (SGP_RES_SPLITJ_MAX).  One process per setting (the knobs are read
once): python tools/probe_res_tail.py N F R T"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgp_amd
from sgp_amd import hip

N, F, R, T = (int(v) for v in sys.argv[1:5])
torch.manual_seed(0)
res = sgp_amd.Reservoir(F, R)
xin = torch.randn(T, N, F, device="cuda"); out = torch.empty(T, N, R, device="cuda")
res.encode_into(xin, out); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(3): res.encode_into(xin, out)
    b.record()
    best = min(best, a.elapsed_ms(b) / 3)
tag = f"SGP_TUNE={os.environ.get('SGP_TUNE', '')}"
print(f"N={N} F={F} R={R} T={T} {tag}: {best:.3f} ms  {N * T * 2 * R * (F + R) / best / 1e9:.1f} TF/s  "
      f"checksum {float(out.double().sum()):.6f} {float(out[-1].abs().max()):.6f}", flush=True)
