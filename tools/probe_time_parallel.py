"""Time pieces of the small-graph reservoir (ReservoirLayer._run_time_parallel): splice gaps and times per warm-up
length, C1 / C2 shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgp_amd
from sgp_amd import tune

dev = torch.device("cuda:0")
for name, N, T, R, alpha in (("c1", 207, 34272, 64, 0.9), ("c2", 325, 52116, 128, 0.9), ("c2 a=0.8", 325, 52116, 128, 0.8)):
    torch.manual_seed(42)
    res = sgp_amd.Reservoir(3, R, leaking_rate=alpha, spectral_radius=0.9, density=0.7)
    layer = res.reservoir_layers[0]
    x = torch.randn(T, N, 3, device=dev)
    out = torch.empty(T, N, R, device=dev)
    ref = torch.empty(T, N, R, device=dev)
    os.environ["SGP_TUNE"] = "time_parallel=0"
    layer.run_sequence(x, ref); torch.cuda.synchronize()
    t0 = time.time(); layer.run_sequence(x, ref); torch.cuda.synchronize(); t_seq = time.time() - t0
    for warm in (64, 128, 192, 256, 384, 512):
        os.environ["SGP_TUNE"] = f"time_parallel_warm={warm},time_parallel_tol=1e-30"     # (never accept: the gap is what we read)
        layer.run_sequence(x, out); torch.cuda.synchronize()
        info = layer.last_time_parallel
        gap = float(info["gap"])
        os.environ["SGP_TUNE"] = f"time_parallel_warm={warm}"
        layer.run_sequence(x, out); torch.cuda.synchronize()
        t0 = time.time(); layer.run_sequence(x, out); torch.cuda.synchronize(); dt = time.time() - t0
        info = layer.last_time_parallel
        print(f"{name}: warm {warm:4d} pieces {info['pieces']:3d} x {info['steps']} steps: gap {gap:.3e} accepted {int(info['flag'])} "
              f"max|out - seq| {float((out - ref).abs().max()):.3e}  {dt * 1e3:.2f} ms (sequential {t_seq * 1e3:.2f} ms)", flush=True)
