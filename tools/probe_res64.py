"""Reservoir layer timing on the target-line shape only (N = 100k, F = R = 64), several repeats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgp_amd
from sgp_amd import hip

N, F, R, T = 100000, 64, 64, int(os.environ.get("SGP_PROBE_T", 256))
res = sgp_amd.Reservoir(F, R, activation=os.environ.get("SGP_ACT", "tanh"))
xin = torch.randn(T, N, F, device="cuda"); out = torch.empty(T, N, R, device="cuda")
res.encode_into(xin, out); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(3): res.encode_into(xin, out)
    b.record()
    best = min(best, a.elapsed_ms(b) / 3)
print(f"{os.environ.get('SGP_AMD_LIB', 'default').split('/')[-2] if os.environ.get('SGP_AMD_LIB') else 'default'}: "
      f"{best:.3f} ms  {N * T * 2 * R * (F + R) / best / 1e9:.1f} TF/s", flush=True)
