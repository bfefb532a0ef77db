"""Split-J reservoir (small graphs): us per step on the METR-LA / PEMS-BAY shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgp_amd
from sgp_amd import hip
for (n, F, R, t) in [(207, 3, 64, 8000), (325, 3, 128, 4000), (207, 3, 32, 8000)]:
    res = sgp_amd.Reservoir(F, R)
    xin = torch.randn(t, n, F, device="cuda"); out = torch.empty(t, n, R, device="cuda")
    res.encode_into(xin, out); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = hip.Event(), hip.Event(); a.record(); res.encode_into(xin, out); b.record()
        best = min(best, a.elapsed_ms(b))
    print(f"N={n} R={R} T={t}: {best:.2f} ms  {best / t * 1e3:.3f} us/step", flush=True)
