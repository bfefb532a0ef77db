"""Reservoir layer timings on the large-N shapes (target line and C5), optional activation sweep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgp_amd
from sgp_amd import hip


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(n): fn()
    b.record()
    return a.elapsed_ms(b) / n


N = 100000
for (F, R, T) in [(64, 64, 256), (3, 64, 256), (128, 256, 64), (128, 128, 128)]:
    for act in os.environ.get("SGP_ACTS", "tanh").split(","):
        res = sgp_amd.Reservoir(F, R, activation=act)
        xin = torch.randn(T, N, F, device="cuda"); out = torch.empty(T, N, R, device="cuda")
        ms = timeit(lambda: res.encode_into(xin, out))
        fl = N * T * 2 * R * (F + R)
        print(f"{act} F={F} R={R} T={T}: {ms:.2f} ms {fl / ms / 1e9:.1f} TF/s", flush=True)
