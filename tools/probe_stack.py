"""Fused multi-layer reservoir vs one launch per layer on the PV-US shape (C4: N = 5016, R = 16 x 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgp_amd
from sgp_amd import hip


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    return a.elapsed_ms(b) / n


for (N, T, F, R, L) in [(5016, 8868, 3, 16, 8), (207, 8000, 3, 32, 2), (100000, 256, 3, 16, 8), (20000, 512, 64, 64, 2)]:
    torch.manual_seed(0)
    res = sgp_amd.Reservoir(F, R, num_layers=L, leaking_rate=1.0, spectral_radius=0.99, density=0.7, alpha_decay=True)
    x = torch.randn(T, N, F, device="cuda")
    out = torch.empty(T, N, L * R, device="cuda")
    line = f"N={N} T={T} F={F} R={R}x{L}:"
    for fused in (True, False):
        res.fused = fused
        ms = timeit(lambda: res.encode_into(x, out))
        flops = N * T * (2 * R * (F + R) + (L - 1) * 2 * R * 2 * R)
        line += f"  {'fused' if fused else 'layered'} {ms:8.2f} ms ({flops / ms / 1e9:6.1f} TF/s, {N * T * 4 * (F + L * R) / ms / 1e6:7.1f} GB/s)"
    print(line, flush=True)
