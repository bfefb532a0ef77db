import os, sys
sys.path.insert(0, "/root/repo")
import torch, sgp_amd
from sgp_amd import hip
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = hip.Event(), hip.Event(); a.record()
    for _ in range(n): fn()
    b.record(); return a.elapsed_ms(b) / n
T, F, R, L = 4000, 3, 16, 8
for N in (1024, 2048, 4096, 5016, 8192, 16384):
    res = sgp_amd.Reservoir(F, R, num_layers=L, leaking_rate=1.0, spectral_radius=0.99, density=0.7, alpha_decay=True)
    x = torch.randn(T, N, F, device="cuda"); out = torch.empty(T, N, L * R, device="cuda")
    ms = timeit(lambda: res.encode_into(x, out))
    print(f"N={N:6d} tiles={N // 16:5d}: {ms:7.2f} ms = {ms / (T + L - 1) * 1e3:.3f} us per iteration", flush=True)
