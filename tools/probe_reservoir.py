import os, sys, torch
sys.path.insert(0, "/root/repo")
import sgp_amd
from sgp_amd import hip
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(n): fn()
    b.record()
    return a.elapsed_ms(b) / n
N, T = 100000, 256
for act in ("tanh", "relu", "identity"):
    for (F, R) in [(64, 64), (3, 64)]:
        try:
            res = sgp_amd.Reservoir(F, R, activation=act)
        except Exception as e:
            print(act, "ctor:", e); continue
        xin = torch.randn(T, N, F, device="cuda"); out = torch.empty(T, N, R, device="cuda")
        try:
            ms = timeit(lambda: res.encode_into(xin, out))
        except Exception as e:
            print(act, "run:", e); continue
        fl = N * T * 2 * R * (F + R)
        print(f"{act} F={F} R={R}: {ms:.2f} ms {fl / ms / 1e9:.1f} TF/s", flush=True)
