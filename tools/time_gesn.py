"""Time the DynGESN baseline encoder on the shipped METR-LA shape (config/traffic/gesn.yaml)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgp_amd
from sgp_amd import synthetic

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 207          # 325 = PEMS-BAY shape
f, r, L = 2, 320, 3
ei, ew = synthetic.sparse_traffic_graph(n, {207: 1515, 325: 2369}.get(n, 7 * n), seed=0)
torch.manual_seed(0)
enc = sgp_amd.GESNEncoder(f, r, L, .9, .9, .7, 1., True)
x = torch.randn(T, n, f, device="cuda")
from sgp_amd import hip
lib = hip.load()
for mode in (1, 0):
    lib.sgp_gesn_tune(mode)
    enc(x[:50], ei, ew)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = enc(x, ei, ew)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"gesn {'persistent' if mode else 'stepwise  '} T={T} N={n} R={r} L={L}: {dt*1e3:.1f} ms, {dt/T*1e6:.1f} us/step, "
          f"{T*n/dt:.3e} node-steps/s", flush=True)
lib.sgp_gesn_tune(1)
