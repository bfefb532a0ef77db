import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sgp_amd
import oracle.sgp_oracle as O
from sgp_amd import hip
def layers_of(res):
    return [dict(w_ih=l.w_ih.data.cpu(), w_hh=l.w_hh.data.cpu(), b_ih=l.b_ih.data.cpu(), alpha=float(l.alpha)) for l in res.reservoir_layers]
for (n, f, r, L, act, rad) in [(333, 5, 32, 4, "relu", 0.5), (333, 5, 32, 4, "tanh", 0.95), (333, 5, 32, 1, "relu", 0.5), (333, 64, 64, 1, "relu", 0.5), (333, 7, 24, 3, "self_norm", 0.95)]:
    torch.manual_seed(n + L)
    t = 70
    res = sgp_amd.Reservoir(f, r, num_layers=L, leaking_rate=0.9, spectral_radius=rad, density=0.7, alpha_decay=True, activation=act)
    x = torch.randn(t, n, f)
    out = torch.empty(t, n, L * r, device="cuda")
    res.fused = False
    res.encode_into(x.cuda(), out)
    ref64 = O.reservoir_forward(x, layers_of(res), activation=act, dtype=torch.float64)
    ref32 = O.reservoir_forward(x, layers_of(res), activation=act)
    g = out.cpu().double()
    print(n, f, r, L, act, "gpu-vs-64 max %.3e rel %.3e | cpu32-vs-64 max %.3e rel %.3e | max|ref| %.2e" % (
        float((g - ref64).abs().max()), float((g - ref64).norm() / ref64.norm()),
        float((ref32.double() - ref64).abs().max()), float((ref32.double() - ref64).norm() / ref64.norm()), float(ref64.abs().max())), flush=True)
