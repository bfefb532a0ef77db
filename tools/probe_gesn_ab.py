"""Persistent DynGESN kernel vs the stepwise path: max |diff| and both against the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgp_amd
from sgp_amd import hip, synthetic
from sgp_amd.nn.encoders.dyn_gesn_encoder import gesn_operator
from oracle import sgp_oracle as O

lib = hip.load()
for (n, f, r, L, act, t) in [(207, 2, 320, 3, "tanh", 40), (70, 3, 64, 1, "relu", 300), (500, 2, 48, 4, "tanh", 20)]:
    torch.manual_seed(n + r)
    ei, ew = synthetic.sparse_traffic_graph(n, 7 * n, seed=n)
    res = sgp_amd.GraphESN(f, r, num_layers=L, leaking_rate=0.9, spectral_radius=0.9, density=0.7,
                           activation=act, alpha_decay=True)
    op = gesn_operator(ei, ew, n)
    x = torch.randn(1, t, n, f)
    lib.sgp_gesn_tune(1); y1, h1 = res(x.cuda(), op)
    lib.sgp_gesn_tune(0); y0, h0 = res(x.cuda(), op)
    lib.sgp_gesn_tune(1)
    layers = [dict(w_ih=l.w_ih.data, w_hh=l.w_hh.data, b_ih=l.b_ih.data, alpha=float(l.alpha)) for l in res.rnn_cells]
    ref64 = O.gesn_forward(x[0], ei, ew, layers, activation=act, dtype=torch.float64)
    ref32 = O.gesn_forward(x[0], ei, ew, layers, activation=act)
    e = lambda y: float((y[0].double().cpu() - ref64).abs().max())
    print(f"n={n} r={r} L={L} {act} t={t}: |persistent - stepwise| = {float((y1 - y0).abs().max()):.3g}  "
          f"vs fp64: persistent {e(y1):.3g} stepwise {e(y0):.3g} cpu-fp32 {float((ref32.double() - ref64).abs().max()):.3g}", flush=True)
