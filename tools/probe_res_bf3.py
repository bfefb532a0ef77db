"""Time and check the three-piece bf16 reservoir kernel against the exact-fp32 one (one process per setting:
SGP_TUNE is read once).  usage: python tools/probe_res_bf3.py [N] [T] [F] [R]"""
import os, subprocess, sys

def child():
    import torch, sgp_amd
    from sgp_amd import hip
    n, t, f, r = (int(v) for v in sys.argv[2:6])
    torch.manual_seed(0)
    res = sgp_amd.Reservoir(f, r)
    x = torch.randn(t, n, f, device="cuda")
    out = torch.empty(t, n, r, device="cuda")
    for _ in range(2):
        res.encode_into(x, out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        res.encode_into(x, out)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    # error against fp64 on the first 64 nodes, 32 steps
    import oracle.sgp_oracle as O
    g = [dict(w_ih=l.w_ih.data.cpu(), w_hh=l.w_hh.data.cpu(), b_ih=l.b_ih.data.cpu(), alpha=float(l.alpha))
         for l in res.reservoir_layers]
    tt = min(t, int(os.environ.get("PROBE_TT", "64")))
    ref = O.reservoir_forward(x[:tt, :64].cpu(), g, "tanh", dtype=torch.float64)
    err = float((out[:tt, :64].cpu().double() - ref).abs().max())
    ref32 = O.reservoir_forward(x[:tt, :64].cpu(), g, "tanh")
    err32 = float((ref32.double() - ref).abs().max())
    gb = (x.numel() + out.numel()) * 4 / 1e9
    print(f"{os.environ.get('SGP_TUNE','')!r:24} {ms:8.3f} ms  {gb / ms:6.2f} TB/s...  max err vs fp64 {err:.2e} (cpu fp32 {err32:.2e})", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    child()
else:
    a = sys.argv[1:] + ["100000", "256", "64", "64"][len(sys.argv) - 1:]
    for tune in ("res_bf3=0", "res_bf3=1"):
        env = dict(os.environ, SGP_TUNE=tune)
        subprocess.run([sys.executable, __file__, "child"] + a, env=env, check=False)
