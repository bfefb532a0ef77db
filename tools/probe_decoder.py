"""Timing of the next-row kernels f1 (IID gather) and f4 (grouped input layer), C2-shaped embedding."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgp_amd import hip
from sgp_amd.nn.models import SGPInputEncoder


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(n): fn()
    b.record()
    return a.elapsed_ms(b) / n * 1e3


T, N, D, order, hidden, B = 4096, 325, 1280, 10, 256, 4096
emb = torch.randn(T, N, D, device="cuda")
enc = SGPInputEncoder(D, order, hidden)
si = torch.randint(0, T, (B,), dtype=torch.int32, device="cuda")
ni = torch.randint(0, N, (B,), dtype=torch.int32, device="cuda")
us_g = timeit(lambda: hip.gather_rows(emb, si, ni))
rows = hip.gather_rows(emb, si, ni)
us_l = timeit(lambda: enc(rows[:, None, :]))
us_f = timeit(lambda: enc.forward_sampled(emb, si, ni))
byt = B * D * 4
print(f"batch {B} x D_out {D}: gather {us_g:.1f} us ({byt / us_g / 1e3:.0f} GB/s read), grouped layer {us_l:.1f} us, "
      f"fused gather+layer {us_f:.1f} us ({byt / us_f / 1e3:.0f} GB/s of gathered rows)")
