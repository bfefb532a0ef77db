import os, sys, torch
sys.path.insert(0, os.getcwd())
import sgp_amd
from sgp_amd import hip
n, F, R, t = 325, 3, 128, 4000
res = sgp_amd.Reservoir(F, R)
xin = torch.randn(t, n, F, device="cuda"); out = torch.empty(t, n, R, device="cuda")
res.encode_into(xin, out); torch.cuda.synchronize()
def timeit():
    a, b = hip.Event(), hip.Event(); a.record(); res.encode_into(xin, out); b.record(); torch.cuda.synchronize()
    return a.elapsed_ms(b)
print("alone      %.3f us/step" % (min(timeit() for _ in range(3)) / t * 1e3))
side = torch.cuda.Stream()
A = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); B = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
for load in ("bf16 gemm", "fp32 copy"):
    with torch.cuda.stream(side):
        if load == "bf16 gemm":
            for _ in range(60): C = A @ B
        else:
            big = torch.empty(1 << 28, device="cuda"); big2 = torch.empty_like(big)
            for _ in range(200): big2.copy_(big)
    print("beside %s  %.3f us/step" % (load, timeit() / t * 1e3))
    torch.cuda.synchronize()
