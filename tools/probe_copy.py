import torch, time
x = torch.randn(256, 100000, 64, device="cuda")
y = torch.empty_like(x)
for fn, name in ((lambda: y.copy_(x), "copy 6.5 GB -> 6.5 GB"), (lambda: x.sum(), "read 6.5 GB"), (lambda: y.fill_(1.0), "write 6.5 GB"), (lambda: torch.tanh(x, out=y), "tanh r+w")):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.3f ms" % (e0.elapsed_time(e1) / 5), flush=True)
