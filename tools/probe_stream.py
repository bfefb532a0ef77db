"""Host-in / host-out encode through the pipelined time-chunk path against its two bounds:
the PCIe time of the same bytes and the device-resident compute time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgp_amd
from sgp_amd import synthetic

N, T, F = int(os.environ.get("N", 10000)), int(os.environ.get("T", 1008)), 64
ei, ew, _ = synthetic.knn_graph(N, 100, seed=1)
enc = sgp_amd.SGPEncoder(input_size=F, reservoir_size=64, reservoir_layers=1, leaking_rate=.9,
                         spectral_radius=.9, density=.7, input_scaling=1., receptive_field=4,
                         bidirectional=False, alpha_decay=False, global_attr=False)
x = torch.randn(T, N, F)
ops = enc.sgp_encoder.operators(N, ei, ew)
xg = x.cuda()
out = enc.encode_device(xg, ops)                       # warm-up: plans, kernels
torch.cuda.synchronize()
t0 = time.perf_counter(); out = enc.encode_device(xg, ops); torch.cuda.synchronize()
compute = time.perf_counter() - t0
# PCIe: the same bytes through pinned buffers, one direction after the other
pin_o = torch.empty(min(T, 64), N, enc.output_size, pin_memory=True)
pin_i = torch.empty(min(T, 64), N, F, pin_memory=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(0, T, 64):
    n = min(64, T - s)
    xg[s:s + n].copy_(pin_i[:n], non_blocking=True)
    pin_o[:n].copy_(out[s:s + n], non_blocking=True)
torch.cuda.synchronize()
pcie = time.perf_counter() - t0
gb = T * N * (F + enc.output_size) * 4 / 1e9
del out, xg
torch.cuda.empty_cache()
for chunk_mb in (256, 1024):
    enc.stream_chunk_bytes = chunk_mb << 20
    enc.stream_threshold_bytes = 1 << 20
    enc(x[:64], ei, ew)                                # warm-up (pinned allocations, streams)
    t0 = time.perf_counter(); host = enc(x, ei, ew); dt = time.perf_counter() - t0
    print(f"N={N} T={T}: {gb:.1f} GB over PCIe; compute {compute * 1e3:.0f} ms, PCIe alone {pcie * 1e3:.0f} ms "
          f"({gb / pcie:.1f} GB/s), pipelined host->host {dt * 1e3:.0f} ms with {chunk_mb} MB chunks = "
          f"{dt / max(compute, pcie):.2f} x max(compute, PCIe)", flush=True)
    del host
