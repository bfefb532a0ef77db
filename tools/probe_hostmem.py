"""How fast can 4 GB of embedding reach ORDINARY host memory?  Variants of the last hop of the
pipelined host path: page-fault cost of a fresh pageable tensor, threaded copy out of a pinned
slot, cudaHostRegister of the destination and direct D2H."""
import ctypes, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import torch

GB = float(os.environ.get("GB", 4))
n = int(GB * (1 << 30) / 4)
dev = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
t0 = time.perf_counter(); pin.copy_(dev, non_blocking=True); torch.cuda.synchronize()
print(f"D2H into a pinned buffer: {GB / (time.perf_counter() - t0):.1f} GB/s", flush=True)
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count(), flush=True)


def timed(label, fn):
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{label}: {dt * 1e3:.0f} ms = {GB / dt:.1f} GB/s", flush=True)


out = torch.empty(n, dtype=torch.float32)
timed("fresh pageable <- pinned, one copy_", lambda: out.copy_(pin))
timed("same again (pages present)", lambda: out.copy_(pin))
for k in (4, 16, 64):
    out = torch.empty(n, dtype=torch.float32)
    pool = ThreadPoolExecutor(k)
    cuts = [n * i // k for i in range(k + 1)]
    timed(f"fresh pageable <- pinned, {k} threads", lambda: list(pool.map(lambda i: out[cuts[i]:cuts[i + 1]].copy_(pin[cuts[i]:cuts[i + 1]]), range(k))))
libc = ctypes.CDLL("libc.so.6")
out = torch.empty(n, dtype=torch.float32)
rc = libc.madvise(ctypes.c_void_p(out.data_ptr() & ~4095), ctypes.c_size_t(n * 4), 14)
print("madvise(MADV_HUGEPAGE) rc", rc, open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), flush=True)
pool = ThreadPoolExecutor(16)
cuts = [n * i // 16 for i in range(17)]
timed("fresh pageable + MADV_HUGEPAGE <- pinned, 16 threads", lambda: list(pool.map(lambda i: out[cuts[i]:cuts[i + 1]].copy_(pin[cuts[i]:cuts[i + 1]]), range(16))))
rt = torch.cuda.cudart()
out = torch.empty(n, dtype=torch.float32)
t0 = time.perf_counter(); rc = rt.cudaHostRegister(out.data_ptr(), n * 4, 0); dt = time.perf_counter() - t0
print(f"cudaHostRegister of fresh pageable memory: rc={rc} {dt * 1e3:.0f} ms = {GB / dt:.1f} GB/s", flush=True)
t0 = time.perf_counter(); out.copy_(dev, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"D2H straight into the registered tensor: {dt * 1e3:.0f} ms = {GB / dt:.1f} GB/s  equal={bool(torch.equal(out, pin))}", flush=True)
t0 = time.perf_counter(); rc = rt.cudaHostUnregister(out.data_ptr()); dt = time.perf_counter() - t0
print(f"cudaHostUnregister: rc={rc} {dt * 1e3:.0f} ms", flush=True)
k = 8
out = torch.empty(n, dtype=torch.float32)
pool = ThreadPoolExecutor(k)
cuts = [(n * i // k) // 1024 * 1024 for i in range(k + 1)]
t0 = time.perf_counter()
list(pool.map(lambda i: rt.cudaHostRegister(out.data_ptr() + 4 * cuts[i], 4 * (cuts[i + 1] - cuts[i]), 0), range(k)))
dt = time.perf_counter() - t0
print(f"cudaHostRegister in {k} threads: {dt * 1e3:.0f} ms = {GB / dt:.1f} GB/s", flush=True)
