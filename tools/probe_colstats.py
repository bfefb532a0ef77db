"""Time sgp_col_stats_f32 on the C5 / target slab shapes for several row strides."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgp_amd import hip
for (T, N, D, out_w) in [(256, 100000, 256, 1536), (256, 100000, 64, 320)]:
    buf = torch.randn(T, N, out_w, device="cuda") if out_w * N * T * 4 < 170e9 else None
    x = buf[:, :, :D]
    for ts, rs in [(32, 1), (32, 4), (32, 13), (32, 52)]:
        hip.col_stats(x, ts, r_stride=rs); torch.cuda.synchronize()
        a, b = hip.Event(), hip.Event(); a.record()
        for _ in range(5): hip.col_stats(x, ts, r_stride=rs)
        b.record(); ms = a.elapsed_ms(b) / 5
        mb = -(-T // ts) * -(-N // rs) * D * 4 / 1e6
        print(f"T={T} N={N} D={D} (row pitch {out_w * 4} B) t_stride={ts} r_stride={rs}: {ms:.3f} ms, {mb:.0f} MB -> {mb / ms / 1e3:.2f} TB/s", flush=True)
    del x, buf
