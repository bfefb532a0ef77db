"""Experiment: staging reads of rows that the next tiles in launch order do not stage again are issued
with `nt` (they should not displace rows that will be re-read: the L2 keeps a row for ~one step)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sgp_amd import graph, hip, synthetic


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = hip.Event(), hip.Event()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    return a.elapsed_ms(b) / n


def stream_mask(ps, n_tiles, look):
    """bit (16 p + wave) of tile k: none of the 4 staged rows of that piece is staged by tiles k+1 .. k+look"""
    uptr = ps["uptr"].cpu().numpy().astype(np.int64)
    ucol = ps["ucol"].cpu().numpy().astype(np.int64)
    mask = np.zeros((n_tiles, 4), dtype=np.uint32)
    sets = [np.unique(ucol[uptr[k]:uptr[k + 1]]) for k in range(n_tiles)]
    kept = tot = 0
    for k in range(n_tiles):
        nxt = np.concatenate([sets[j] for j in range(k + 1, min(n_tiles, k + 1 + look))] or [np.zeros(0, np.int64)])
        cols = ucol[uptr[k]:uptr[k + 1]]
        shared = np.isin(cols, nxt)
        npieces = (len(cols) + 3) // 4
        pad = np.zeros(npieces * 4, dtype=bool); pad[:len(cols)] = shared
        piece_keep = pad.reshape(npieces, 4).any(1)
        kept += piece_keep.sum(); tot += npieces
        for j in np.flatnonzero(~piece_keep):
            p, w = j // 16, j % 16
            bit = p * 16 + w
            mask[k, bit >> 5] |= np.uint32(1 << (bit & 31))
    return mask, kept / max(tot, 1)


def main():
    N, D = int(os.environ.get("SGP_PROBE_N", 100000)), 64
    T = int(os.environ.get("SGP_PROBE_T", 512))
    ei, ew, _ = synthetic.knn_graph(N, 100)
    op = graph.ShiftOperator.from_edges(ei, ew, N)
    x = torch.randn(T, N, D, device="cuda")
    y = torch.empty_like(x); y0 = torch.empty_like(x)
    lib = hip.load()
    lib.sgp_spmm_res_set_stream_mask.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    bytes_hop = 2 * N * T * D * 4 + op.nnz() * 8 + (N + 1) * 4
    plan = op.tile_plan(D, x.device)
    lib.sgp_spmm_res_tune(0)
    ms = timeit(lambda: op.propagate(x, y0, force="res"))
    print(f"res:            {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}", flush=True)
    for look in (1, 2, 4, 8):
        m, kept = stream_mask(plan.pipe, plan.n_tiles, look)
        md = torch.from_numpy(m.view(np.int32)).cuda()
        lib.sgp_spmm_res_set_stream_mask(md.data_ptr(), plan.n_tiles)
        ms = timeit(lambda: op.propagate(x, y, force="res"))
        print(f"res look={look} keep {kept:.2f}: {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}  equal={bool(torch.equal(y, y0))}", flush=True)
        lib.sgp_spmm_res_set_stream_mask(None, 0)
    ms = timeit(lambda: op.propagate(x, y0, force="res"))
    print(f"res:            {ms:7.2f} ms  frac {bytes_hop / ms / 1e6 / 8000:.3f}", flush=True)


if __name__ == "__main__":
    main()
