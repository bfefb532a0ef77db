"""Experiment (round 6, not product): long-row operators by 2-D blocking at the TILE level -- a tile's sorted column
union cut into slabs that every wave's column budget admits, pass p = slab p of every tile.  Stages every column of a
tile once per unit (PV-US full graph: 7.3 staged rows per result row against 16.7 for per-group column segments) but
needs MORE tile-passes (161 against 124: the band structure lets one wave reach its 224-column budget while the others
hold few, chunk fill 0.5), and a unit's cost is dominated by its fixed part (16 waves x 7 chunks of operand reads and
MFMAs, padded or not): no gain over splitplan.build_split_passes.  Kept for the record; run:
    python tools/experiments/slab_passes.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgp_amd.splitplan import SplitPlan  # noqa: E402


def _fill_plan(rowptr, col, val, n_rows, n_cols, wave_of_row, slot_of_row, tile_of_wave, rows, waves, chunks, max_union,
               threads=0):
    """The kernel's arrays for an EXPLICIT deal (``sgp_split_plan_fill``): which wave / slot every row takes, which tile
    every wave belongs to."""
    import ctypes
    from sgp_amd import hip
    lib = hip.load()
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rowptr, col = np.ascontiguousarray(rowptr, dtype=np.int64), np.ascontiguousarray(col, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float32)
    wave_of_row, slot_of_row = np.ascontiguousarray(wave_of_row, dtype=np.int64), np.ascontiguousarray(slot_of_row, dtype=np.int64)
    tile_of_wave, rows = np.ascontiguousarray(tile_of_wave, dtype=np.int64), np.ascontiguousarray(rows, dtype=np.int64)
    n_waves, n_tiles = int(tile_of_wave.size), int(tile_of_wave[-1]) + 1
    hdr = torch.empty((n_tiles, 64), dtype=torch.int32)
    rowid = torch.empty((n_tiles, waves, 16), dtype=torch.int32)
    ucol = torch.empty((n_tiles, max_union), dtype=torch.int32)
    afr = torch.empty((n_tiles, waves, chunks, 2, 64, 8), dtype=torch.float16)
    adr = torch.empty((n_tiles, waves, chunks, 64), dtype=torch.int32)
    rinv = torch.empty((n_tiles, waves, 16), dtype=torch.float32)
    st = np.zeros(8, dtype=np.float64)
    hip._check(lib.sgp_split_plan_fill(ptr(rowptr), ptr(col), ptr(val), n_rows, n_cols, ptr(wave_of_row), ptr(slot_of_row),
                                       ptr(tile_of_wave), ptr(rows), n_waves, n_tiles, waves, chunks, max_union,
                                       hdr.data_ptr(), rowid.data_ptr(), ucol.data_ptr(), afr.data_ptr(), adr.data_ptr(),
                                       rinv.data_ptr(), ptr(st), int(threads)), "sgp_split_plan_fill")
    stats = dict(tiles=n_tiles, waves=n_waves, rows_per_wave=float(st[2]), rows_per_tile=float(st[3]),
                 staged_per_row=float(st[4]), chunk_fill=float(st[5]), max_union=int(st[6]))
    return SplitPlan(hdr, rowid, ucol, afr, adr, rinv, n_tiles, n_rows, n_cols, float(st[7]), stats)


def build_split_slab_passes(rowptr, col, val, n_rows, n_cols, waves=16, chunks=7, max_union=768, rows_per_wave=16,
                            max_passes=12):
    """Plans for an operator with LONG rows by 2-D blocking at the TILE level (round 6; the reference's full PV-US /
    CER-En graphs, config/largescale/sgp_pv.yaml + experiments/run_largescale_sgp.py:167-170).  A tile is ``waves`` x
    ``rows_per_wave`` consecutive rows; its sorted column union is cut into SLABS -- as long as possible with at most
    ``max_union`` columns and at most ``32 * chunks`` columns of any one wave -- and pass p multiplies slab p of every tile
    that has one (from the second pass on: accumulating).  Every column of a tile's union is therefore staged ONCE per
    unit across the passes -- 7 staged rows per result row on the PV-US shape, where per-group column segments
    (``build_split_passes``) stage 26 -- at the price of waves whose rows own few columns of a slab (their chunks are
    padding).  Returns a list of plans or None."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    if n_rows == 0 or col.size == 0 or not np.isfinite(val).all():
        return None
    cap, R = 32 * chunks, waves * rows_per_wave
    n_tiles = -(-n_rows // R)
    row_of_edge = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr[:n_rows + 1]))
    tile_of_edge, wave_of_edge = row_of_edge // R, (row_of_edge % R) // rows_per_wave
    slab_of_edge = np.empty(col.size, dtype=np.int64)
    n_slabs = np.zeros(n_tiles, dtype=np.int64)
    e0 = np.searchsorted(tile_of_edge, np.arange(n_tiles + 1))
    for t in range(n_tiles):
        a, b = e0[t], e0[t + 1]
        if a == b:
            continue
        u, inv = np.unique(col[a:b], return_inverse=True)             # the tile's sorted union; column k of every entry
        # a (wave, column) pair counts once however many of the wave's rows hold the column
        pair = np.unique(wave_of_edge[a:b] * u.size + inv)
        use = np.zeros((waves, u.size + 1), dtype=np.int64)
        use[pair // u.size, pair % u.size + 1] = 1
        use = np.cumsum(use, axis=1)                                   # use[w, k] = columns of wave w among the first k
        cut, k = [0], 0
        while k < u.size:
            hi = min(u.size, k + max_union)
            # the furthest end <= hi at which no wave exceeds its column budget (use is non-decreasing: binary search per wave)
            for w in range(waves):
                hi = min(hi, int(np.searchsorted(use[w], use[w, k] + cap, side="right")) - 1)
            if hi <= k:
                return None
            cut.append(hi)
            k = hi
        n_slabs[t] = len(cut) - 1
        slab_of_edge[a:b] = np.searchsorted(np.asarray(cut[1:]), inv, side="right")
    n_pass = int(n_slabs.max())
    if n_pass > max_passes or n_pass == 0:
        return None
    plans = []
    rows_in_tile = np.minimum(R, n_rows - np.arange(n_tiles) * R)
    for p in range(n_pass):
        live = np.flatnonzero((n_slabs > p) | ((p == 0)))             # the first pass writes every row (also empty tiles)
        take = slab_of_edge == p
        cnt = np.bincount(row_of_edge[take], minlength=n_rows)
        rp = np.zeros(n_rows + 1, dtype=np.int64)
        rp[1:] = np.cumsum(cnt)
        new_tile = np.full(n_tiles, -1, dtype=np.int64)
        new_tile[live] = np.arange(live.size)
        r = np.arange(n_rows, dtype=np.int64)
        in_live = new_tile[r // R] >= 0
        waves_of_tile = -(-rows_in_tile[live] // rows_per_wave)
        first_wave = np.concatenate([[0], np.cumsum(waves_of_tile)])
        wave_of_row = np.where(in_live, first_wave[np.maximum(new_tile[r // R], 0)] + (r % R) // rows_per_wave, -1)
        slot_of_row = np.where(in_live, r % rows_per_wave, -1)
        tile_of_wave = np.repeat(np.arange(live.size, dtype=np.int64), waves_of_tile)
        rows = np.bincount(wave_of_row[in_live], minlength=int(first_wave[-1]))
        plan = _fill_plan(rp, col[take], val[take], n_rows, n_cols, wave_of_row, slot_of_row, tile_of_wave, rows,
                          waves, chunks, max_union)
        plan.accumulate = p > 0
        plans.append(plan)
    return plans




if __name__ == "__main__":
    from sgp_amd import hip, splitplan, synthetic
    from sgp_amd.graph import ShiftOperator
    for n, deg in ((5016, 740), (6435, 495)):
        ei, ew, _ = synthetic.threshold_graph(n, deg, seed=1)
        op = ShiftOperator.from_edges(ei, ew, n)
        args = (op.rowptr.numpy(), op.col.numpy(), op.val.numpy(), n, n)
        for name, lim in (("standard", hip.split_limits()), ("wide", hip.split_limits(True))):
            slabs = build_split_slab_passes(*args, **lim)
            groups = splitplan.build_split_passes(*args, **lim)
            per_row = lambda ps: sum(q.stats["staged_per_row"] * q.stats["rows_per_tile"] * q.n_tiles for q in ps) / n
            print(f"N = {n}, ~{deg} per row, {name} form: slab passes {len(slabs)} ({sum(q.n_tiles for q in slabs)} tile-passes, "
                  f"{per_row(slabs):.1f} staged rows per result row) | group passes {len(groups)} "
                  f"({sum(q.n_tiles for q in groups)} tile-passes, {per_row(groups):.1f})")
