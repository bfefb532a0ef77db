"""Host plan of ``sgp_spmm_blk_f32`` (include/sgp_amd.h): the row-block form of the row-group SpMM
that stands behind ``x = adj @ x`` (reference: lib/sgp_preprocessing.py:202).

A tile is up to ``4 * 4 * waves`` consecutive rows (128 with 8 waves).  Rows are clustered into
groups of 4 that share most of their source columns (``graph.cluster_rows_in_tiles``); a wave owns
FOUR groups, one per 16-lane class of ``v_mfma_f32_4x4x1_16b_f32``: in a super-step lane (q, li)
reads 16 bytes of class q's next source row and the 4 MFMAs add that column's contribution to the
4 rows of group q -- every class walks its OWN column list, so the accumulators of a lane already
hold finished sums for (row, 4 features) and no cross-lane fold is needed.  The price is padding:
a wave runs max(class lengths) super-steps per phase, so groups of similar length share a wave.

As in the two-phase kernels the tile's distinct-column list is cut by position parity into
segment A | B (staged alternately by LDS-DMA), and every class's list is stored A part first.

Arrays (all int32 / float32, one entry per tile unless noted):
  uptr[n_tiles + 1], ucol[...]   staged source rows of a tile (A padded to a multiple of 4 rows)
  usplit[n_tiles]                rows of segment A
  wptr[2 * W * n_tiles + 1]      first super-step of (tile, wave, phase) in ``soff`` / ``sw``
  nsteps[2 * W * n_tiles]        super-steps of that range (its longest class)
  soff[n_super, 4]               LDS byte offset of class q's staged row in that super-step
  sw[n_super4, 64]               weights, one float per lane per 4 super-steps: lane
                                 (q, b, i) = row i of class q in super-step 4 p + b; every
                                 (tile, wave, phase) range starts on a multiple of 4 super-steps
  rowmap[n_tiles * 16 * W]       output row of (tile, wave, class, i), -1 = none
"""
from dataclasses import dataclass

import numpy as np
import torch

from .graph import GROUP_ROWS, cluster_rows_in_tiles, split_tiles, tile_unions


@dataclass
class RowBlockPlan:
    uptr: torch.Tensor
    ucol: torch.Tensor
    usplit: torch.Tensor
    wptr: torch.Tensor
    nsteps: torch.Tensor
    soff: torch.Tensor
    sw: torch.Tensor
    rowmap: torch.Tensor
    n_tiles: int
    n_rows: int
    waves: int
    max_union: int              # staged rows of the largest tile (padded)
    max_steps: int              # longest (wave, phase) range in super-steps
    fill: float                 # useful / issued FMAs
    tile_rows: int
    reordered: bool = False

    def to(self, device):
        mv = lambda t: t.to(device)
        return RowBlockPlan(mv(self.uptr), mv(self.ucol), mv(self.usplit), mv(self.wptr),
                            mv(self.nsteps), mv(self.soff), mv(self.sw), mv(self.rowmap), self.n_tiles, self.n_rows,
                            self.waves, self.max_union, self.max_steps, self.fill, self.tile_rows,
                            self.reordered)


def _deal_groups(na, nb, waves):
    """(wave, class) of every group of ONE tile.  ``na, nb``: columns per group in segment A / B.
    Groups of similar length share a wave (a wave runs its longest class), then the waves are
    paired onto the 4 SIMDs (wave slots w and w + 4 share one) longest with shortest."""
    g = len(na)
    order = np.argsort(-(na + nb), kind="stable")
    n_w = (g + 3) // 4
    cost = np.zeros(waves, dtype=np.int64)
    members = [order[4 * k:4 * k + 4] for k in range(n_w)]
    for k, m in enumerate(members):
        cost[k] = na[m].max() + nb[m].max()
    # SIMD pairing: heaviest wave with the lightest one
    by_cost = np.argsort(-cost[:n_w], kind="stable").tolist() + list(range(n_w, waves))
    half = waves // 2
    slot_of_rank = [0] * waves
    for r in range(waves):
        slot_of_rank[r] = r if r < half else (waves - 1 - r) + half     # ranks r and W-1-r -> slots k, k + half
    wave_slot = np.full(waves, -1, dtype=np.int64)
    for r, k in enumerate(by_cost):
        wave_slot[k] = slot_of_rank[r]
    out_wave = np.empty(g, dtype=np.int64)
    out_cls = np.empty(g, dtype=np.int64)
    for k, m in enumerate(members):
        out_wave[m] = wave_slot[k]
        out_cls[m] = np.arange(len(m))
    return out_wave, out_cls


def build_rowblock_plan(rowptr, col, val, n_rows, max_union, waves=8, cluster=True, order=None):
    """Plan for an operator in CSR form; None when no tiling of >= 32 rows fits ``max_union``
    staged rows (graphs without locality).  ``order`` (optional, new id k = old id ``order[k]``)
    tiles the renumbered operator but keeps ``ucol`` / ``rowmap`` in the ORIGINAL ids."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col)
    val = np.asarray(val, dtype=np.float32)
    if n_rows == 0 or col.size == 0:
        return None
    if order is not None:
        import scipy.sparse as sp
        order = np.asarray(order, dtype=np.int64)
        pos = np.empty(n_rows, dtype=np.int64)
        pos[order] = np.arange(n_rows)
        rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
        a = sp.csr_matrix((val, (pos[rows], pos[col.astype(np.int64)])), shape=(n_rows, n_rows))
        a.sort_indices()
        plan = build_rowblock_plan(a.indptr.astype(np.int64), a.indices.astype(np.int32),
                                   a.data.astype(np.float32), n_rows, max_union, waves, cluster)
        if plan is None:
            return None
        o32 = torch.from_numpy(order.astype(np.int32))
        plan.ucol = o32[plan.ucol.long()]
        rm = plan.rowmap.long()
        plan.rowmap = torch.where(rm >= 0, o32[rm.clamp_min(0)].long(), rm).int()
        plan.reordered = True
        return plan
    G = 4 * waves                                   # groups per tile
    tile_rows = GROUP_ROWS * G
    trow = None
    for tr in (tile_rows, tile_rows // 2, tile_rows // 4):
        if tr < 32:
            break
        trow = split_tiles(rowptr, col, n_rows, tr, max_union - 4, min_rows=8)
        if trow is not None and len(trow) - 1 <= 1.5 * ((n_rows + tr - 1) // tr) + 1:
            break
        trow = None
    if trow is None:
        return None
    n_tiles = len(trow) - 1
    uptr, ucol, lcol, row_of_edge = tile_unions(rowptr, col, trow)
    tile_of_row = np.repeat(np.arange(n_tiles, dtype=np.int64), np.diff(trow))
    slot = cluster_rows_in_tiles(trow, uptr, lcol, row_of_edge) if cluster else \
        np.arange(n_rows, dtype=np.int64) - trow[tile_of_row]
    grp_of_row = tile_of_row * G + slot // GROUP_ROWS          # provisional group id
    row_in_grp = slot % GROUP_ROWS
    key = grp_of_row[row_of_edge] * 65536 + lcol
    uniq, inv = np.unique(key, return_inverse=True)            # one entry per (group, column)
    g_s, lc = uniq >> 16, uniq & 0xffff
    seg = lc & 1                                               # parity cut of the tile's list
    n_groups = n_tiles * G
    na = np.bincount(g_s, weights=(seg == 0), minlength=n_groups).astype(np.int64).reshape(n_tiles, G)
    nb = np.bincount(g_s, weights=(seg == 1), minlength=n_groups).astype(np.int64).reshape(n_tiles, G)
    heights = np.diff(trow)
    wave_of_g = np.zeros((n_tiles, G), dtype=np.int64)
    cls_of_g = np.zeros((n_tiles, G), dtype=np.int64)
    for k in range(n_tiles):
        ng = int((heights[k] + GROUP_ROWS - 1) // GROUP_ROWS)
        w_, c_ = _deal_groups(na[k, :ng], nb[k, :ng], waves)
        wave_of_g[k, :ng], cls_of_g[k, :ng] = w_, c_
        # unused group ids of a short tile: park them on distinct (wave, class) pairs
        used = set((int(a), int(b)) for a, b in zip(w_, c_))
        free = [(a, b) for a in range(waves) for b in range(4) if (a, b) not in used]
        for j, (a, b) in zip(range(ng, G), free):
            wave_of_g[k, j], cls_of_g[k, j] = a, b
    wave_of_g, cls_of_g = wave_of_g.reshape(-1), cls_of_g.reshape(-1)
    # staged layout: even positions of the tile's list -> A, odd -> B
    U = np.diff(uptr)
    usplit = ((U + 1) // 2 + 3) // 4 * 4
    upad = usplit + U // 2
    t_s = g_s // G
    stage_slot = np.where(seg == 0, lc >> 1, usplit[t_s] + (lc >> 1))
    uptr2 = np.zeros(n_tiles + 1, dtype=np.int64)
    uptr2[1:] = np.cumsum(upad)
    first_col = ucol[np.minimum(uptr[:-1], max(len(ucol) - 1, 0))]
    ucol2 = np.repeat(first_col, upad).astype(np.int32)
    tile_of_u = np.repeat(np.arange(n_tiles, dtype=np.int64), U)
    l_of_u = np.arange(len(ucol), dtype=np.int64) - uptr[tile_of_u]
    s_of_u = np.where((l_of_u & 1) == 0, l_of_u >> 1, usplit[tile_of_u] + (l_of_u >> 1))
    ucol2[uptr2[tile_of_u] + s_of_u] = ucol
    # super-step ranges per (tile, wave, phase): length = longest class, padded to 4 super-steps
    # in the WEIGHT array only (sw rows hold 4 super-steps); soff is indexed by super-step
    wp = (t_s * waves + wave_of_g[g_s]) * 2 + seg              # (tile, wave, phase) of every entry
    cls = cls_of_g[g_s]
    n_wp = n_tiles * waves * 2
    cnt = np.zeros((n_wp, 4), dtype=np.int64)
    np.add.at(cnt, (wp, cls), 1)
    steps = cnt.max(1)                                         # super-steps of every range
    steps4 = (steps + 3) // 4
    wptr = np.zeros(n_wp + 1, dtype=np.int64)
    wptr[1:] = np.cumsum(steps4 * 4)                           # ranges start on multiples of 4
    n_super = int(wptr[-1])
    # position of every entry inside its (range, class) list, ordered by staged slot
    o = np.lexsort((stage_slot, cls, wp))
    wp_o, cls_o = wp[o], cls[o]
    run = wp_o * 4 + cls_o
    start = np.r_[0, np.flatnonzero(np.diff(run)) + 1]
    run_id = np.cumsum(np.r_[0, (np.diff(run) != 0).astype(np.int64)])
    p_o = np.arange(o.size, dtype=np.int64) - start[run_id]
    p = np.empty_like(p_o)
    p[o] = p_o
    sup = wptr[wp] + p                                         # global super-step of every entry
    soff = np.zeros((max(n_super, 1), 4), dtype=np.int32)      # padding: staged row 0, weight 0
    soff[sup, cls] = (stage_slot * 256).astype(np.int32)
    sw = np.zeros((max(n_super // 4, 1), 4, 4, GROUP_ROWS), dtype=np.float32)   # [.., q, b, i]
    e_sup = sup[inv]
    sw[e_sup // 4, cls[inv], e_sup % 4, row_in_grp[row_of_edge]] = val
    fill = float(col.size) / max(1, int(steps.sum()) * 4 * GROUP_ROWS)
    rowmap = np.full(n_tiles * GROUP_ROWS * G, -1, dtype=np.int32)
    g_of_row = grp_of_row
    rowmap[(tile_of_row * waves + wave_of_g[g_of_row]) * 16 + cls_of_g[g_of_row] * 4 + row_in_grp] = \
        np.arange(n_rows, dtype=np.int32)
    return RowBlockPlan(torch.from_numpy(uptr2.astype(np.int32)), torch.from_numpy(ucol2),
                        torch.from_numpy(usplit.astype(np.int32)), torch.from_numpy(wptr.astype(np.int32)),
                        torch.from_numpy(steps.astype(np.int32)), torch.from_numpy(soff), torch.from_numpy(sw.reshape(-1, 64)),
                        torch.from_numpy(rowmap), n_tiles, int(n_rows), waves,
                        int(upad.max(initial=0)), int(steps.max(initial=0)), fill,
                        int(heights.max(initial=0)))


def plan_reference(plan: RowBlockPlan, x):
    """y = A x evaluated FROM THE PLAN ARRAYS on the host (numpy, fp64 accumulation): test
    infrastructure that checks the planner independently of the kernel."""
    x = np.asarray(x, dtype=np.float64)                        # [N_cols, F]
    y = np.zeros((plan.n_rows, x.shape[1]))
    uptr, ucol = plan.uptr.numpy(), plan.ucol.numpy()
    wptr, soff, sw = plan.wptr.numpy(), plan.soff.numpy(), plan.sw.numpy().reshape(-1, 4, 4, 4)
    nsteps = plan.nsteps.numpy()
    rowmap = plan.rowmap.numpy()
    W = plan.waves
    for t in range(plan.n_tiles):
        staged = x[ucol[uptr[t]:uptr[t + 1]]]
        for w in range(W):
            for ph in range(2):
                r = (t * W + w) * 2 + ph
                for s in range(wptr[r], wptr[r] + nsteps[r]):
                    for q in range(4):
                        xr = staged[soff[s, q] // 256] if soff[s, q] // 256 < len(staged) else 0.0
                        for i in range(4):
                            wt = sw[s // 4, q, s % 4, i]
                            if wt != 0.0:
                                row = rowmap[(t * W + w) * 16 + q * 4 + i]
                                y[row] += wt * xr
    return y
