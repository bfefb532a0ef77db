"""Host plan of ``sgp_spmm_mix_f32`` (include/sgp_amd.h): the mixed dense / sparse form of the
row-group SpMM behind ``x = adj @ x`` (reference: lib/sgp_preprocessing.py:200-203).

Tiles, staged rows, the parity cut into segments A | B and the 4-row groups are those of the
two-phase kernels (``graph.build_phase_stream``).  New: the 16 groups of a tile form 4 BLOCKS of
16 rows, and a block's columns are split in two sets --

* **dense** columns: used by at least ``thr`` of the block's 4 groups.  They go through
  ``v_mfma_f32_16x16x4_f32`` -- 16 rows x 4 columns x 16 features per instruction, one 4-byte LDS
  read per lane, no cross-lane fold: wave (block, fq) computes the block's dense part for the
  feature quarter fq.  A (block, phase) list holds a multiple of 4 columns (what does not fill a
  whole instruction is demoted to the sparse set, least shared columns first) and at most ``dh``
  instructions (the kernel keeps the list in registers).
* **sparse** columns: everything else, through the 4x4x1 row-group stream exactly as before (the
  wave of group g walks group g's remaining columns for all 64 features).

Arrays (int32 / float32):
  uptr, ucol, usplit, rowmap, gptr, gsup, gidx, gw   as in ``build_phase_stream`` (sparse part only)
  dptr[2 * 4 * n_tiles + 1]    first dense instruction of (tile, block, phase)
  didx[n_dense, 4]             LDS byte offsets of the 4 staged rows (columns k = 0..3) of an instruction
  dw[n_dense, 64]              its A operand in lane order: lane 16 k + i = weight of (row i of the block, column k)
Slot s of a tile = 16 block + 4 group-in-block + row-in-group; wave w = s // 4 owns sparse group w,
dense block w // 4 and feature quarter w % 4 (= its SIMD class, which ``_deal_classes`` balances).
"""
import itertools

import numpy as np
import torch

from .graph import GROUP_ROWS, GROUPS_PER_TILE, build_phase_stream, tile_unions

BLOCK_GROUPS = 4
BLOCKS_PER_TILE = GROUPS_PER_TILE // BLOCK_GROUPS
TILE_SLOTS = GROUP_ROWS * GROUPS_PER_TILE
_PERMS = np.array(list(itertools.permutations(range(BLOCK_GROUPS))), dtype=np.int64)     # [24, 4]


def _cluster_blocks(member):
    """Blocks of 4 groups that share most columns: ``member`` = bool [16, U] (group uses column).
    Greedy like ``graph.cluster_rows_in_tiles``: seed with the group that overlaps least with the
    free ones (a corner of the tile), add the 3 groups that overlap most with the block.  Returns
    the block of every group."""
    m = member.astype(np.float32)
    g = m @ m.T
    free = np.ones(GROUPS_PER_TILE, dtype=bool)
    block = np.zeros(GROUPS_PER_TILE, dtype=np.int64)
    for b in range(BLOCKS_PER_TILE):
        idx = np.flatnonzero(free)
        seed = idx[np.argmin(g[idx][:, idx].sum(1))]
        free[seed] = False
        block[seed] = b
        score = g[seed].copy()
        for _ in range(BLOCK_GROUPS - 1):
            cand = np.flatnonzero(free)
            nxt = cand[np.argmax(score[cand])]
            free[nxt] = False
            block[nxt] = b
            score += g[nxt]
    return block


def _deal_classes(sp):
    """SIMD class (= position inside its block) of every group of ONE tile: ``sp[b, j, ph]`` =
    sparse super-steps of group j of block b in phase ph.  The dense part of a block costs its 4
    waves the same, so only the sparse part needs balancing: heaviest block first, each taking the
    permutation that keeps (max class load in A) + (max class load in B) smallest."""
    load = np.zeros((BLOCK_GROUPS, 2), dtype=np.int64)
    cls = np.zeros((BLOCKS_PER_TILE, BLOCK_GROUPS), dtype=np.int64)
    for b in np.argsort(-sp.sum((1, 2)), kind="stable"):
        # cand[p, c, ph] = load of class c if group j goes to class _PERMS[p, j]
        cand = np.repeat(load[None], len(_PERMS), 0)
        for j in range(BLOCK_GROUPS):
            cand[np.arange(len(_PERMS)), _PERMS[:, j]] += sp[b, j]
        cost = cand.max(1).sum(1) * 4096 + cand.sum(2).max(1)
        p = int(np.argmin(cost))
        cls[b] = _PERMS[p]
        load = cand[p]
    return cls


def build_mix_stream(rowptr, col, val, trow, slot_of_row, thr=4, dh=10, pad_dense=False):
    """Mixed plan for tiles ``trow`` whose rows already sit in group order (``slot_of_row``: position
    of every row inside its tile; rows 4 g .. 4 g + 3 form group g).  Returns a dict of numpy arrays
    (see the module docstring) plus statistics.  ``pad_dense`` (the all-dense kernel, ``sgp_spmm_dense_f32``,
    with ``thr=1``): EVERY candidate column stays dense and a (block, phase) list is padded to whole
    instructions with weight-0 columns (staged row 0) instead of demoting what does not fill one."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    n_tiles = len(trow) - 1
    n_rows = int(trow[-1])
    uptr, ucol, lcol, row_of_edge = tile_unions(rowptr, col, trow)
    tile_of_row = np.repeat(np.arange(n_tiles, dtype=np.int64), np.diff(trow))
    slot_of_row = np.asarray(slot_of_row, dtype=np.int64)
    n_groups = n_tiles * GROUPS_PER_TILE
    group_of_row = tile_of_row * GROUPS_PER_TILE + slot_of_row // GROUP_ROWS
    g_e = group_of_row[row_of_edge]
    uniq, inv = np.unique(g_e * 65536 + lcol, return_inverse=True)        # (group, column) entries
    g_s, lc = uniq >> 16, uniq & 0xffff
    t_s = g_s // GROUPS_PER_TILE
    U = np.diff(uptr)
    first_entry = np.searchsorted(t_s, np.arange(n_tiles + 1))

    # ---- blocks of 4 groups (per tile)
    block_of_group = np.zeros(n_groups, dtype=np.int64)
    for k in range(n_tiles):
        e0, e1 = first_entry[k], first_entry[k + 1]
        member = np.zeros((GROUPS_PER_TILE, max(int(U[k]), 1)), dtype=bool)
        member[g_s[e0:e1] - k * GROUPS_PER_TILE, lc[e0:e1]] = True
        block_of_group[k * GROUPS_PER_TILE:(k + 1) * GROUPS_PER_TILE] = _cluster_blocks(member)
    gblock_of_group = np.arange(n_groups) // GROUPS_PER_TILE * BLOCKS_PER_TILE + block_of_group

    # ---- dense columns of every (block, phase)
    b_s = gblock_of_group[g_s]
    uniq2, inv2, cnt2 = np.unique(b_s * 65536 + lc, return_inverse=True, return_counts=True)
    blk2, lc2 = uniq2 >> 16, uniq2 & 0xffff
    bp2 = blk2 * 2 + (lc2 & 1)
    cand = cnt2 >= thr
    n_bp = n_tiles * BLOCKS_PER_TILE * 2
    n_cand = np.bincount(bp2[cand], minlength=n_bp)
    n_keep = n_cand.copy() if pad_dense else np.minimum(n_cand // 4, dh) * 4
    order = np.lexsort((lc2, -cnt2, bp2, ~cand))                          # candidates first, by (bp, most shared, column)
    order = order[:int(cand.sum())]
    start = np.zeros(n_bp + 1, dtype=np.int64)
    start[1:] = np.cumsum(n_cand)
    rank = np.arange(order.size) - start[bp2[order]]
    dense2 = np.zeros(uniq2.size, dtype=bool)
    dense2[order[rank < n_keep[bp2[order]]]] = True
    dense_s = dense2[inv2]                                                # per (group, column) entry
    dense_e = dense_s[inv]                                                # per edge

    # ---- SIMD classes: balance the sparse super-steps
    ph_s = lc & 1
    ns = np.bincount((g_s * 2 + ph_s)[~dense_s], minlength=2 * n_groups).reshape(n_groups, 2)
    ss = (ns + 3) // 4
    pos_in_block = np.zeros(n_groups, dtype=np.int64)
    for k in range(n_tiles):
        gs = np.arange(k * GROUPS_PER_TILE, (k + 1) * GROUPS_PER_TILE)
        members = np.stack([gs[block_of_group[gs] == b] for b in range(BLOCKS_PER_TILE)])   # [4, 4]
        cls = _deal_classes(ss[members])
        pos_in_block[members.reshape(-1)] = cls.reshape(-1)
    new_group = np.arange(n_groups) // GROUPS_PER_TILE * GROUPS_PER_TILE + block_of_group * BLOCK_GROUPS + pos_in_block
    new_slot = (new_group[group_of_row] % GROUPS_PER_TILE) * GROUP_ROWS + slot_of_row % GROUP_ROWS

    # ---- sparse part: the two-phase row-group stream of the remaining edges
    keep = ~dense_e
    ps = build_phase_stream(trow, uptr, ucol, lcol[keep], row_of_edge[keep], val[keep], new_slot,
                            rebalance=False, mode="parity")
    usplit = ps["usplit"].astype(np.int64)

    # ---- dense part
    t2 = blk2 // BLOCKS_PER_TILE
    stage2 = np.where((lc2 & 1) == 0, lc2 >> 1, usplit[t2] + (lc2 >> 1))
    d_ids = np.flatnonzero(dense2)
    d_ids = d_ids[np.lexsort((stage2[d_ids], bp2[d_ids]))]                # by (block, phase), then staged slot
    # (the block index inside a tile must follow the NEW slots: block b of the tile = slots 16 b ..)
    n_inst = (n_keep + 3) // 4
    dptr = np.zeros(n_bp + 1, dtype=np.int64)
    dptr[1:] = np.cumsum(n_inst)
    dstart = np.zeros(n_bp + 1, dtype=np.int64)
    dstart[1:] = np.cumsum(n_keep)
    p = np.arange(d_ids.size) - dstart[bp2[d_ids]]
    inst_of = np.full(uniq2.size, -1, dtype=np.int64)
    k_of = np.zeros(uniq2.size, dtype=np.int64)
    inst_of[d_ids] = dptr[bp2[d_ids]] + p // 4
    k_of[d_ids] = p % 4
    n_dense = int(dptr[-1])
    didx = np.zeros((max(n_dense, 1), 4), dtype=np.int32)
    didx[inst_of[d_ids], k_of[d_ids]] = (stage2[d_ids] * 256).astype(np.int32)
    dw = np.zeros((max(n_dense, 1), 64), dtype=np.float32)
    e2 = inv2[inv]                                                        # (block, column) entry of every edge
    de = np.flatnonzero(dense_e)
    row_in_block = new_slot[row_of_edge[de]] % (GROUP_ROWS * BLOCK_GROUPS)
    dw[inst_of[e2[de]], 16 * k_of[e2[de]] + row_in_block] = val[de]

    # ---- statistics: matrix-pipe work per step in units of 32 cycles (one super-step = one 16x16x4)
    gsup = ps["gsup"].astype(np.int64).reshape(n_tiles, BLOCKS_PER_TILE, BLOCK_GROUPS, 2)
    dn = n_inst.reshape(n_tiles, BLOCKS_PER_TILE, 2)
    cls_sparse = gsup.sum(1)                                              # [tile, class, phase]
    cls_dense = dn.sum(1)[:, None, :]                                     # every class carries each block's dense part
    phase_cost = (cls_sparse + cls_dense).max(1).sum(1)
    pairs_total = uniq.size
    out = dict(ps)
    out.update(dptr=dptr.astype(np.int32), didx=didx, dw=dw, n_dense=n_dense,
               max_dense=int(n_inst.max(initial=0)), dense_share=float(dense_s.sum()) / max(1, pairs_total),
               phase_cost=phase_cost, sparse_steps=cls_sparse, dense_steps=cls_dense, thr=thr, dh=dh)
    return out


def mix_reference(mix, n_tiles, x):
    """The product evaluated from the plan arrays the way the kernel does it (numpy, float64):
    ``x[num_cols, F]`` -> ``y[n_rows, F]``.  Test helper: it takes no shortcut through the CSR."""
    x = np.asarray(x, dtype=np.float64)
    uptr, ucol, rowmap = mix["uptr"], mix["ucol"], mix["rowmap"]
    gptr, gidx, gw = mix["gptr"], mix["gidx"], mix["gw"]
    dptr, didx, dw = mix["dptr"], mix["didx"], mix["dw"]
    n_rows = int(rowmap.max()) + 1 if rowmap.size else 0
    y = np.zeros((n_rows, x.shape[1]))
    written = np.zeros(n_rows, dtype=np.int64)
    for t in range(n_tiles):
        stage = x[ucol[uptr[t]:uptr[t + 1]]]
        acc = np.zeros((TILE_SLOTS, x.shape[1]))
        for g in range(GROUPS_PER_TILE):
            for qd in range(gptr[(t * GROUPS_PER_TILE + g) * 2], gptr[(t * GROUPS_PER_TILE + g) * 2 + 2]):
                for cls in range(4):
                    for sup in range(4):
                        w = gw[qd, cls, sup].astype(np.float64)            # [4 rows]
                        if (w != 0).any():
                            acc[4 * g:4 * g + 4] += w[:, None] * stage[gidx[qd, cls, sup] // 256][None, :]
        for b in range(BLOCKS_PER_TILE):
            for m in range(dptr[(t * BLOCKS_PER_TILE + b) * 2], dptr[(t * BLOCKS_PER_TILE + b) * 2 + 2]):
                a = dw[m].astype(np.float64).reshape(4, 16)               # [k, row]
                acc[16 * b:16 * b + 16] += a.T @ stage[didx[m] // 256]
        rows = rowmap[t * TILE_SLOTS:(t + 1) * TILE_SLOTS]
        ok = rows >= 0
        y[rows[ok]] = acc[ok]
        written[rows[ok]] += 1
    assert (written == 1).all(), "every row must be written exactly once"
    return y


class MixPlan:
    """Device-ready arrays of ``sgp_spmm_mix_f32`` (torch tensors) + plan statistics."""
    TENSORS = ("uptr", "ucol", "usplit", "gptr", "gsup", "gidx", "gw", "rowmap", "dptr", "didx", "dw")

    def __init__(self, arrays, n_tiles, n_rows, reordered=False):
        self.n_tiles, self.n_rows, self.reordered = int(n_tiles), int(n_rows), reordered
        for k in self.TENSORS:
            v = arrays[k]
            setattr(self, k, v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v)))
        for k in ("max_union", "max_tile_quads", "max_range_steps", "max_dense", "n_dense", "dense_share",
                  "fill", "thr", "dh"):
            setattr(self, k, arrays[k])
        pc = arrays["phase_cost"]
        self.mean_phase_cost = float(np.mean(pc)) if len(pc) else 0.0

    def arrays(self):
        return {k: getattr(self, k).cpu().numpy() for k in self.TENSORS}

    def to(self, device):
        d = {k: getattr(self, k).to(device) for k in self.TENSORS}
        for k in ("max_union", "max_tile_quads", "max_range_steps", "max_dense", "n_dense", "dense_share",
                  "fill", "thr", "dh"):
            d[k] = getattr(self, k)
        d["phase_cost"] = [self.mean_phase_cost]
        return MixPlan(d, self.n_tiles, self.n_rows, self.reordered)


def build_mix_plan(rowptr, col, val, n_rows, base, thr=4, dh=10, order=None, pad_dense=False):
    """Mixed plan on the tiles and row groups of ``base`` (a ``graph.TilePlan`` with a two-phase
    stream).  ``order`` (new id k = old id ``order[k]``): ``base`` was built on the renumbered
    operator (``graph.build_reordered_plan``); the plan then addresses ORIGINAL ids through ``ucol``
    / ``rowmap`` like the base plan does."""
    if base is None or base.pipe is None:
        return None
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col)
    val = np.asarray(val, dtype=np.float32)
    trow = base.trow.cpu().numpy().astype(np.int64)
    rowmap = base.pipe["rowmap"].cpu().numpy().astype(np.int64)
    if order is not None:
        import scipy.sparse as sp
        order = np.asarray(order, dtype=np.int64)
        pos = np.empty(n_rows, dtype=np.int64)
        pos[order] = np.arange(n_rows)
        rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
        a = sp.csr_matrix((val, (pos[rows], pos[col.astype(np.int64)])), shape=(n_rows, n_rows))
        a.sort_indices()
        rowptr, col, val = a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.float32)
        ok = rowmap >= 0
        rowmap = rowmap.copy()
        rowmap[ok] = pos[rowmap[ok]]                                     # back to the renumbered ids
    slot_of_row = np.zeros(n_rows, dtype=np.int64)
    ok = np.flatnonzero(rowmap >= 0)
    slot_of_row[rowmap[ok]] = ok % TILE_SLOTS
    arrays = build_mix_stream(rowptr, col, val, trow, slot_of_row, thr=thr, dh=dh, pad_dense=pad_dense)
    if order is not None:
        o32 = order.astype(np.int32)
        arrays["ucol"] = o32[arrays["ucol"].astype(np.int64)]
        rm = arrays["rowmap"].astype(np.int64)
        arrays["rowmap"] = np.where(rm >= 0, o32[np.maximum(rm, 0)], -1).astype(np.int32)
    return MixPlan(arrays, len(trow) - 1, n_rows, reordered=order is not None)
