"""Host plan of ``sgp_spmm_colblock_f32`` (include/sgp_amd.h): the column-blocked hop for graphs
without locality behind ``x = adj @ x`` (reference: lib/sgp_preprocessing.py:200-203).

The columns are cut into ``n_blocks`` blocks of ``cols_per_block`` consecutive columns whose source
rows (``feat * 4`` bytes each) fit the L2 of an XCD; the rows are cut into ``n_wg`` contiguous ranges
of (nearly) equal edge count and at most ``rows_cap`` rows, one persistent workgroup each.  Local row
r of a workgroup belongs to SLOT r % 64 (one of the 64 (wave, lane group) pairs of the workgroup);
for every (workgroup, block) segment each slot gets the list of its rows' edges in that block, row by
row, as 8-byte entries

    entry.x = column | local row << 22 | last edge of the row's run << 31     entry.y = weight (fp32 bits)

and the 64 lists are padded to ONE length, a multiple of ``round_pad``, with entries (first column of the
block, local row 511 = a row nobody owns, weight 0; the last one of a list carries the flag).  The array ends with ``2 * round_pad``
spare rounds of such entries (the kernel's look-ahead reads them).  Storage is round-major: entry i of slot s of a segment that starts at
round q sits at ``(q + i) * 64 + s``; ``segptr[wg * n_blocks + b]`` = q.
"""
import numpy as np
import torch


class ColBlockPlan:
    def __init__(self, entries, segptr, wg_row0, n_wg, n_blocks, cols_per_block, n_rows, n_cols, pad_share):
        self.entries, self.segptr, self.wg_row0 = entries, segptr, wg_row0
        self.n_wg, self.n_blocks, self.cols_per_block = n_wg, n_blocks, cols_per_block
        self.n_rows, self.n_cols, self.pad_share = n_rows, n_cols, pad_share

    def to(self, device):
        return ColBlockPlan(self.entries.to(device), self.segptr.to(device), self.wg_row0.to(device), self.n_wg,
                            self.n_blocks, self.cols_per_block, self.n_rows, self.n_cols, self.pad_share)


def row_ranges(rowptr, n_rows, n_wg, rows_cap):
    """Contiguous row ranges of (nearly) equal edge count, none longer than ``rows_cap``."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    nnz = int(rowptr[n_rows])
    n_wg = max(1, min(n_wg, n_rows))
    targets = (np.arange(1, n_wg, dtype=np.int64) * nnz) // n_wg
    cuts = np.searchsorted(rowptr[1:n_rows + 1], targets, side="left") + 1 if nnz else \
        (np.arange(1, n_wg, dtype=np.int64) * n_rows) // n_wg
    bounds = np.unique(np.concatenate(([0], np.clip(cuts, 0, n_rows), [n_rows])))
    out = [0]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        k = -(-(hi - lo) // rows_cap)                      # split ranges that exceed the LDS budget
        for j in range(1, k + 1):
            out.append(lo + (hi - lo) * j // k)
    return np.asarray(out, dtype=np.int64)


PAD_ROW = 511
MAX_COLS = 1 << 22           # a plan entry packs the column in 22 bits (the kernel checks the same limit)


def deal_rows_to_slots(wg, lrow, blk, bounds, n_blocks):
    """Slot (0..63) of every row.  The 64 lists of a (workgroup, block) segment are padded to the longest
    one, so rows are dealt to keep the per-block loads of a workgroup's slots level: every workgroup
    takes its rows by decreasing edge count, each row going to the slot whose worst block load grows
    least (ties: lightest slot).  ``r % 64`` leaves the longest list 27 % above the mean on 100 random
    columns per row; this deal ~10 %.  All workgroups are processed side by side (one numpy step per
    rank)."""
    n_wg = len(bounds) - 1
    n_rows = int(bounds[-1])
    rows_of = np.diff(bounds)
    max_rows = int(rows_of.max()) if n_wg else 0
    cnt = np.zeros((n_rows, n_blocks), dtype=np.int64)
    np.add.at(cnt, (bounds[wg] + lrow, blk), 1)
    tot = cnt.sum(1)
    # rank rows inside their workgroup by decreasing edge count
    wg_of_row = np.repeat(np.arange(n_wg, dtype=np.int64), rows_of)
    order = np.lexsort((-tot, wg_of_row))
    rank_rows = np.full((n_wg, max_rows), -1, dtype=np.int64)
    rank_rows[wg_of_row[order], np.arange(n_rows) - bounds[wg_of_row[order]]] = order
    load = np.zeros((n_wg, 64, n_blocks), dtype=np.int64)
    slot_of = np.zeros(n_rows, dtype=np.int64)
    ar = np.arange(n_wg)
    for k in range(max_rows):
        r = rank_rows[:, k]
        ok = r >= 0
        c = np.where(ok[:, None], cnt[np.maximum(r, 0)], 0)            # [n_wg, n_blocks]
        worst = (load + c[:, None, :]).max(2)                           # [n_wg, 64]
        cost = worst * (1 << 20) + load.sum(2)
        best = cost.argmin(1)
        load[ar, best] += c
        slot_of[r[ok]] = best[ok]
    return slot_of


def build_colblock_plan(rowptr, col, val, n_rows, n_cols, feat, rows_cap=511, round_pad=4, n_wg=256,
                        l2_bytes=2.5 * 2 ** 20):
    """Plan for an ``[n_rows, n_cols]`` CSR matrix and feature width ``feat`` (a multiple of 64)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    if n_cols >= MAX_COLS:
        return None
    cpb = int(max(256, l2_bytes // (feat * 4)))
    n_blocks = max(1, -(-n_cols // cpb))
    bounds = row_ranges(rowptr, n_rows, n_wg, min(rows_cap, PAD_ROW))
    n_wg = len(bounds) - 1
    nnz = len(col)
    row = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr[:n_rows + 1]))
    wg = np.searchsorted(bounds, row, side="right") - 1
    lrow = row - bounds[wg]
    n_seg = n_wg * n_blocks
    seg = wg * n_blocks + col // cpb
    slot_of = deal_rows_to_slots(wg, lrow, col // cpb, bounds, n_blocks)
    key = seg * 64 + slot_of[row]                            # (segment, slot)
    order = np.lexsort((lrow, key))                          # a slot's rows in order inside a segment
    key_s, lrow_s = key[order], lrow[order]
    counts = np.bincount(key_s, minlength=n_seg * 64).astype(np.int64)
    rounds = counts.reshape(n_seg, 64).max(1)
    rounds = -(-rounds // round_pad) * round_pad             # one list length per segment
    start = np.concatenate(([0], np.cumsum(rounds)))         # in rounds
    total = (int(start[-1]) + 2 * round_pad) * 64            # + spare rounds for the kernel's look-ahead
    first = np.concatenate(([0], np.cumsum(counts)))[:-1]
    rank = np.arange(nnz, dtype=np.int64) - first[key_s]     # position inside the slot's list
    pos = (start[key_s // 64] + rank) * 64 + key_s % 64
    last = np.ones(nnz, dtype=bool)                          # last edge of its (slot, row) run
    if nnz > 1:
        last[:-1] = (key_s[1:] != key_s[:-1]) | (lrow_s[1:] != lrow_s[:-1])
    rows_cap = min(rows_cap, PAD_ROW)
    seg_of_round = np.concatenate((np.repeat(np.arange(n_seg, dtype=np.int64), rounds),
                                   np.zeros(2 * round_pad, dtype=np.int64)))
    ex = np.repeat(np.minimum((seg_of_round % n_blocks) * cpb, max(n_cols - 1, 0)), 64).astype(np.int64)
    ex |= PAD_ROW << 22                                      # padding sums into the spare row ...
    cnt2 = counts.reshape(n_seg, 64)
    sg, sl = np.nonzero(cnt2 < rounds[:, None])              # ... and its run ends with the list
    ex[(start[sg] + rounds[sg] - 1) * 64 + sl] |= 1 << 31
    ex[int(start[-1]) * 64:] |= 1 << 31
    ew = np.zeros(total, dtype=np.float32)
    ex[pos] = col[order] | (lrow_s << 22) | (last.astype(np.int64) << 31)
    ew[pos] = val[order]
    entries = np.empty((total, 2), dtype=np.int32)
    entries[:, 0] = ex.astype(np.uint32).view(np.int32)
    entries[:, 1] = ew.view(np.int32)
    return ColBlockPlan(torch.from_numpy(entries), torch.from_numpy(start.astype(np.int32)),
                        torch.from_numpy(bounds.astype(np.int32)), n_wg, n_blocks, cpb, n_rows, n_cols,
                        float(total - nnz) / max(total, 1))


def plan_to_dense(plan):
    """The matrix a plan encodes, formed the way the kernel forms it: every slot walks its list, sums a
    row's run and adds the sum to the row when the flag says so (tests)."""
    e = plan.entries.numpy()
    ex = e[:, 0].view(np.uint32).astype(np.int64).reshape(-1, 64)
    w = e[:, 1].view(np.float32).reshape(-1, 64)
    seg = plan.segptr.numpy().astype(np.int64)
    bounds = plan.wg_row0.numpy().astype(np.int64)
    a = np.zeros((plan.n_rows, plan.n_cols), dtype=np.float64)
    for s in range(plan.n_wg * plan.n_blocks):
        g = s // plan.n_blocks
        for slot in range(64):
            run = {}
            for i in range(seg[s], seg[s + 1]):
                x = ex[i, slot]
                c = x & 0x3fffff
                run[c] = run.get(c, 0.0) + float(w[i, slot])
                if x >> 31:
                    if ((x >> 22) & 511) != PAD_ROW:
                        r = bounds[g] + ((x >> 22) & 511)
                        for cc, v in run.items():
                            a[r, cc] += v
                    else:
                        assert all(v == 0.0 for v in run.values()), "a padding entry drops a non-zero sum"
                    run = {}
            assert all(v == 0.0 for v in run.values()), "edges after the last flush of a list"
    return a
