"""Host plan of ``sgp_spmm_split_f32`` (include/sgp_amd.h; kernel ``csrc/spmm_split.hip``): the split-fp16
hop behind ``x = adj @ x`` (reference: lib/sgp_preprocessing.py:200-203).

Rows are dealt in order to WAVES of at most ``rows_per_wave`` (16; 32 in the 8-wave build) rows whose entries touch at most ``32 * chunks`` distinct
columns, waves to TILES of at most ``waves`` waves whose rows touch at most ``max_union`` distinct columns
(the tile's staged rows).  Per tile the kernel reads

* ``hdr[tile]``            64 ints: rows of every wave at [waves : 2 waves], staged rows U at [2 waves]
* ``rowid[tile, w, slot]``  result row of slot ``16 half + m`` of wave w, -1 = empty slot (rows are dealt in the
                           given numbering or, for numberings without locality, in ``order``)
* ``ucol[tile, s]``        column (= source row) staged at position s, -1 beyond U
* ``afr[tile, w, c, q]``   A fragments of ``v_mfma_f32_16x16x32_f16`` in lane order, q = piece:
                           lane ``m + 16 g``, element e = piece of ``a[row slot m, column k = 8 g + e of
                           chunk c] * 2^e_row`` (piece 0: the value truncated to fp16, piece 1: the rounded rest);
                           every row has its own power-of-two scale (its largest entry lands in [2^13, 2^14]), so the
                           kernel is as indifferent to the scale of a row of A as fp32 is
* ``rinv[tile, w, slot]``  ``2^-e_row`` of the slot's row (0 for empty slots): the kernel multiplies the result by it
* ``adr[tile, w, c]``      per-lane byte addresses (inside a staged unit: a group of 8 staged rows is 512 bytes, the hi
                           pieces of row r at ``32 r``, the lo pieces at ``256 + 32 r``) of the two transpose reads that
                           fetch a chunk's B operand, packed ``a0 | a1 << 16``: read j of lane ``i + 16 g`` points at the
                           staged row of column ``k = 8 g + 4 j + i / 4``, bytes ``8 (i % 4)``

Columns a wave does not use up to ``32 * chunks`` are padded with weight 0 and the address of staged row 0
(finite data, so 0 * x stays 0).  Duplicate entries of a row are summed before the split.
"""
import numpy as np
import torch


class SplitPlan:
    def __init__(self, hdr, rowid, ucol, afr, adr, rinv, n_tiles, n_rows, n_cols, norm_inf, stats):
        self.hdr, self.rowid, self.ucol, self.afr, self.adr, self.rinv = hdr, rowid, ucol, afr, adr, rinv
        self.n_tiles, self.n_rows, self.n_cols = n_tiles, n_rows, n_cols
        self.norm_inf, self.stats = norm_inf, stats
        self.accumulate = False          # later passes of an operator whose long rows were cut into column segments

    def to(self, device):
        p = SplitPlan(self.hdr.to(device), self.rowid.to(device), self.ucol.to(device), self.afr.to(device),
                      self.adr.to(device), self.rinv.to(device), self.n_tiles, self.n_rows, self.n_cols, self.norm_inf,
                      self.stats)
        p.accumulate = self.accumulate
        return p


def deal_rows(rowptr, col, n_rows, n_cols, waves, chunks, max_union, rows_per_wave=32, order=None):
    """Greedy deal in row order (``order``: the sequence in which rows are taken).  Returns (wave_of_row,
    slot_of_row, tile_of_wave, rows) or None when a single row exceeds a wave's column budget."""
    cap = 32 * chunks
    wmark = np.full(n_cols, -1, dtype=np.int64)
    tmark = np.full(n_cols, -1, dtype=np.int64)
    wave_of_row = np.full(n_rows, -1, dtype=np.int64)     # (-1: a row outside ``order`` -- a later pass of a long-row operator)
    slot_of_row = np.full(n_rows, -1, dtype=np.int64)
    tile_of_wave, rows = [], []
    wave, tile = -1, -1
    w_rows = w_cols = t_cols = t_waves = 0

    def open_wave(r, new_tile):
        nonlocal wave, tile, w_rows, w_cols, t_cols, t_waves
        wave += 1
        if new_tile:
            tile += 1
            t_cols = t_waves = 0
        t_waves += 1
        w_rows = w_cols = 0
        tile_of_wave.append(tile)
        rows.append(0)

    for r in (range(n_rows) if order is None else order):
        r = int(r)
        c = np.unique(col[rowptr[r]:rowptr[r + 1]])
        if c.size > cap or c.size > max_union:
            return None
        if wave < 0:
            open_wave(r, True)
        new_w = int((wmark[c] != wave).sum())
        new_t = int((tmark[c] != tile).sum())
        need_wave = w_rows == rows_per_wave or w_cols + new_w > cap
        if t_cols + new_t > max_union or (need_wave and t_waves == waves):
            open_wave(r, True)
            new_w = new_t = c.size
        elif need_wave:
            open_wave(r, False)
            new_w = c.size
        wmark[c] = wave
        tmark[c] = tile
        w_cols += new_w
        t_cols += new_t
        wave_of_row[r] = wave
        slot_of_row[r] = w_rows
        w_rows += 1
        rows[wave] = w_rows
    return (wave_of_row, slot_of_row, np.asarray(tile_of_wave, dtype=np.int64), np.asarray(rows, dtype=np.int64))


def split_fp16(v):
    """v (float32, already scaled) -> (hi, lo) fp16 pieces with hi + lo ~ v to 2^-22: hi truncated towards
    zero (what ``v_cvt_pkrtz_f16_f32`` does on the device side of x), lo the rounded remainder."""
    v = np.asarray(v, dtype=np.float32)
    bits = v.view(np.uint32) & np.uint32(0xFFFFE000)        # keep 10 explicit mantissa bits
    hi32 = bits.view(np.float32)
    small = np.abs(v) < np.float32(2.0 ** -14)              # fp16 subnormal range: let the conversion round
    hi32 = np.where(small, v.astype(np.float16).astype(np.float32), hi32)
    hi = hi32.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


# k-slots of the four 8-row sets a chunk's two transpose reads serve together (lanes 0-31 / 32-63 of each read)
_SET_SLOTS = np.array([[0, 1, 2, 3, 8, 9, 10, 11], [4, 5, 6, 7, 12, 13, 14, 15],
                       [16, 17, 18, 19, 24, 25, 26, 27], [20, 21, 22, 23, 28, 29, 30, 31]])


def _bank_aware_slots(w_of_key, pos_of_key, stage, n_chunks_total, chunks):
    """New position (chunk * 32 + k-slot) of every (wave, column) key: same chunk, k-slot chosen so that the rows
    one LDS cycle reads together lie in different banks where the chunk's rows allow it."""
    chunk = pos_of_key // 32
    wc = w_of_key * chunks + chunk
    res = stage & 7
    idx = np.arange(wc.size)

    def rank_within(group, *minor):
        order = np.lexsort(tuple(reversed(minor)) + (group,))
        g = group[order]
        first = np.r_[True, g[1:] != g[:-1]]
        start = np.maximum.accumulate(np.where(first, idx, 0))
        out = np.empty_like(idx)
        out[order] = idx - start
        return out

    rank = rank_within(wc * 8 + res, stage)              # j-th row of its residue in its chunk
    primary = rank < 4
    slot = np.full(wc.size, -1, dtype=np.int64)
    j = rank_within(np.where(primary, wc * 4 + rank, -1), res)       # index inside its set (at most one row per residue)
    slot[primary] = _SET_SLOTS[rank[primary], j[primary]]
    occ = np.zeros((n_chunks_total, 32), dtype=bool)
    occ[wc[primary], slot[primary]] = True
    extra = ~primary
    if extra.any():
        free = np.argsort(occ, axis=1, kind="stable")    # free k-slots of every chunk first, in order
        e = rank_within(np.where(extra, wc, -1), res, stage)
        slot[extra] = free[wc[extra], e[extra]]
    return chunk * 32 + slot


def build_split_plan(rowptr, col, val, n_rows, n_cols, waves=16, chunks=7, max_union=768, order=None, rows_per_wave=16,
                     threads=0):
    """``order``: optional sequence of rows (a locality order of the graph, or the subset of rows a later pass of a
    long-row operator touches): rows are dealt to waves in that sequence -- rows outside it get no slot -- while the plan
    keeps addressing rows and columns by their ORIGINAL ids, so no tensor is ever permuted.

    The work is done by the library's host-side planner (``csrc/plan_split.hip``: ``sgp_split_plan_deal`` /
    ``sgp_split_plan_fill``, all cores; no GPU needed); ``build_split_plan_numpy`` below is the same algorithm in numpy,
    kept as its cross-check (tests/test_splitplan.py holds the two to the same bytes) -- 2 s instead of 23 s on the
    target graph."""
    import ctypes
    from . import hip
    lib = hip.load()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.float32)
    if n_rows == 0 or col.size == 0 or not np.isfinite(val).all():
        return None
    assert rows_per_wave == 16 and rowptr.size >= n_rows + 1
    if col.min() < 0 or col.max() >= n_cols:
        raise ValueError("column index out of range")
    seq = None if order is None else np.ascontiguousarray(order, dtype=np.int64)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    wave_of_row, slot_of_row = np.empty(n_rows, dtype=np.int64), np.empty(n_rows, dtype=np.int64)
    tile_of_wave, rows = np.empty(n_rows, dtype=np.int64), np.empty(n_rows, dtype=np.int64)
    n_waves = lib.sgp_split_plan_deal(ptr(rowptr), ptr(col), n_rows, n_cols, None if seq is None else ptr(seq),
                                      0 if seq is None else seq.size, waves, chunks, max_union, rows_per_wave,
                                      ptr(wave_of_row), ptr(slot_of_row), ptr(tile_of_wave), ptr(rows))
    if n_waves == -2 or n_waves == 0:
        return None                                   # a row beyond a wave's column budget / nothing dealt
    if n_waves < 0:
        raise RuntimeError("sgp_split_plan_deal: " + lib.sgp_last_error().decode())
    tile_of_wave, rows = tile_of_wave[:n_waves], rows[:n_waves]
    n_tiles = int(tile_of_wave[-1]) + 1
    hdr = torch.empty((n_tiles, 64), dtype=torch.int32)
    rowid = torch.empty((n_tiles, waves, rows_per_wave), dtype=torch.int32)
    ucol = torch.empty((n_tiles, max_union), dtype=torch.int32)
    afr = torch.empty((n_tiles, waves, chunks, 2, 64, 8), dtype=torch.float16)
    adr = torch.empty((n_tiles, waves, chunks, 64), dtype=torch.int32)
    rinv = torch.empty((n_tiles, waves, 16), dtype=torch.float32)
    st = np.zeros(8, dtype=np.float64)
    hip._check(lib.sgp_split_plan_fill(ptr(rowptr), ptr(col), ptr(val), n_rows, n_cols, ptr(wave_of_row), ptr(slot_of_row),
                                       ptr(tile_of_wave), ptr(rows), n_waves, n_tiles, waves, chunks, max_union,
                                       hdr.data_ptr(), rowid.data_ptr(), ucol.data_ptr(), afr.data_ptr(), adr.data_ptr(),
                                       rinv.data_ptr(), ptr(st), int(threads)), "sgp_split_plan_fill")
    stats = dict(tiles=n_tiles, waves=int(n_waves), rows_per_wave=float(st[2]), rows_per_tile=float(st[3]),
                 staged_per_row=float(st[4]), chunk_fill=float(st[5]), max_union=int(st[6]))
    return SplitPlan(hdr, rowid, ucol, afr, adr, rinv, n_tiles, n_rows, n_cols, float(st[7]), stats)


def build_split_plan_numpy(rowptr, col, val, n_rows, n_cols, waves=16, chunks=7, max_union=768, order=None, rows_per_wave=16):
    """The planner in numpy: the reference the native planner is tested against (same arrays, byte for byte)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    if n_rows == 0 or col.size == 0 or not np.isfinite(val).all():
        return None
    assert rows_per_wave == 16
    deal = deal_rows(rowptr, col, n_rows, n_cols, waves, chunks, max_union, rows_per_wave=rows_per_wave, order=order)
    if deal is None:
        return None
    wave_of_row, slot_of_row, tile_of_wave, rows = deal
    n_waves = tile_of_wave.size
    n_tiles = int(tile_of_wave[-1]) + 1
    first_wave_of_tile = np.searchsorted(tile_of_wave, np.arange(n_tiles))
    w_in_tile = np.arange(n_waves) - first_wave_of_tile[tile_of_wave]

    row_of_edge = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr[:n_rows + 1]))
    rowsum = np.zeros(n_rows, dtype=np.float64)
    np.add.at(rowsum, row_of_edge, np.abs(val.astype(np.float64)))
    dealt = wave_of_row[row_of_edge] >= 0                # entries of rows outside ``order`` belong to no wave
    row_of_edge, col, val = row_of_edge[dealt], col[dealt], val[dealt]
    e_wave = wave_of_row[row_of_edge]
    e_tile = tile_of_wave[e_wave]
    # staged position of every (tile, column)
    tkey, tinv = np.unique(e_tile * n_cols + col, return_inverse=True)
    t_of_key = tkey // n_cols
    t_first = np.searchsorted(t_of_key, np.arange(n_tiles))
    union = np.diff(np.append(t_first, tkey.size))
    stage_of_key = np.arange(tkey.size) - t_first[t_of_key]
    ucol = np.full((n_tiles, max_union), -1, dtype=np.int32)
    ucol[t_of_key, stage_of_key] = (tkey % n_cols).astype(np.int32)
    empty = union == 0                                    # a tile of empty rows still stages one (finite) row: its padding reads
    ucol[empty, 0] = 0
    union = np.where(empty, 1, union)
    # position of every (wave, column) in the wave's column list
    wkey, winv = np.unique(e_wave * n_cols + col, return_inverse=True)
    w_of_key = wkey // n_cols
    w_first = np.searchsorted(w_of_key, np.arange(n_waves))
    pos_of_key = np.arange(wkey.size) - w_first[w_of_key]
    assert int(pos_of_key.max()) < 32 * chunks and int(union.max()) <= max_union
    stage_of_wkey = stage_of_key[np.searchsorted(tkey, tile_of_wave[w_of_key] * n_cols + wkey % n_cols)]

    # k-slots inside a chunk: a transpose read serves lanes 0-31 and 32-63 in one LDS cycle each, i.e. the 8 staged rows
    # behind k = {0-3, 8-11}, {16-19, 24-27} (first read of a chunk) and {4-7, 12-15}, {20-23, 28-31} (second) together;
    # a staged row s covers the 8 banks (s & 7) * 8 .., so a set of 8 rows is conflict-free when their s & 7 differ.
    # Columns keep their chunk but are dealt to the four sets by residue (the j-th row of a residue goes to set j),
    # rows beyond four of a residue take what is left; measured on the target graph: 1.83 -> 1.1 LDS cycles per lane group.
    pos_of_key = _bank_aware_slots(w_of_key, pos_of_key, stage_of_wkey, n_waves * chunks, chunks)

    # addresses: staged row of (wave, chunk, k); padding slots repeat a row of their own set (same address = broadcast)
    srow = np.full((n_waves, chunks * 32), -1, dtype=np.int64)
    srow[w_of_key, pos_of_key] = stage_of_wkey
    sets = srow.reshape(n_waves * chunks, 32)[:, _SET_SLOTS]                # [chunk, set, 8]
    rep = np.where(sets >= 0, sets, np.iinfo(np.int64).max).min(-1, keepdims=True)
    rep = np.where(rep == np.iinfo(np.int64).max, 0, rep)
    filled = np.where(sets >= 0, sets, rep)
    tmp = np.empty((n_waves * chunks, 32), dtype=np.int64)
    tmp[:, _SET_SLOTS.reshape(-1)] = filled.reshape(n_waves * chunks, 32)
    srow = tmp.reshape(n_waves, chunks * 32)
    srow = srow.reshape(n_waves, chunks, 4, 2, 4)           # [wave, chunk, g, j, i / 4]: k = 8 g + 4 j + i / 4
    lane_i = np.arange(16)
    sl = srow[:, :, :, :, lane_i >> 2]
    a = (sl >> 3) * 512 + (sl & 7) * 32 + 8 * (lane_i & 3)               # [wave, chunk, g, j, i]: hi piece of the row
    a = a.transpose(0, 1, 3, 2, 4).reshape(n_waves, chunks, 2, 64)       # lane = 16 g + i
    assert int(a.max()) < 65536
    adr = np.zeros((n_tiles, waves, chunks, 64), dtype=np.int32)
    adr[tile_of_wave, w_in_tile] = (a[:, :, 0] | (a[:, :, 1] << 16)).astype(np.int32)

    # A fragments: sum duplicates in fp32, scale every row by its own power of two, split
    pos = pos_of_key[winv]
    slot = slot_of_row[row_of_edge]
    k = pos % 32
    dense = np.zeros((n_waves, chunks, 64, 8), dtype=np.float32)         # [wave, chunk, lane, e], lane = slot + 16 (k / 8)
    np.add.at(dense, (e_wave, pos // 32, slot + 16 * (k // 8), k % 8), val)
    rmax = np.abs(dense).reshape(n_waves, chunks, 4, 16, 8).max(axis=(1, 2, 4))          # [wave, slot], after the duplicates were summed
    with np.errstate(divide="ignore", over="ignore"):
        q = np.float32(16384.0) / rmax                     # floor(log2(q)) by exponent extraction (exact)
    e_row = np.where(rmax > 0, np.where(np.isinf(q), 126.0, np.frexp(np.where(np.isfinite(q), q, 1.0))[1] - 1.0), 0.0)
    e_row = np.clip(e_row, -126, 126)
    rscale = np.exp2(e_row).astype(np.float32)                           # [wave, slot]
    lane_scale = np.tile(rscale, (1, 4))                                 # lane = slot + 16 g
    hi, lo = split_fp16(dense * lane_scale[:, None, :, None])
    afr = np.zeros((n_tiles, waves, chunks, 2, 64, 8), dtype=np.float16)
    afr[tile_of_wave, w_in_tile, :, 0] = hi
    afr[tile_of_wave, w_in_tile, :, 1] = lo
    rinv = np.zeros((n_tiles, waves, 16), dtype=np.float32)
    filled = np.arange(16)[None, :] < rows[:, None]
    rinv[tile_of_wave, w_in_tile] = np.where(filled, np.exp2(-e_row), 0.0).astype(np.float32)

    hdr = np.zeros((n_tiles, 64), dtype=np.int32)
    hdr[tile_of_wave, waves + w_in_tile] = rows
    rowid = np.full((n_tiles, waves, rows_per_wave), -1, dtype=np.int32)
    all_rows = np.flatnonzero(wave_of_row >= 0)
    rowid[tile_of_wave[wave_of_row[all_rows]], w_in_tile[wave_of_row[all_rows]], slot_of_row[all_rows]] = all_rows
    hdr[:, 2 * waves] = union
    n_dealt = max(1, int(all_rows.size))
    stats = dict(tiles=n_tiles, waves=n_waves, rows_per_wave=float(rows.mean()),
                 rows_per_tile=float(n_dealt / n_tiles), staged_per_row=float(union.sum() / n_dealt),
                 chunk_fill=float(wkey.size / (n_waves * chunks * 32)), max_union=int(union.max()))
    return SplitPlan(torch.from_numpy(hdr), torch.from_numpy(rowid), torch.from_numpy(ucol), torch.from_numpy(afr), torch.from_numpy(adr),
                     torch.from_numpy(rinv), n_tiles, n_rows, n_cols, float(rowsum.max()), stats)


def build_split_passes(rowptr, col, val, n_rows, n_cols, waves=16, chunks=7, max_union=768, order=None,
                       rows_per_wave=16, segment=None, max_passes=12):
    """Plans for an operator with rows LONGER than a wave's column budget (the reference's full PV-US / CER-En graphs:
    ~740 / ~495 entries per row, config/largescale/sgp_pv.yaml + experiments/run_largescale_sgp.py:167-170): every
    group of ``rows_per_wave`` consecutive rows has its sorted column union cut into segments of one wave's column
    budget, pass p multiplies segment p of every group that has one and -- from the second pass on -- ADDS to the result (``plan.accumulate``; one launch per pass,
    fixed order: reproducible sums).  Returns a list of plans (one entry for an operator
    without long rows) or None."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    lim = dict(waves=waves, chunks=chunks, max_union=max_union, rows_per_wave=rows_per_wave)
    one = build_split_plan(rowptr, col, val, n_rows, n_cols, order=order, **lim)
    if one is not None:
        return [one]
    if n_rows == 0 or col.size == 0 or not np.isfinite(val).all():
        return None
    cap = int(segment or min(32 * chunks, max_union))
    # 2-D blocking: groups of ``rows_per_wave`` consecutive rows (in dealing order) x segments of ``cap`` columns of the
    # group's sorted column union -- every (group, segment) block fits one wave with all of the group's rows in it
    seq = np.arange(n_rows) if order is None else np.asarray(order, dtype=np.int64)
    group_of_row = np.full(n_rows, -1, dtype=np.int64)
    group_of_row[seq] = np.arange(seq.size) // rows_per_wave
    n_groups = (seq.size + rows_per_wave - 1) // rows_per_wave
    row_of_edge = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr[:n_rows + 1]))
    g_of_edge = group_of_row[row_of_edge]
    gkey, ginv = np.unique(g_of_edge * n_cols + col, return_inverse=True)       # distinct (group, column) pairs, sorted
    g_of_key = gkey // n_cols
    first = np.searchsorted(g_of_key, np.arange(n_groups))
    rank = np.arange(gkey.size) - first[g_of_key]                               # rank of the column in its group's union
    pass_of_edge = (rank // cap)[ginv]
    n_pass = int(pass_of_edge.max()) + 1
    if n_pass > max_passes:
        return None
    plans = []
    for p in range(n_pass):
        take = pass_of_edge == p
        cnt = np.bincount(row_of_edge[take], minlength=n_rows)
        rp = np.zeros(n_rows + 1, dtype=np.int64)
        rp[1:] = np.cumsum(cnt)
        if p == 0:
            rows_p = seq                                     # the first pass writes every row (also the empty ones)
        else:
            live = np.zeros(n_groups, dtype=bool)            # whole groups, so that waves stay aligned with them
            live[np.unique(g_of_edge[take])] = True
            rows_p = seq[live[group_of_row[seq]]]
        plan = build_split_plan(rp, col[take], val[take], n_rows, n_cols, order=rows_p, **lim)
        if plan is None:
            return None
        plan.accumulate = p > 0
        plans.append(plan)
    return plans


def plan_matrix(plan, n_rows, n_cols):
    """Dense matrix a plan encodes (hi + lo pieces, unscaled): test helper."""
    hdr, ucol, rowid = plan.hdr.numpy(), plan.ucol.numpy(), plan.rowid.numpy()
    afr, adr, rinv = plan.afr.numpy().astype(np.float64), plan.adr.numpy(), plan.rinv.numpy().astype(np.float64)
    waves, chunks = afr.shape[1], afr.shape[2]
    out = np.zeros((n_rows, n_cols))
    for t in range(plan.n_tiles):
        for w in range(waves):
            if int(hdr[t, waves + w]) == 0:
                continue
            for c in range(chunks):
                for lane in range(64):
                    m, g = lane & 15, lane >> 4
                    for e in range(8):
                        kk = 8 * g + e
                        j, i4 = (kk % 8) // 4, kk % 4
                        src_lane = 16 * g + 4 * i4                      # any lane with i / 4 == i4
                        a_ = (int(adr[t, w, c, src_lane]) >> (16 * j)) & 0xFFFF
                        s = (a_ // 512) * 8 + (a_ % 512) // 32
                        v = afr[t, w, c, 0, lane, e] + afr[t, w, c, 1, lane, e]
                        if v != 0.0:
                            out[int(rowid[t, w, m]), int(ucol[t, s])] += v * rinv[t, w, m]
    return out
