"""Host-side graph preparation: edge list -> normalised graph-shift operator in
int32 CSR, plus the tile plan the LDS-staged SpMM kernel consumes.

Semantics follow ``lib/sgp_preprocessing.py:67-105`` (``preprocess_adj``) and
``:177-192, 205-216`` (``sgp_spatial_embedding``) of the reference:
``A[i, j]`` is the weight of edge ``j -> i`` with ``edge_index[0] = j`` (source)
and ``edge_index[1] = i`` (target); duplicate edges add; ``set_diag`` replaces
the diagonal by ones, ``remove_diag`` drops it; rows are normalised by their
weighted in-degree (``D^-1 A``, zero-degree rows stay zero) or symmetrically
(``D^-1/2 A D^-1/2``) when ``gcn_norm``.

All of this is one-off work per graph (E <= a few 10^7) and runs on the host
with torch CPU ops; only the resulting arrays travel to the GPU.
"""
from dataclasses import dataclass
from typing import Optional

import os

from . import tune

import numpy as np
import torch


def _as_edge_tensors(edge_index, edge_weight):
    if isinstance(edge_index, np.ndarray):          # sgp_preprocessing.py:73-76
        edge_index = torch.from_numpy(edge_index)
        if edge_weight is not None and isinstance(edge_weight, np.ndarray):
            edge_weight = torch.from_numpy(edge_weight)
    if not torch.is_tensor(edge_index):
        raise RuntimeError("Edge index must be (edge_index, edge_weight) tuple "
                           "or SparseTensor.")          # sgp_preprocessing.py:85-87
    ei = edge_index.detach().to("cpu", torch.long)
    ew = None if edge_weight is None else edge_weight.detach().to("cpu", torch.float32)
    return ei, ew


def _coalesce(row, col, val, n):
    """Sort by (row, col) and add duplicates."""
    key = row * n + col
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    out = torch.zeros(uniq.numel(), dtype=val.dtype).index_add_(0, inv, val)
    return uniq // n, uniq % n, out


class ShiftOperator:
    """Normalised N x N operator in CSR (int32 indices, fp32 values) on the host,
    with lazily built per-device copies and tile plans.  ``op @ x`` runs the HIP
    SpMM for a CUDA tensor ``x[..., N, F]`` (the reference's ``adj @ x``,
    ``lib/sgp_preprocessing.py:202``)."""

    def __init__(self, rowptr, col, val, num_nodes, num_cols=None):
        self.rowptr = rowptr.to(torch.int32).contiguous()
        self.col = col.to(torch.int32).contiguous()
        self.val = val.to(torch.float32).contiguous()
        self.num_nodes = int(num_nodes)            # rows (= nodes owned by this operator)
        # columns; > num_nodes for the local block of a node partition, whose columns
        # num_nodes .. num_cols-1 address halo rows received from peer GPUs
        self.num_cols = int(num_nodes if num_cols is None else num_cols)
        self._dev = {}
        self._plans = {}
        self._exact_seen = False        # a split-hop admission flag came back 0: the exact kernels get their plans
        self._flag_host, self._flag_pending = None, []

    # ---- construction -----------------------------------------------------
    @classmethod
    def from_coo(cls, row, col, val, num_nodes, gcn_norm=False, set_diag=False,
                 remove_diag=False):
        n = int(num_nodes)
        row, col, val = _coalesce(row, col, val, n)
        if set_diag or remove_diag:                  # set_diag wins (:89-92)
            keep = row != col
            row, col, val = row[keep], col[keep], val[keep]
        if set_diag:
            idx = torch.arange(n, dtype=torch.long)
            row, col = torch.cat([row, idx]), torch.cat([col, idx])
            val = torch.cat([val, torch.ones(n, dtype=val.dtype)])
            order = torch.argsort(row * n + col)
            row, col, val = row[order], col[order], val[order]
        deg = torch.zeros(n, dtype=torch.float32).index_add_(0, row, val)
        if gcn_norm:                                 # :94-98
            d = deg.pow(-0.5)
            d[d == float("inf")] = 0
            val = d[row] * val * d[col]
        else:                                        # :99-103
            d = deg.pow(-1.0)
            d[d == float("inf")] = 0
            val = d[row] * val
        counts = torch.bincount(row, minlength=n)
        rowptr = torch.zeros(n + 1, dtype=torch.long)
        rowptr[1:] = torch.cumsum(counts, 0)
        return cls(rowptr, col, val, n)

    @classmethod
    def from_edges(cls, edge_index, edge_weight=None, num_nodes=None, gcn_norm=False,
                   set_diag=False, remove_diag=False, undirected=False, transpose=False):
        ei, ew = _as_edge_tensors(edge_index, edge_weight)
        n = int(num_nodes) if num_nodes is not None else (int(ei.max()) + 1 if ei.numel() else 0)
        if ei.numel() and (int(ei.min()) < 0 or int(ei.max()) >= n):
            raise ValueError("edge_index out of range for num_nodes")
        if ew is None:
            ew = torch.ones(ei.shape[1], dtype=torch.float32)
        col, row = ei[0], ei[1]                      # "transpose", :80-82
        if transpose:                                # edge_index[[1, 0]], :207
            row, col = col, row
        if undirected:                               # to_undirected, :182-185
            row, col = torch.cat([row, col]), torch.cat([col, row])
            ew = torch.cat([ew, ew])
        return cls.from_coo(row, col, ew, n, gcn_norm=gcn_norm, set_diag=set_diag,
                            remove_diag=remove_diag)

    # ---- duck-typed accessors (torch_sparse.SparseTensor-like) -------------
    def size(self, dim):
        return self.num_nodes if dim == 0 else self.num_cols

    def sparse_sizes(self):
        return (self.num_nodes, self.num_cols)

    def nnz(self):
        return int(self.col.numel())

    def csr(self):
        return self.rowptr.long(), self.col.long(), self.val

    def coo(self):
        counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
        row = torch.repeat_interleave(torch.arange(self.num_nodes), counts)
        return row, self.col.long(), self.val

    def to_dense(self):
        row, col, val = self.coo()
        a = torch.zeros(self.num_nodes, self.num_cols, dtype=torch.float32)
        a.index_put_((row, col), val, accumulate=True)
        return a

    def max_degree(self):
        if self.num_nodes == 0:
            return 0
        return int((self.rowptr[1:] - self.rowptr[:-1]).max())

    # ---- device side --------------------------------------------------------
    def device_csr(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.rowptr.to(device), self.col.to(device), self.val.to(device))
        return self._dev[key]

    def tile_plan(self, feat, device, limits=None, tall=True):
        """Tile plan for feature width ``feat`` on ``device`` or None when the graph
        has no exploitable locality (then the generic CSR kernel is used).  ``tall=False``: the
        plan with <= 64-row tiles that every LDS-staged kernel accepts, even where a tall-tile plan
        (VALU kernel only) exists."""
        key = (feat % 64 == 0, str(device))
        if key not in self._plans:
            plan = std = None
            if feat % 64 == 0 and self.nnz() > 0:
                from . import hip, plancache
                if limits is None:
                    limits = hip.tiled_limits(feat)
                tl = hip.tall_tile_limits(feat)
                plan, std = plancache.fetch(self, "tile", (sorted(limits.items()), sorted(tl.items())),
                                            lambda: self._build_tile_plans(limits, tl))
            if std is not None:
                self._plans[(key, "std")] = std.to(device)
            self._plans[key] = None if plan is None else plan.to(device)
        if not tall and (key, "std") in self._plans:     # also on the call that built both variants
            return self._plans[(key, "std")]
        return self._plans[key]

    def _build_tile_plans(self, limits, tl):
        """Host side of ``tile_plan``: ``(plan, std)`` -- the plan the staged kernels get and, where a tall-tile plan (VALU
        kernel only) replaced it, the <= 64-row plan as ``std``."""
        std = None
        plan = build_tile_plan(self.rowptr.numpy(), self.col.numpy(), self.val.numpy(), self.num_nodes, **limits)
        # No locality in the node numbering (e.g. a k-NN graph of stations listed in
        # file order): tile by a locality order computed from the graph itself.
        poor = plan is None or plan.tile_rows < 32
        if poor and self.num_cols == self.num_nodes and self.num_nodes >= 2048 and \
                self.nnz() >= 8 * self.num_nodes:
            order = locality_order(self.rowptr.numpy(), self.col.numpy(), self.num_nodes)
            alt = build_reordered_plan(self.rowptr.numpy(), self.col.numpy(), self.val.numpy(),
                                       self.num_nodes, order, **limits)
            if alt is not None and (plan is None or alt.tile_rows > plan.tile_rows):
                plan = alt
        # Sparse graphs (4-row groups share few columns: the VALU kernel serves them) gain from
        # TALL tiles: a tile stages every distinct source row of its rows once per step, so a
        # small traffic graph as ONE tile (325 rows: the whole slab of a step, 83 KB, in LDS)
        # stages each row once instead of once per 64-row tile (3.9x at 325 nodes).
        if plan is not None and not plan.reordered and plan.tile_rows <= 64 and \
                (plan.gw is None or plan.group_fill < 0.5) and limits.get("max_tile_rows", 64) <= 64 \
                and plan.max_row_edges <= 32:
            for tr in (384, 320, 256, 192, 128):
                if tr > tl["max_tile_rows"]:
                    continue
                # LDS: staged rows (whole passes of 64) + 6 bytes per edge slot of the tile's rows
                rpg = 4 if tr <= 256 else 6
                nb = 1 if plan.max_row_edges <= 16 else 2
                room = 160 * 1024 - rpg * 64 * nb * 16 * 6
                mu = min(tl["max_union"], room // (64 * 256) * 64)
                tp = build_tile_plan(self.rowptr.numpy(), self.col.numpy(), self.val.numpy(),
                                     self.num_nodes, mu, tl["max_tile_rows"], 32,
                                     candidates=(tr,)) if mu >= tr // 2 else None
                if tp is not None and tp.tile_rows > 128:
                    std, plan = plan, tp
                    break
        return plan, std

    def colblock_plan(self, feat, device):
        """Column-blocked plan (``sgp_amd.colblock``, kernel ``sgp_spmm_colblock_f32``) for graphs
        without locality, or None (feature widths that are not multiples of 64, >= ``colblock.MAX_COLS`` = 2^22 columns)."""
        key = ("colblock", feat, str(device))
        if key not in self._plans:
            plan = None
            from . import colblock
            if feat % 64 == 0 and self.num_cols < colblock.MAX_COLS and self.nnz() > 0:
                from . import hip
                lib = hip.load()
                plan = colblock.build_colblock_plan(self.rowptr.numpy(), self.col.numpy(), self.val.numpy(),
                                                    self.num_nodes, self.num_cols, feat,
                                                    rows_cap=lib.sgp_spmm_colblock_rows_cap(),
                                                    round_pad=lib.sgp_spmm_colblock_round_pad(),
                                                    l2_bytes=tune.get("colblock_l2_mb", 2.5, float) * 2 ** 20)
                if plan is not None:
                    plan = plan.to(device)
            self._plans[key] = plan
        return self._plans[key]

    def mix_plan(self, feat, device, strict=True):
        """Mixed dense / sparse plan (``sgp_amd.mixplan``, kernel ``sgp_spmm_mix_f32``) on the tiles and
        row groups of the 64-row plan, or None: needs feature widths that are multiples of 64, a
        two-phase stream, and blocks of 16 rows that share enough columns for the dense form to pay
        (k-NN-like graphs; ``SGP_TUNE=mix_min_share=..``, default 0.25 of the (group, column) pairs)."""
        key = ("mix", feat % 64 == 0, str(device), bool(strict))
        if key not in self._plans:
            plan = None
            base = self.tile_plan(feat, device, tall=False)
            if base is not None and base.pipe is not None and (base.group_fill >= 0.5 or not strict):
                from . import hip, mixplan, plancache
                lib = hip.load()
                thr, dh = tune.get("mix_thr", 4, int), lib.sgp_spmm_mix_max_dense(int(self.num_cols > self.num_nodes))

                def build():
                    order = None
                    if base.reordered:
                        order = locality_order(self.rowptr.numpy(), self.col.numpy(), self.num_nodes)
                    return mixplan.build_mix_plan(self.rowptr.numpy(), self.col.numpy(), self.val.numpy(), self.num_nodes,
                                                  base, thr=thr, dh=dh, order=order)
                # (the base plan is a function of the operator and the kernels' limits, which the key carries)
                plan = plancache.fetch(self, "mix", (thr, dh, sorted(hip.tiled_limits(feat).items())), build)
                min_share = tune.get("mix_min_share", 0.25, float) if strict else -1.0
                if plan is not None and (plan.dense_share < min_share
                                         or plan.max_union > lib.sgp_spmm_mix_max_union()):
                    plan = None
                if plan is not None:
                    plan = plan.to(device)
            self._plans[key] = plan
        return self._plans[key]

    def split_plan(self, device):
        """Plan of the split-fp16 hop (``sgp_amd.splitplan``, kernel ``sgp_spmm_split_f32``) or None.  Operators with
        rows longer than a wave's column budget (the reference's full PV-US / CER-En graphs) get a LIST of plans, one
        per pass over a segment of the columns (``splitplan.build_split_passes``); ``hip.spmm_split`` takes either."""
        key = ("split", str(device))
        if key not in self._plans:
            plan = None
            if self.nnz() > 0:
                from . import hip, plancache
                lib = hip.load()
                lim, wide = hip.split_limits(), hip.split_limits(wide=True)
                if tune.get("split_wide", 1, int) == 0:
                    wide = None
                plan = plancache.fetch(self, "split", (sorted(lim.items()), wide and sorted(wide.items()),
                                                       tune.get("split_passes", 1, int)),
                                       lambda: self._build_split_plan(lim, wide))
                if isinstance(plan, list):
                    plan = SplitPasses(p.to(device) for p in plan)
                elif plan is not None:
                    plan = plan.to(device)
            self._plans[key] = plan
        return self._plans[key]

    def _build_split_plan(self, lim, wide=None):
        """Host side of ``split_plan``: a SplitPlan, a list of them (long rows, one per pass) or None.  ``wide``: limits of
        the kernel's wide form (448 instead of 224 columns per wave at half the waves): long-row operators are planned
        for it -- half the passes, less than half the staged rows per result row."""
        from . import splitplan
        args = (self.rowptr.numpy(), self.col.numpy(), self.val.numpy(), self.num_nodes, self.num_cols)
        plan = splitplan.build_split_plan(*args, **lim)
        # numberings without locality (16 consecutive rows share no columns): deal the rows in a
        # locality order of the graph itself, as the tile plans do
        if plan is not None and plan.stats["rows_per_wave"] < 0.75 * lim["rows_per_wave"] and self.num_nodes >= 2048 and \
                self.num_cols == self.num_nodes:
            alt = splitplan.build_split_plan(*args, order=locality_order(
                self.rowptr.numpy(), self.col.numpy(), self.num_nodes), **lim)
            if alt is not None and alt.stats["staged_per_row"] < plan.stats["staged_per_row"]:
                plan = alt
        # a plan that stages many rows per result row (no locality at all) loses to the other kernels
        if plan is not None and (plan.stats["rows_per_wave"] < 0.375 * lim["rows_per_wave"] or
                                 plan.stats["staged_per_row"] > 8):
            plan = None
        if plan is None and self.max_degree() > 32 * lim["chunks"] and tune.get("split_passes", 1, int) != 0:
            # long rows: several passes over column segments, accumulated in place -- in the kernel's WIDE form (448
            # columns per wave, two waves per SIMD) where most rows are long (the full large-scale graphs), in the
            # standard form where a few hub rows sit in a graph of short ones (every row's first segment then runs at
            # the standard form's rate)
            deg = (self.rowptr[1:] - self.rowptr[:-1])
            if wide is not None and float((deg > 32 * lim["chunks"]).float().mean()) < 0.5:
                wide = None
            passes = splitplan.build_split_passes(*args, max_passes=12, **(wide or lim))
            if passes is not None and passes[0].stats["rows_per_wave"] >= 0.5 * lim["rows_per_wave"] and \
                    passes[0].stats["staged_per_row"] <= 8:
                plan = list(passes)
            elif wide is not None:                               # (the wide deal did not work out: the standard passes)
                return self._build_split_plan(lim, None)
        return plan

    def prepare(self, feat, device, halo=False):
        """Build (or load from the plan cache, ``sgp_amd.plancache``) the host-side plans ``propagate``'s DEFAULT dispatch
        needs for ``feat``-wide float32 operands on ``device``, and the device CSR.  Returns the names of what was
        prepared.  ``propagate`` does the same on first use; callers that want the one-off host work out of their
        timed region (or want to time it: ``bench.py``'s ``plan_build_s``) call this first."""
        from . import hip
        self.device_csr(device)
        if self.nnz() == 0:
            return ["csr"]
        made = []
        split_ok = tune.get("hop", "split") == "split" and feat % 16 == 0 and \
            feat <= hip.load().sgp_spmm_split_max_feat() and self.nnz() >= 8 * self.num_nodes and self.num_nodes >= 2048
        if split_ok and self.split_plan(device) is not None:
            made.append("split")
            if tune.get("exact_plans", "lazy") != "eager" and not self._exact_seen:
                return made + ["csr (behind the split hop's predicate until a flag asks for the exact kernels)"]
        if self.tile_plan(feat, device) is not None:
            made.append("tile")
            if tune.get("exact", "mix") == "mix" and self.mix_plan(feat, device) is not None:
                made.append("mix")
        elif not made and feat % 64 == 0 and self.num_cols * feat * 4 > 3 * 2 ** 20 and self.nnz() >= 16 * self.num_nodes \
                and tune.get("colblock", 1, int) != 0 and self.colblock_plan(feat, device) is not None:
            made.append("colblock")
        return made or ["csr"]

    def split_eligible(self, x, y, halo=None):
        """Whether ``propagate`` would pick the split-fp16 hop on its own for these operands (callers that
        know a bound on |x| pass it; others let ``propagate`` measure one)."""
        def ok(t):
            return t.is_cuda and t.stride(1) % 4 == 0 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and \
                t.shape[1] * max(t.stride(1), 1) < 2 ** 29
        return (tune.get("hop", "split") == "split" and ok(x) and ok(y) and (halo is None or ok(halo))
                and self._split_shape_ok(x)
                and self.nnz() >= 8 * self.num_nodes and self.num_nodes >= 2048
                and self.split_plan(x.device) is not None)

    def norm_inf(self):
        """max_i sum_j |a_ij|: |A x| <= norm_inf * max |x| (bound bookkeeping of the split-fp16 hop)."""
        if not hasattr(self, "_norm_inf"):
            rp = self.rowptr.long()
            sums = torch.zeros(self.num_nodes, dtype=torch.float64)
            if self.nnz():
                sums.index_add_(0, torch.repeat_interleave(torch.arange(self.num_nodes), rp[1:] - rp[:-1]),
                                self.val.double().abs())
            self._norm_inf = float(sums.max()) if self.num_nodes else 0.0
        return self._norm_inf

    def propagate(self, x, y, force=None, halo=None, x_bound=None):
        """y[b] = A [x[b]; halo[b]] for strided [B, N, F] CUDA views (no allocation).
        ``halo[B, num_cols - num_nodes, F]`` (any strides) supplies the columns past the
        owned rows for the local block of a node partition.  ``x_bound``: what the caller knows about max |x| -- a
        float (bounded activations), the ``hip.ColumnBound`` the previous hop left in ``self.next_bound``, or None
        (measured by one pass when the split-fp16 hop is a candidate); it sets that kernel's per-column scales and,
        with sampled statistics of x, the device-side choice between it and the exact kernel."""
        from . import hip
        if (halo is None) != (self.num_cols == self.num_nodes):
            raise ValueError("halo rows are required exactly when num_cols > num_nodes")
        if x.dim() != 3 or y.dim() != 3 or y.shape[0] != x.shape[0] or y.shape[2] != x.shape[2]:
            raise ValueError("propagate expects [B, N, F] operands of equal batch and feature size")
        n_src = x.shape[1] + (0 if halo is None else halo.shape[1])
        if n_src != self.num_cols or y.shape[1] != self.num_nodes:
            raise ValueError(f"operand shapes do not match the operator: {n_src} source rows for "
                             f"{self.num_cols} columns, {y.shape[1]} result rows for {self.num_nodes}")
        if halo is not None and (halo.shape[0] != x.shape[0] or halo.shape[2] != x.shape[2]):
            raise ValueError("halo batch / feature size differs from x")
        if force not in (None, "csr", "tiled", "res", "mix", "colblock", "split"):
            raise ValueError(f"unknown kernel {force!r} (csr, tiled, res, mix, colblock, split)")
        if x.shape[0] == 0 or x.shape[2] == 0 or self.num_nodes == 0:
            self.next_bound = self.last_split_flag = None
            return y                                      # nothing to compute (an empty time chunk)
        self._poll_flags()
        # (the tile plan of the exact kernels is built further down, only on the paths that use it: where the split-fp16
        # hop is the default the exact kernels sit behind a predicate that admits them on no shipped configuration, and
        # their plans -- 16 s of host work on the target graph -- are built the first time a flag shows they ran)
        bound_unusable = isinstance(x_bound, (int, float)) and x_bound != 0 and not (0 < x_bound < float("inf"))
        lazy_exact = force is None and not bound_unusable and tune.get("exact_plans", "lazy") != "eager" and \
            not self._exact_seen and self.split_eligible(x, y, halo)
        plan = None if force in ("csr", "colblock", "split") or lazy_exact else \
            self.tile_plan(x.shape[2], x.device, tall=force in (None, "tiled"))
        # the LDS-staged kernels address rows with 32-bit element offsets (SGP_REQUIRE in csrc: own * xrs,
        # far * xhrs, n_rows * yrs < 2^30); beyond that -- e.g. a [rows, T, D] halo receive buffer of a
        # long time chunk, whose row stride is T * D -- the generic CSR kernel (64-bit addressing) serves
        fits32 = x.shape[1] * max(x.stride(1), 1) < 2 ** 30 and y.shape[1] * max(y.stride(1), 1) < 2 ** 30 and \
            not (halo is not None and halo.shape[1] * max(halo.stride(1), 1) >= 2 ** 30)
        if not fits32 and force in (None, "csr"):
            plan = None
        # 1. split-fp16 hop: first choice where the plan exists -- UNDER A DEVICE-SIDE PREDICATE: the
        # operand's profile (per-column bounds + sampled statistics, hip.split_profile) decides on the device
        # whether the split kernel meets fp32's accuracy on this operand (flag 1) or the exact kernel enqueued
        # right behind it must run (flag 0); no host round trip.  SGP_TUNE=hop=exact keeps the exact kernels only,
        # force="split" runs the split kernel unconditionally (tests, probes).
        self.next_bound = None
        self.last_split_flag = None
        pending = None
        if isinstance(x_bound, (int, float)):
            if x_bound == 0:
                x_bound = None                          # nothing known: measured
            elif not (0 < x_bound < float("inf")):
                if force == "split":
                    raise ValueError("the split-fp16 hop needs a finite bound on |x|")
                force = force or "exact"                # a non-finite bound: exact kernels
        if force == "exact":
            force = None
        elif force == "split" or (force is None and self.split_eligible(x, y, halo)):
            splan = self.split_plan(x.device) if self._split_shape_ok(x) else None
            if splan is not None:
                if force == "split":
                    prof = hip.spmm_split(splan, x, y, hip.split_profile(x, halo, x_bound, self.norm_inf(), guard=False),
                                          halo=halo, n_own=self.num_nodes)
                    self.last_kernel = "spmm_split"
                    self.next_bound = prof.bound_out
                    return y
                prof = hip.split_profile(x, halo, x_bound, self.norm_inf(), guard=tune.get("split_guard", 1, int) != 0)
                hip.spmm_split(splan, x, y, prof, halo=halo, n_own=self.num_nodes, predicated=True)
                self.next_bound = prof.bound_out
                pending = prof.flag
            elif force == "split":
                raise NotImplementedError("no split-fp16 plan for this operator / feature width / halo / operand")
            if lazy_exact and pending is not None:
                # behind the predicate, until a flag says otherwise: the generic CSR kernel (exact fp32, no plan)
                rowptr, col, val = self.device_csr(x.device)
                hip.spmm_csr(rowptr, col, val, x, y, halo, self.num_nodes, pred=(pending, 0))
                self._watch_flag(pending)
                self.last_kernel, self.last_exact_kernel, self.last_split_flag = "spmm_split", "spmm_csr_rows", pending
                return y
            plan = self.tile_plan(x.shape[2], x.device, tall=True) if fits32 else None
        name = self._propagate_exact(x, y, force, halo, plan, fits32, pending)
        if pending is not None:
            self.last_kernel, self.last_exact_kernel, self.last_split_flag = "spmm_split", name, pending
        else:
            self.last_kernel = name
        return y

    def _split_shape_ok(self, x):
        from . import hip
        return x.shape[2] % 16 == 0 and x.shape[2] <= hip.load().sgp_spmm_split_max_feat()

    def resolved_kernel(self):
        """Name of the kernel that computed the last hop.  Where the split-fp16 hop ran under its predicate this reads
        the device flag (one 4-byte copy: a sync -- reporting paths only)."""
        flag = getattr(self, "last_split_flag", None)
        if flag is None:
            return getattr(self, "last_kernel", None)
        if int(flag.item()) == 1:
            return "spmm_split"
        self._exact_seen = True                          # the exact path ran: its planned kernels from the next hop on
        return self.last_exact_kernel

    # ---- lazily planned exact kernels: watching the admission flags without a host sync
    _FLAG_SLOTS = 32

    def _watch_flag(self, flag):
        """Enqueue a 4-byte copy of a split-hop admission flag into pinned host memory (+ an event): ``_poll_flags`` reads
        it at a later ``propagate`` once the event has passed -- never a synchronisation."""
        if self._flag_host is None:
            self._flag_host = torch.empty(self._FLAG_SLOTS, dtype=torch.int32).pin_memory()
            self._flag_pending = []
        if len(self._flag_pending) >= self._FLAG_SLOTS:
            return                                        # (every slot in flight: this hop goes unwatched)
        used = {i for i, _ in self._flag_pending}
        slot = next(i for i in range(self._FLAG_SLOTS) if i not in used)
        self._flag_host[slot:slot + 1].copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flag.device))
        self._flag_pending.append((slot, ev))

    def _poll_flags(self):
        if not self._flag_pending:
            return
        still = []
        for slot, ev in self._flag_pending:
            if ev.query():
                if int(self._flag_host[slot]) == 0:
                    self._exact_seen = True
            else:
                still.append((slot, ev))
        self._flag_pending = still

    def _propagate_exact(self, x, y, force, halo, plan, fits32, pending):
        """The exact-fp32 dispatch; ``pending``: device flag of a split-fp16 launch already enqueued for this hop --
        the kernel chosen here then runs only where that flag is 0.  Returns the kernel's name."""
        from . import hip

        def launch(fn, *args):
            fn(*args, pred=None if pending is None else (pending, 0))

        # 2. exact fp32 on the matrix cores: the mixed dense (16x16x4) / sparse (4x4x1) kernel where the planner
        # finds enough shared columns (k-NN-like graphs), else the register-resident row-group kernel
        if force in (None, "mix") and plan is not None and fits32 and \
                (force == "mix" or tune.get("exact", "mix") == "mix"):
            mplan = self.mix_plan(x.shape[2], x.device, strict=force is None)
            if mplan is not None:
                launch(hip.spmm_mix, mplan, x, y, halo, self.num_nodes)
                return "spmm_mix"
        if force == "mix":
            raise NotImplementedError("no mixed dense / sparse plan for this graph / feature width")
        if force in ("tiled", "res") and plan is None:
            raise NotImplementedError("no tile plan for this graph / feature width")
        lib = hip.load()
        use_res = plan is not None and plan.gw is not None and plan.pipe is not None and \
            (force == "res" or (force is None and plan.group_fill >= 0.5)) and \
            plan.pipe["max_tile_quads"] <= lib.sgp_spmm_res_max_quads() and \
            plan.pipe["max_union"] <= lib.sgp_spmm_res_max_union()
        if force == "res" and not use_res:
            raise NotImplementedError("no two-phase row-group stream for this plan")
        if use_res:
            launch(hip.spmm_res, plan, x, y, halo, self.num_nodes)
            return "spmm_res"
        if plan is not None and plan.reordered:
            if force == "tiled":
                raise NotImplementedError("a reordered plan serves the row-group kernels only")
            plan = None                       # generic CSR kernel
        # 3. no tile plan (no locality to stage): when a time step's source rows exceed an L2 and the rows
        # are not nearly empty, the column-blocked kernel keeps the gathers inside the L2 (random 100-column
        # rows at N = 100k: 1.3x the generic kernel); small or very sparse operators stay with CSR
        if force == "colblock" or (force is None and plan is None and fits32
                                   and x.shape[2] % 64 == 0 and x.shape[0] >= 4
                                   and self.num_cols * x.shape[2] * 4 > 3 * 2 ** 20
                                   and self.nnz() >= 16 * self.num_nodes
                                   and tune.get("colblock", 1, int) != 0):
            ok_halo = halo is None or (halo.stride(1) % 4 == 0 and halo.stride(0) % 4 == 0 and halo.data_ptr() % 16 == 0)
            cplan = self.colblock_plan(x.shape[2], x.device) if ok_halo else None
            if cplan is not None:
                launch(hip.spmm_colblock, cplan, x, y, halo, self.num_nodes)
                return "spmm_colblock"
            if force == "colblock":
                raise NotImplementedError("no column-blocked plan for this operator / feature width / halo")
        # 4. VALU form of the staged kernel (sparse graphs, tall tiles), else the generic CSR kernel
        if plan is not None:
            launch(hip.spmm_tiled, plan, x, y, halo, self.num_nodes)
            return "spmm_tiled"
        rowptr, col, val = self.device_csr(x.device)
        launch(hip.spmm_csr, rowptr, col, val, x, y, halo, self.num_nodes)
        return "spmm_csr_rows"

    def index_select(self, dim, index):
        """Rows ``index`` (order kept, repeats allowed) as a new rectangular operator -- the
        ``adj.index_select(0, node_index)`` of lib/datasets/iid_dataset.py:113."""
        if dim != 0:
            raise NotImplementedError("only row selection (dim=0) is used by the reference")
        idx = torch.as_tensor(index, dtype=torch.long).cpu()
        rp = self.rowptr.long()
        counts = (rp[1:] - rp[:-1])[idx]
        new_rp = torch.zeros(idx.numel() + 1, dtype=torch.long)
        new_rp[1:] = torch.cumsum(counts, 0)
        take = torch.repeat_interleave(rp[idx] - new_rp[:-1], counts) + torch.arange(int(new_rp[-1]))
        return ShiftOperator(new_rp, self.col[take], self.val[take], idx.numel(), num_cols=self.num_cols)

    def propagate_rect(self, x, y):
        """y[b] = A x[b] for an operator whose columns all address ``x`` (square, or the rectangular
        row subset made by ``index_select``): generic CSR kernel, no halo."""
        from . import hip
        if x.shape[1] != self.num_cols or y.shape[1] != self.num_nodes:
            raise ValueError("operand shapes do not match the operator")
        if self.num_cols == self.num_nodes:
            return self.propagate(x, y)
        rowptr, col, val = self.device_csr(x.device)
        hip.spmm_csr(rowptr, col, val, x, y)
        return y

    def __matmul__(self, x):
        if not torch.is_tensor(x) or x.dim() < 2:
            raise TypeError("ShiftOperator @ expects a dense tensor [..., N, F]")
        lead = x.shape[:-2]
        x3 = x.reshape(-1, x.shape[-2], x.shape[-1])
        if x3.dtype != torch.float32:
            x3 = x3.float()
        if x3.stride(2) != 1:
            x3 = x3.contiguous()
        if x3.shape[1] != self.num_cols:
            raise ValueError(f"operand has {x3.shape[1]} rows, the operator {self.num_cols} columns")
        on_cpu = not x3.is_cuda
        if on_cpu:
            from . import hip
            hip.require_gpu()
            x3 = x3.cuda()
        y = torch.empty(x3.shape[0], self.num_nodes, x3.shape[2], dtype=torch.float32, device=x3.device)
        self.propagate_rect(x3, y)
        if on_cpu:
            y = y.cpu()
        return y.reshape(*lead, self.num_nodes, x.shape[-1])


class SplitPasses(list):
    """Plans of the passes of a long-row operator (``splitplan.build_split_passes``); ``stats`` of the first pass."""

    @property
    def stats(self):
        return self[0].stats

    @property
    def n_tiles(self):
        return sum(p.n_tiles for p in self)


@dataclass
class TilePlan:
    """Arrays of ``sgp_spmm_tiled_f32`` (include/sgp_amd.h)."""
    trow: torch.Tensor        # int32 [n_tiles + 1], first row of every tile
    uptr: torch.Tensor        # int32 [n_tiles + 1]
    ucol: torch.Tensor        # int32 [sum of per-tile distinct columns]
    erow: torch.Tensor        # int32 [n_rows + 1], padded edge ranges (multiples of 16)
    ecol: torch.Tensor        # uint16 as int16 storage [padded nnz]
    eval: torch.Tensor        # float32 [padded nnz]
    tile_rows: int            # tallest tile
    n_tiles: int
    n_rows: int
    max_union: int
    max_row_edges: int
    gptr: Optional[torch.Tensor] = None     # int32 [16 * n_tiles + 1], quad ranges of the 4-row groups
    group_fill: float = 0.0                 # useful / issued FMAs of the row-group stream
    gidx: Optional[torch.Tensor] = None     # int32 [n_quads, 4 classes, 4] LDS byte offsets
    gw: Optional[torch.Tensor] = None       # float32 [n_quads, 4 classes, 4 rows, 4]
    max_tile_quads: int = 0
    rowmap: Optional[torch.Tensor] = None   # int32 [64 * n_tiles] output row of every (tile, slot), -1 = none
    pipe: Optional[dict] = None             # two-phase row-group stream of sgp_spmm_res_f32 / _mix (build_phase_stream)
    reordered: bool = False                 # tiles follow locality_order, not the row numbering

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        pipe = None if self.pipe is None else {
            k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.pipe.items()}
        return TilePlan(self.trow.to(device), self.uptr.to(device), self.ucol.to(device),
                        self.erow.to(device), self.ecol.to(device), self.eval.to(device),
                        self.tile_rows, self.n_tiles, self.n_rows, self.max_union,
                        self.max_row_edges, mv(self.gptr), self.group_fill,
                        mv(self.gidx), mv(self.gw), self.max_tile_quads, mv(self.rowmap), pipe,
                        self.reordered)


def tile_unions(rowptr, col, trow):
    """Per-tile sorted distinct columns for tiles of consecutive rows
    ``trow[k] .. trow[k+1]``: returns (uptr, ucol, local index per edge, row of edge)."""
    n_rows = int(trow[-1])
    n_tiles = len(trow) - 1
    deg = np.diff(rowptr).astype(np.int64)
    row_of_edge = np.repeat(np.arange(n_rows, dtype=np.int64), deg)
    tile_of_row = np.repeat(np.arange(n_tiles, dtype=np.int64), np.diff(trow))
    tile_of_edge = tile_of_row[row_of_edge]
    n_cols = int(col.max()) + 1 if col.size else 1
    key = tile_of_edge * n_cols + col.astype(np.int64)
    uniq, inv = np.unique(key, return_inverse=True)
    utile = uniq // n_cols
    ucol = (uniq % n_cols).astype(np.int32)
    uptr = np.zeros(n_tiles + 1, dtype=np.int64)
    np.add.at(uptr, utile + 1, 1)
    uptr = np.cumsum(uptr)
    lcol = inv.astype(np.int64) - uptr[tile_of_edge]
    return uptr, ucol, lcol, row_of_edge


def split_tiles(rowptr, col, n_rows, tile_rows, max_union, min_rows=8):
    """Uniform tiles of ``tile_rows`` consecutive rows; any tile that references more than
    ``max_union`` distinct columns is halved until it fits (node orders such as Morton
    have a few tiles that straddle distant regions).  None if even ``min_rows`` rows do
    not fit."""
    trow = np.arange(0, n_rows + tile_rows, tile_rows, dtype=np.int64)
    trow[-1] = n_rows
    trow = np.unique(trow)
    return refine_tiles(rowptr, col, trow, max_union, min_rows)


def refine_tiles(rowptr, col, trow, max_union, min_rows=8):
    """Halve every tile of ``trow`` that references more than ``max_union`` distinct columns until
    all fit; None if a tile of ``min_rows`` rows still does not."""
    trow = np.unique(np.asarray(trow, dtype=np.int64))
    while True:
        uptr, _, _, _ = tile_unions(rowptr, col, trow)
        over = np.nonzero(np.diff(uptr) > max_union)[0]
        if over.size == 0:
            return trow
        heights = np.diff(trow)[over]
        if (heights <= min_rows).any():
            return None
        mids = trow[over] + heights // 2
        trow = np.unique(np.concatenate([trow, mids]))


GROUP_ROWS = 4          # rows per group of the exact-fp32 row-group kernels (sgp_spmm_res_f32 / _mix)
GROUPS_PER_TILE = 16    # 16 waves per workgroup -> tiles of at most 64 rows


def cluster_rows_in_tiles(trow, uptr, lcol, row_of_edge):
    """Within every tile, order the rows so that each consecutive 4 ("row group") share as many
    source columns as possible: a group's column union is what its wave walks, so similar rows
    mean fewer steps (higher fill) and -- just as important -- groups of similar length, since a
    workgroup waits for its longest group.  Greedy: seed with the row that overlaps least with
    the rest (a corner of the tile), then add the 3 rows that overlap most with the group.
    Returns ``slot_of_row`` (position of every row inside its tile)."""
    n_tiles = len(trow) - 1
    n_rows = int(trow[-1])
    slot_of_row = np.zeros(n_rows, dtype=np.int64)
    order = np.argsort(row_of_edge, kind="stable")
    re, le = row_of_edge[order], lcol[order]
    estart = np.searchsorted(re, np.arange(n_rows + 1))
    for k in range(n_tiles):
        r0, r1 = int(trow[k]), int(trow[k + 1])
        h = r1 - r0
        if h <= GROUP_ROWS:
            slot_of_row[r0:r1] = np.arange(h)
            continue
        u = int(uptr[k + 1] - uptr[k])
        m = np.zeros((h, max(u, 1)), dtype=np.float32)
        e0, e1 = estart[r0], estart[r1]
        m[re[e0:e1] - r0, le[e0:e1]] = 1.0
        g = m @ m.T                                           # pairwise overlaps
        free = np.ones(h, dtype=bool)
        pos = 0
        while free.any():
            idx = np.flatnonzero(free)
            seed = idx[np.argmin(g[idx][:, idx].sum(1))]
            members = [seed]
            free[seed] = False
            score = g[seed].copy()
            for _ in range(GROUP_ROWS - 1):
                if not free.any():
                    break
                cand = np.flatnonzero(free)
                nxt = cand[np.argmax(score[cand])]
                members.append(nxt)
                free[nxt] = False
                score += g[nxt]
            for mrow in members:
                slot_of_row[r0 + mrow] = pos
                pos += 1
    return slot_of_row


def balance_groups_over_simds(trow, lcol, row_of_edge, slot_of_row):
    """Permute the 16 row groups of every tile over the wave slots so that the four slot
    classes s % 4 (the waves that share a SIMD, if the hardware deals a workgroup's waves
    cyclically -- a speed assumption only) carry about the same number of columns: longest
    group first, each into the least loaded class that still has room."""
    n_tiles = len(trow) - 1
    n_rows = int(trow[-1])
    tile_of_row = np.repeat(np.arange(n_tiles, dtype=np.int64), np.diff(trow))
    grp_of_row = tile_of_row * GROUPS_PER_TILE + slot_of_row // GROUP_ROWS
    key = grp_of_row[row_of_edge] * 65536 + lcol
    uniq = np.unique(key)
    counts = np.bincount(uniq >> 16, minlength=n_tiles * GROUPS_PER_TILE).reshape(n_tiles, GROUPS_PER_TILE)
    order = np.argsort(-counts, axis=1, kind="stable")               # longest first
    new_slot_of_group = np.empty_like(order)
    per_class = GROUPS_PER_TILE // 4
    for k in range(n_tiles):
        load = np.zeros(4, dtype=np.int64)
        used = np.zeros(4, dtype=np.int64)
        for g in order[k]:
            c = int(np.argmin(np.where(used < per_class, load, np.iinfo(np.int64).max)))
            new_slot_of_group[k, g] = c + 4 * used[c]
            load[c] += counts[k, g]
            used[c] += 1
    old_group = slot_of_row // GROUP_ROWS
    return new_slot_of_group[tile_of_row, old_group] * GROUP_ROWS + slot_of_row % GROUP_ROWS


def build_group_stream(trow, lcol, row_of_edge, val, slot_of_row=None):
    """Single-range row-group stream (round 1's layout; kept because ``group_fill`` and the row clustering of the
    two-phase stream are derived from it).  Slot s of a tile belongs
    to group s // 4; for every group: the sorted union of its rows' local column indices,
    dealt round-robin to 4 classes (position p -> super-step p // 4, class p % 4), stored 4
    super-steps per "quad" as weights ``gw[quad][class][row][4]`` and LDS byte offsets of the
    staged rows ``gidx[quad][class][4]``."""
    n_tiles = len(trow) - 1
    n_rows = int(trow[-1])
    tile_of_row = np.repeat(np.arange(n_tiles, dtype=np.int64), np.diff(trow))
    in_tile = np.arange(n_rows, dtype=np.int64) - trow[tile_of_row] if slot_of_row is None \
        else slot_of_row
    assert in_tile.max(initial=0) < GROUP_ROWS * GROUPS_PER_TILE
    group_of_row = tile_of_row * GROUPS_PER_TILE + in_tile // GROUP_ROWS
    slot_in_group = in_tile % GROUP_ROWS
    n_groups = n_tiles * GROUPS_PER_TILE
    g_e = group_of_row[row_of_edge]
    key = g_e * 65536 + lcol
    uniq, inv = np.unique(key, return_inverse=True)          # one entry per (group, column)
    g_s = uniq >> 16
    counts = np.bincount(g_s, minlength=n_groups)
    quads = (counts + 15) // 16                              # 16 columns per quad
    gptr = np.zeros(n_groups + 1, dtype=np.int64)
    gptr[1:] = np.cumsum(quads)
    first = np.zeros(n_groups + 1, dtype=np.int64)
    first[1:] = np.cumsum(counts)
    p = np.arange(uniq.size, dtype=np.int64) - first[g_s]    # position in the group's union
    quad = gptr[g_s] + p // 16
    sup, cls = (p // 4) % 4, p % 4
    n_quads = int(gptr[-1])
    gidx = np.zeros((n_quads, 4, 4), dtype=np.int32)           # byte offset of the staged row
    gidx[quad, cls, sup] = ((uniq & 0xffff) * 256).astype(np.int32)
    gw = np.zeros((n_quads, 4, GROUP_ROWS, 4), dtype=np.float32)
    gw[quad[inv], cls[inv], slot_in_group[row_of_edge], sup[inv]] = val
    fill = float(lcol.size) / max(1, n_quads * 16 * GROUP_ROWS)
    # row id stored per (tile, slot); -1 = no row in this slot
    rowmap = np.full(n_tiles * GROUP_ROWS * GROUPS_PER_TILE, -1, dtype=np.int32)
    rowmap[tile_of_row * (GROUP_ROWS * GROUPS_PER_TILE) + in_tile] = np.arange(n_rows, dtype=np.int32)
    return gptr.astype(np.int32), fill, gidx, gw, rowmap


def choose_segment_split(n_tiles, g_s, lc, counts):
    """Cut point ``uA`` (a multiple of 4) of every tile's distinct-column list for the two-phase
    kernel: columns < uA are staged in region A, the rest in region B, and every group walks
    ceil(nA / 16) + ceil(nB / 16) quads.  Picks, per tile, the cut that minimises the sum over
    the two phases of the busiest SIMD class (slot % 4) in quads.  ``g_s, lc`` = (group, local
    column) of every distinct (group, column) entry, ``counts`` = entries per group."""
    G = GROUPS_PER_TILE
    # columns are < 65536; histogram of entries per (group, column // 4)
    width = int(lc.max(initial=0)) // 4 + 2
    hist = np.zeros((n_tiles * G, width), dtype=np.int32)
    np.add.at(hist, (g_s, lc // 4), 1)
    below = np.zeros((n_tiles * G, width + 1), dtype=np.int32)   # below[g, j] = #cols < 4 j
    np.cumsum(hist, axis=1, out=below[:, 1:])
    tot = counts.astype(np.int32)[:, None]
    qa = (below + 15) // 16
    qb = (tot - below + 15) // 16
    # SIMD class of wave slot s is s % 4 (speed assumption only)
    qa_c = qa.reshape(n_tiles, G // 4, 4, width + 1).sum(1)
    qb_c = qb.reshape(n_tiles, G // 4, 4, width + 1).sum(1)
    ma, mb = qa_c.max(1), qb_c.max(1)                             # [n_tiles, width + 1]
    cost = ma + mb
    # each phase must be long enough to hide the DMA of the other segment: keep the shorter
    # phase at >= 40 % of the step where possible, then the cheapest cut, then the most even
    lopsided = np.minimum(ma, mb) * 5 < cost * 2
    score = lopsided.astype(np.int64) * (1 << 40) + cost.astype(np.int64) * 4096 + \
        np.minimum(np.abs(ma - mb), 4095)
    j = np.argmin(score, axis=1)
    jg = np.repeat(j, G)
    rows = np.arange(n_tiles * G)
    return (4 * j).astype(np.int32), cost[np.arange(n_tiles), j], \
        qa[rows, jg].reshape(n_tiles, G), qb[rows, jg].reshape(n_tiles, G)


def place_groups_two_phase(qa, qb):
    """Wave slot of every group (per tile) so that the four SIMD classes (slot % 4) carry about
    the same number of quads in BOTH phases: longest group first, each into the class (with a
    free slot) that keeps max_A + max_B smallest."""
    n_tiles, G = qa.shape
    per_class = G // 4
    order = np.argsort(-(qa + qb), axis=1, kind="stable")
    new_slot = np.empty_like(order)
    for k in range(n_tiles):
        la = np.zeros(4, dtype=np.int64)
        lb = np.zeros(4, dtype=np.int64)
        used = np.zeros(4, dtype=np.int64)
        for g in order[k]:
            a, b = qa[k, g], qb[k, g]
            best, best_cost = -1, None
            for c in range(4):
                if used[c] >= per_class:
                    continue
                ca = max(la.max(), la[c] + a) + max(lb.max(), lb[c] + b)
                cst = (ca, la[c] + lb[c])
                if best_cost is None or cst < best_cost:
                    best, best_cost = c, cst
            new_slot[k, g] = best + 4 * used[best]
            la[best] += a
            lb[best] += b
            used[best] += 1
    # (groups were dealt longest first, so inside a class the heaviest group sits in the lowest
    # wave slot: the SIMD serves its oldest wave first, the youngest -- which only gets the
    # matrix pipe's leftovers and finishes last -- carries the least work)
    return new_slot


def build_phase_stream(trow, uptr, ucol, lcol, row_of_edge, val, slot_of_row=None, rebalance=True,
                       mode="parity"):
    """Two-phase row-group stream of ``sgp_spmm_res_f32`` / ``sgp_spmm_mix_f32`` (include/sgp_amd.h): as ``build_group_stream`` but
    every tile's distinct-column list is cut into two segments A | B that the kernel stages
    alternately, and every group's quads are stored A-part first: ``gptr[2 g] .. gptr[2 g + 1]`` =
    quads that only touch segment A, ``gptr[2 g + 1] .. gptr[2 g + 2]`` = quads of segment B.

    ``mode="parity"``: even positions of the sorted list -> A, odd -> B, so every group finds
    about half of its columns in either segment and all waves have the same amount of work in
    both phases.  ``mode="sorted"``: A = the first ``usplit`` columns (groups at the rim of a
    tile then work in one phase only).  Returns the permuted column list (``uptr``/``ucol``,
    segment A padded to a multiple of 4 rows) along with the stream."""
    n_tiles = len(trow) - 1
    n_rows = int(trow[-1])
    tile_of_row = np.repeat(np.arange(n_tiles, dtype=np.int64), np.diff(trow))
    in_tile = np.arange(n_rows, dtype=np.int64) - trow[tile_of_row] if slot_of_row is None \
        else slot_of_row
    assert in_tile.max(initial=0) < GROUP_ROWS * GROUPS_PER_TILE
    group_of_row = tile_of_row * GROUPS_PER_TILE + in_tile // GROUP_ROWS
    slot_in_group = in_tile % GROUP_ROWS
    n_groups = n_tiles * GROUPS_PER_TILE
    g_e = group_of_row[row_of_edge]
    key = g_e * 65536 + lcol
    uniq, inv = np.unique(key, return_inverse=True)          # one entry per (group, column)
    g_s = uniq >> 16
    lc = uniq & 0xffff
    t_s = g_s // GROUPS_PER_TILE
    counts = np.bincount(g_s, minlength=n_groups)
    uptr = np.asarray(uptr, dtype=np.int64)
    U = np.diff(uptr)
    if mode == "parity":
        usplit = ((U + 1) // 2 + 3) // 4 * 4                  # rows of region A (padded)
        seg = lc & 1
        stage_slot = np.where(seg == 0, lc >> 1, usplit[t_s] + (lc >> 1))
        upad = usplit + U // 2
        # work per phase in super-steps (4 columns): the kernel skips a range's padding at that grain
        qa = (np.bincount(g_s, weights=(seg == 0), minlength=n_groups).astype(np.int64) + 3) // 4
        qb = (np.bincount(g_s, weights=(seg == 1), minlength=n_groups).astype(np.int64) + 3) // 4
        qa, qb = qa.reshape(n_tiles, -1), qb.reshape(n_tiles, -1)
    else:
        usplit, _, qa, qb = choose_segment_split(n_tiles, g_s, lc, counts)
        usplit = usplit.astype(np.int64)
        seg = (lc >= usplit[t_s]).astype(np.int64)
        stage_slot = lc
        upad = np.maximum(U, usplit)
    if rebalance:
        # re-deal the groups over the wave slots for the two-phase loads
        new_slot = place_groups_two_phase(qa, qb)
        t_of_g = np.arange(n_groups) // GROUPS_PER_TILE
        new_group = t_of_g * GROUPS_PER_TILE + new_slot.reshape(-1)
        in_tile = (new_group[group_of_row] % GROUPS_PER_TILE) * GROUP_ROWS + slot_in_group
        return build_phase_stream(trow, uptr, ucol, lcol, row_of_edge, val, in_tile,
                                  rebalance=False, mode=mode)
    # permuted / padded column list (padding repeats the tile's first column)
    uptr2 = np.zeros(n_tiles + 1, dtype=np.int64)
    uptr2[1:] = np.cumsum(upad)
    ucol = np.asarray(ucol)
    first_col = ucol[np.minimum(uptr[:-1], max(len(ucol) - 1, 0))] if len(ucol) else np.zeros(n_tiles, np.int32)
    ucol2 = np.repeat(first_col, upad).astype(np.int32)
    tile_of_u = np.repeat(np.arange(n_tiles, dtype=np.int64), U)
    l_of_u = np.arange(len(ucol), dtype=np.int64) - uptr[tile_of_u]
    if mode == "parity":
        s_of_u = np.where((l_of_u & 1) == 0, l_of_u >> 1, usplit[tile_of_u] + (l_of_u >> 1))
    else:
        s_of_u = l_of_u
    ucol2[uptr2[tile_of_u] + s_of_u] = ucol
    # entries ordered by (half-group, staged slot)
    h_s = 2 * g_s + seg
    order = np.argsort(h_s * 65536 + stage_slot, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    h_s, stage_slot = h_s[order], stage_slot[order]
    inv = rank[inv]
    hcounts = np.bincount(h_s, minlength=2 * n_groups)
    quads = (hcounts + 15) // 16
    gsup = (hcounts + 3) // 4                                 # super-steps actually occupied
    gptr = np.zeros(2 * n_groups + 1, dtype=np.int64)
    gptr[1:] = np.cumsum(quads)
    first = np.zeros(2 * n_groups + 1, dtype=np.int64)
    first[1:] = np.cumsum(hcounts)
    p = np.arange(h_s.size, dtype=np.int64) - first[h_s]     # position in the half-group's list
    quad = gptr[h_s] + p // 16
    sup, cls = (p // 4) % 4, p % 4
    n_quads = int(gptr[-1])
    gidx = np.zeros((n_quads, 4, 4), dtype=np.int32)
    gidx[quad, cls, sup] = (stage_slot * 256).astype(np.int32)
    # [quad][class][super-step][row]: one float per lane of the wave (lane = 16 class + 4 sup + row)
    gw = np.zeros((n_quads, 4, 4, GROUP_ROWS), dtype=np.float32)
    gw[quad[inv], cls[inv], sup[inv], slot_in_group[row_of_edge]] = val
    fill = float(lcol.size) / max(1, int(gsup.sum()) * 4 * GROUP_ROWS)
    max_tile_quads = int(np.diff(gptr[::2 * GROUPS_PER_TILE]).max()) if n_tiles else 0
    rowmap = np.full(n_tiles * GROUP_ROWS * GROUPS_PER_TILE, -1, dtype=np.int32)
    rowmap[tile_of_row * (GROUP_ROWS * GROUPS_PER_TILE) + in_tile] = np.arange(n_rows, dtype=np.int32)
    hq = gsup.reshape(n_tiles, GROUPS_PER_TILE // 4, 4, 2).sum(1)              # [tile, class, phase]
    phase_cost = hq.max(1).sum(1)                                              # in super-steps
    return dict(usplit=usplit.astype(np.int32), uptr=uptr2.astype(np.int32), ucol=ucol2,
                gptr=gptr.astype(np.int32), gsup=gsup.astype(np.int32), gidx=gidx, gw=gw, fill=fill,
                max_tile_quads=max_tile_quads, max_union=int(upad.max(initial=0)),
                max_range_steps=int(gsup.max(initial=0)),
                phase_cost=phase_cost, rowmap=rowmap)


def locality_order(rowptr, col, n):
    """A node order with 2-D locality computed from the graph alone (no coordinates): hop
    distances from two pairs of far-apart landmarks (each found by a double BFS sweep) act as two
    axes, ``x = d(a, .) - d(b, .)``, ``y = d(c, .) - d(d, .)``, and the nodes are sorted by the
    Morton code of ``(x, y)``.  For a geometric k-NN graph whose node labels are scrambled this
    brings the distinct-column count of a 64-row tile to within ~6 % of the order by the true
    coordinates (396 vs 372 staged rows at N = 100 000; 6 300 without reordering).  Six BFS
    sweeps: ~11 s at nnz = 10^7, one-off per graph."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import dijkstra
    from .synthetic import morton_order
    adj = sp.csr_matrix((np.ones(col.size, np.float32), col.astype(np.int64), rowptr.astype(np.int64)),
                        shape=(n, n))
    adj = (adj + adj.T).tocsr()

    def hops(src):
        d = dijkstra(adj, directed=False, indices=int(src), unweighted=True)
        finite = np.isfinite(d)
        d[~finite] = (d[finite].max() if finite.any() else 0) + 1     # other components: far away
        return d

    da = hops(0)
    a = int(np.argmax(da)); da = hops(a)
    b = int(np.argmax(da)); db = hops(b)
    c = int(np.argmax(np.minimum(da, db))); dc = hops(c)             # far from both ends of axis 1
    d_ = int(np.argmax(dc)); dd = hops(d_)
    xy = np.stack([da - db, dc - dd], 1).astype(np.float64)
    xy -= xy.min(0)
    xy /= np.maximum(xy.max(0), 1.0)
    return morton_order(xy, bits=12)


def build_reordered_plan(rowptr, col, val, n_rows, order, **limits):
    """Tile plan of the operator with rows and columns renumbered by ``order`` (new id k = old id
    ``order[k]``), expressed in the ORIGINAL ids: the kernels gather source rows through ``ucol``
    and write output rows through ``rowmap``, so a tile need not be a run of consecutive rows and
    no tensor is ever permuted.  (The DPP kernel addresses a tile's rows as ``row0 + r``: a
    reordered plan serves the row-group kernels only.)"""
    import scipy.sparse as sp
    order = np.asarray(order, dtype=np.int64)
    pos = np.empty(n_rows, dtype=np.int64)
    pos[order] = np.arange(n_rows)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(rowptr))
    a = sp.csr_matrix((np.asarray(val), (pos[rows], pos[np.asarray(col, dtype=np.int64)])),
                      shape=(n_rows, n_rows))
    a.sort_indices()
    plan = build_tile_plan(a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.float32),
                           n_rows, **limits)
    if plan is None or plan.gw is None:
        return None
    order32 = torch.from_numpy(order.astype(np.int32))

    def back_rows(t):                       # row ids (-1 = empty slot) -> original ids
        t = t.long()
        return torch.where(t >= 0, order32[t.clamp_min(0)].long(), t).int()

    plan.ucol = order32[plan.ucol.long()]
    plan.rowmap = back_rows(plan.rowmap)
    if plan.pipe is not None:
        plan.pipe["ucol"] = order32[plan.pipe["ucol"].long()]
        plan.pipe["rowmap"] = back_rows(plan.pipe["rowmap"])
    plan.reordered = True
    return plan


def equal_cost_tiles(rowptr, col, n_rows, trow, tile_cost, max_rows, max_union, quantile=0.15):
    """Tile boundaries with (about) EQUAL cost per tile.  Workgroups of an XCD that take the same
    time per step stay on the same time steps, and the staged rows they share are then read from
    the L2 instead of the fabric: with uniform 64-row tiles the cost of a step varies
    by +-20 % (and by 2x for tiles halved at the LDS limit), the workgroups drift ~8 steps apart
    and half of the staging reads miss.  ``tile_cost`` = critical super-steps per step of the
    current tiles; the per-row cost density derived from it is re-cut greedily into runs of
    ``target`` cost (the ``quantile`` of the full tiles' costs: cheaper regions keep ``max_rows``
    rows, dearer ones get fewer), then halved where the LDS limit still bites."""
    trow = np.asarray(trow, dtype=np.int64)
    rows = np.diff(trow)
    cost = np.asarray(tile_cost, dtype=np.float64)
    full = rows == rows.max()
    if not full.any():
        return trow
    target = float(np.quantile(cost[full], quantile))
    dens = np.repeat(cost / np.maximum(rows, 1), rows)            # cost per row
    cum = np.concatenate([[0.0], np.cumsum(dens)])
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col)
    stamp = np.full(int(col.max()) + 1 if col.size else 1, -1, dtype=np.int64)
    cuts = [0]
    r, tid = 0, 0
    while r < n_rows:
        # grow the tile 4 rows (one row group) at a time while it stays within the cost target,
        # the row limit and the LDS limit on distinct source rows
        end, union = r, 0
        while end < n_rows and end - r < max_rows:
            nxt = min(n_rows, end + 4)
            cols = np.unique(col[rowptr[end]:rowptr[nxt]])
            fresh = cols[stamp[cols] != tid]
            if end > r and (union + fresh.size > max_union or cum[nxt] - cum[r] > target * 1.0001):
                break
            stamp[fresh] = tid
            union += fresh.size
            end = nxt
        cuts.append(end)
        r, tid = end, tid + 1
    return refine_tiles(rowptr, col, np.asarray(cuts, dtype=np.int64), max_union)


def build_tile_plan(rowptr, col, val, n_rows, max_union, max_tile_rows, max_row_edges,
                    candidates=(64, 32, 16), cluster=True, trow_override=None,
                    equalize=None) -> Optional[TilePlan]:
    """Tallest tiling whose per-tile working set fits the LDS stage, or None when the
    graph has no locality to exploit (average tile would stage more than it reuses)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    deg = np.diff(rowptr)
    if n_rows == 0 or col.size == 0 or max_union <= 0:
        return None
    pad_deg = ((deg + 15) // 16) * 16
    mre = int(pad_deg.max())
    if mre > max_row_edges:
        return None
    for tr in candidates:
        if tr > max_tile_rows:
            continue
        # the 4-rows-per-group x 8-batch kernel variant spills; keep tall tiles for short rows
        if tr > 64 and mre > 32:
            continue
        if trow_override is not None:
            trow = np.asarray(trow_override, dtype=np.int64)
        else:
            trow = split_tiles(rowptr, col, n_rows, tr, min(max_union, 65535))
        if trow is None:
            continue
        n_tiles = len(trow) - 1
        if trow_override is None and n_tiles > 1.5 * ((n_rows + tr - 1) // tr) + 1:
            continue                       # mostly split: a smaller uniform height is better
        uptr, ucol, lcol, row_of_edge = tile_unions(rowptr, col, trow)
        mu = int(np.diff(uptr).max())
        erow = np.zeros(n_rows + 1, dtype=np.int64)
        erow[1:] = np.cumsum(pad_deg)
        ecol = np.zeros(int(erow[-1]), dtype=np.uint16)
        evalv = np.zeros(int(erow[-1]), dtype=np.float32)
        pos = erow[row_of_edge] + (np.arange(col.size, dtype=np.int64) - rowptr[row_of_edge])
        ecol[pos] = lcol.astype(np.uint16)
        evalv[pos] = val
        plan = TilePlan(torch.from_numpy(trow.astype(np.int32)),
                        torch.from_numpy(uptr.astype(np.int32)), torch.from_numpy(ucol),
                        torch.from_numpy(erow.astype(np.int32)),
                        torch.from_numpy(ecol.view(np.int16)), torch.from_numpy(evalv),
                        int(np.diff(trow).max()), n_tiles, int(n_rows), mu, mre)
        if plan.tile_rows <= GROUP_ROWS * GROUPS_PER_TILE:
            slots = cluster_rows_in_tiles(trow, uptr, lcol, row_of_edge) if cluster else None
            if slots is not None:
                slots = balance_groups_over_simds(trow, lcol, row_of_edge, slots)
            gptr, fill, gidx, gw, rowmap = build_group_stream(
                trow, lcol, row_of_edge, np.asarray(val), slots)
            plan.gptr, plan.group_fill = torch.from_numpy(gptr), fill
            plan.gidx, plan.gw = torch.from_numpy(gidx), torch.from_numpy(gw)
            plan.rowmap = torch.from_numpy(rowmap)
            plan.max_tile_quads = int(np.diff(gptr[::GROUPS_PER_TILE].astype(np.int64)).max())
            ps = build_phase_stream(trow, uptr, ucol, lcol, row_of_edge, np.asarray(val), slots)
            plan.pipe = dict(usplit=torch.from_numpy(ps["usplit"]), gptr=torch.from_numpy(ps["gptr"]),
                             gsup=torch.from_numpy(ps["gsup"]),
                             uptr=torch.from_numpy(ps["uptr"]), ucol=torch.from_numpy(ps["ucol"]),
                             max_union=ps["max_union"],
                             gidx=torch.from_numpy(ps["gidx"]), gw=torch.from_numpy(ps["gw"]),
                             rowmap=torch.from_numpy(ps["rowmap"]), fill=ps["fill"],
                             max_tile_quads=ps["max_tile_quads"],
                             max_range_steps=ps["max_range_steps"],
                             phase_cost=ps["phase_cost"])
            if equalize is None:
                equalize = tune.get("equal_cost_tiles", 0, int) == 1
            if equalize and trow_override is None and tr == 64 and n_tiles >= 512:
                new_trow = equal_cost_tiles(rowptr, col, n_rows, trow, ps["phase_cost"], tr,
                                            min(max_union, 65535),
                                            tune.get("equal_cost_q", 0.15, float))
                if new_trow is not None and len(new_trow) - 1 <= 1.4 * n_tiles:
                    alt = build_tile_plan(rowptr, col, val, n_rows, max_union, max_tile_rows,
                                          max_row_edges, candidates=(tr,), cluster=cluster,
                                          trow_override=new_trow, equalize=False)
                    if alt is not None and alt.pipe is not None:
                        return alt
        return plan
    return None
