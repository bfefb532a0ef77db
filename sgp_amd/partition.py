"""Node-partitioned encoder across the GPUs of one node.

The reference is single-process; the encoder nevertheless shards naturally:

* reservoir: every node's recurrence is independent (``reservoir.py:166`` flattens
  ``(b n)``), so rank g runs rows ``[lo_g, hi_g)`` with no communication;
* hop k: rank g computes its rows of ``A . X`` and needs ``X[:, c, :]`` for every column its
  rows reference -- owned columns are local, the rest ("halo") are fetched from their owners
  with ONE ``all_to_all_single`` per hop over RCCL/xGMI (point-to-point links: every peer's
  rows travel their own link, nothing is relayed).  When the halo is nearly everything (a graph
  without locality that no renumbering rescues: the ranks together need more than half of all
  remote rows) the packing buys nothing and costs a gather kernel per hop: the exchange is then ONE
  ``all_gather`` of every rank's full shard (SURVEY.md 8e's general case) and the local operator
  addresses the gathered buffer directly;
* ``global_attr``: ``all_reduce(sum)`` of the ``[T, D_h]`` partial column sums.

One process per GPU; ``torch.distributed`` supplies the process group (``nccl`` == RCCL on
ROCm, ``gloo`` in the CPU tests).  Device work goes through an ``ops`` object -- the HIP
binding by default.  The CPU tests substitute a plain-torch stand-in for it so that the
partition / halo / collective logic is exercised with world_size 2 on a box without GPUs;
that stand-in lives in ``tests/`` and is never used by the product.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import hip
from .graph import ShiftOperator


def partition_bounds(n_nodes, world_size, rowptr=None):
    """Contiguous row blocks: equal rows, or equal nnz when a CSR ``rowptr`` is given."""
    if rowptr is None:
        return [(n_nodes * r) // world_size for r in range(world_size + 1)]
    rp = np.asarray(rowptr, dtype=np.int64)
    targets = (rp[-1] * np.arange(1, world_size)) / world_size
    cuts = np.searchsorted(rp, targets, side="left")
    return [0] + [int(c) for c in cuts] + [int(n_nodes)]


def permute_operator(op: ShiftOperator, order) -> ShiftOperator:
    """The square operator with nodes renumbered so that new id k is old id ``order[k]``
    (P A P^T): row k of the result is row ``order[k]`` of ``op`` with its columns relabelled."""
    order = torch.as_tensor(order, dtype=torch.long)
    n = op.num_nodes
    pos = torch.empty(n, dtype=torch.long)
    pos[order] = torch.arange(n)
    rp = op.rowptr.long()
    counts = (rp[1:] - rp[:-1])[order]
    new_rp = torch.zeros(n + 1, dtype=torch.long)
    new_rp[1:] = torch.cumsum(counts, 0)
    take = torch.repeat_interleave(rp[order] - new_rp[:-1], counts) + torch.arange(int(new_rp[-1]))
    col, val = pos[op.col.long()[take]], op.val[take]
    # columns sorted inside every row (the CSR kernels and the tile planner expect it)
    key = torch.repeat_interleave(torch.arange(n), counts) * n + col
    srt = torch.argsort(key)
    return ShiftOperator(new_rp, col[srt], val[srt], n)


def halo_rows(op: ShiftOperator, bounds, rank):
    """Number of distinct columns outside ``rank``'s row block that its rows reference."""
    return int(_halo_of(op, bounds[rank], bounds[rank + 1])[0].numel())


@dataclass
class LocalBlock:
    """Rows ``[lo, hi)`` of a global operator, columns renumbered ``[owned | halo]``."""
    op: ShiftOperator                 # hi-lo rows, (hi-lo) + n_halo columns
    lo: int
    hi: int
    halo_global: torch.Tensor         # int64 [n_halo] global ids of the halo columns (sorted)
    recv_counts: List[int]            # halo rows owned by each peer (contiguous in halo order)
    send_index: torch.Tensor          # int32 [sum(send_counts)] LOCAL row ids, peer-major
    send_counts: List[int]            # rows each peer needs from me
    gather_rows: int = 0              # > 0: all_gather exchange; every rank contributes this many rows
    #                                   (its shard padded to the largest one) and ``op``'s halo columns
    #                                   index the gathered buffer: column n_own + p * gather_rows + i is
    #                                   row bounds[p] + i

    @property
    def n_own(self):
        return self.hi - self.lo

    @property
    def n_halo(self):
        if self.gather_rows:
            return self.gather_rows * len(self.recv_counts)
        return int(self.halo_global.numel())


def _halo_of(op, lo, hi):
    rp = op.rowptr.long()
    cols = op.col[rp[lo]:rp[hi]].long()
    outside = cols[(cols < lo) | (cols >= hi)]
    return torch.unique(outside, sorted=True), cols


def split_operator(op: ShiftOperator, bounds, rank, exchange="packed", halos=None) -> LocalBlock:
    """Local block of ``rank`` plus the halo bookkeeping, computed from the full operator
    (every rank holds the whole graph: it is tiny next to the node features).
    ``exchange``: "packed" (default here) = all_to_all of the rows peers reference, "gather" =
    all_gather of full shards, "auto" (what ``make_partitioned_spatial`` passes) = gather when the
    ranks together reference more than half of all remote rows (every rank evaluates the same global
    figure, so all take the same branch)."""
    world = len(bounds) - 1
    lo, hi = bounds[rank], bounds[rank + 1]
    halo, cols = _halo_of(op, lo, hi)                               # (``halos``: every rank's halo list, computed once
    n_own = hi - lo                                                 # by ``plan_partition`` when one process cuts all blocks)
    own = (cols >= lo) & (cols < hi)
    rp = op.rowptr.long()
    rowptr = rp[lo:hi + 1] - rp[lo]
    vals = op.val[rp[lo]:rp[hi]]
    b = torch.tensor(bounds)
    owner = torch.bucketize(halo, b[1:], right=True)                # rank owning each halo column
    recv_counts = torch.bincount(owner, minlength=world).tolist()
    send_idx, send_counts = [], []
    halo_total = int(halo.numel())
    for p in range(world):                                          # what does p need from me?
        if p == rank:
            send_counts.append(0)
            continue
        ph = halos[p] if halos is not None else _halo_of(op, bounds[p], bounds[p + 1])[0]
        halo_total += int(ph.numel())
        mine = ph[(ph >= lo) & (ph < hi)] - lo
        send_idx.append(mine)
        send_counts.append(int(mine.numel()))
    send_index = (torch.cat(send_idx) if send_idx else torch.zeros(0, dtype=torch.long)).int()
    n = int(bounds[-1])
    remote_total = sum(n - (bounds[p + 1] - bounds[p]) for p in range(world))
    if exchange == "gather" or (exchange == "auto" and world > 1 and 2 * halo_total > remote_total):
        g_rows = max(bounds[p + 1] - bounds[p] for p in range(world))
        col_owner = torch.bucketize(cols, b[1:], right=True)
        local = torch.where(own, cols - lo, n_own + col_owner * g_rows + (cols - b[col_owner]))
        block = ShiftOperator(rowptr, local, vals, n_own, num_cols=n_own + world * g_rows)
        return LocalBlock(block, lo, hi, halo, recv_counts, send_index, send_counts, gather_rows=int(g_rows))
    local = torch.where(own, cols - lo, n_own + torch.searchsorted(halo, cols))
    block = ShiftOperator(rowptr, local, vals, n_own, num_cols=n_own + int(halo.numel()))
    return LocalBlock(block, lo, hi, halo, recv_counts, send_index, send_counts)


class HipOps:
    """Device operations of the partitioned encoder on the MI355X (the product path)."""

    @staticmethod
    def gather_nodes(x, index, out):
        return hip.gather_nodes(x, index, out)

    @staticmethod
    def propagate(op, x, y, halo, x_bound=None):
        return op.propagate(x, y, halo=halo, x_bound=x_bound)

    @staticmethod
    def node_sums(x):
        return hip.node_sums(x)

    @staticmethod
    def bcast_rows(src, scale, y):
        return hip.bcast_rows(src, scale, y)


class HaloExchange:
    """Per-hop exchange of the rows peers reference.  Buffers are ``[rows, T, D]`` so that
    ``all_to_all_single`` splits them along dim 0; the SpMM reads the receive buffer in place
    through its (row, batch) strides.  One instance per (operator, time chunk)."""

    def __init__(self, block: LocalBlock, group=None, ops=HipOps):
        self.block, self.group, self.ops = block, group, ops
        self._send = self._recv = None
        self._idx = None

    def _buffers(self, T, D, device):
        """Contiguous ``[rows, T, D]`` views of flat buffers that only ever grow: time pieces of unequal length
        (``encode_partitioned`` cuts T into ``(T * j) // pieces``) re-use one allocation instead of alternating
        between two shapes."""
        g = self.block.gather_rows
        rows_s, rows_r = (g if g else sum(self.block.send_counts)), self.block.n_halo
        need_s, need_r = rows_s * T * D, rows_r * T * D
        if self._send is None or self._send.device != device or self._send.numel() < need_s \
                or self._recv.numel() < need_r:
            # (gather form: the rows past a short shard are never referenced, but they travel: keep them finite)
            self._send = (torch.zeros if g else torch.empty)(max(need_s, 1), dtype=torch.float32, device=device)
            self._recv = torch.empty(max(need_r, 1), dtype=torch.float32, device=device)
            self._idx = (torch.arange(self.block.n_own, dtype=torch.int32) if g else self.block.send_index).to(device)
        return self._send[:need_s].view(rows_s, T, D), self._recv[:need_r].view(rows_r, T, D)

    def _all_gather(self, x, send, recv):
        """General case: every rank's full shard (padded to the largest) to every rank."""
        if self.block.n_own:
            self.ops.gather_nodes(x, self._idx, send.permute(1, 0, 2)[:, :self.block.n_own])
        world = len(self.block.recv_counts)
        if send.is_cuda and dist.get_backend(self.group) == "gloo":      # shared-GPU test rigs only
            parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(world)]
            dist.all_gather(parts, send.cpu(), group=self.group)
            recv.copy_(torch.cat(parts, 0))
        else:
            dist.all_gather(list(recv.chunk(world, 0)), send, group=self.group)
        return recv.permute(1, 0, 2)

    def __call__(self, x):
        """x[T, n_own, D] (strided view) -> halo[T, n_halo, D] view of the receive buffer."""
        T, _, D = x.shape
        send, recv = self._buffers(T, D, x.device)
        if self.block.gather_rows:
            return self._all_gather(x, send, recv)
        if send.shape[0]:
            self.ops.gather_nodes(x, self._idx, send.permute(1, 0, 2))
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            # test rigs only (several ranks sharing one GPU): gloo has no device all_to_all
            r_cpu = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(r_cpu, send.cpu(), output_split_sizes=self.block.recv_counts,
                                   input_split_sizes=self.block.send_counts, group=self.group)
            recv.copy_(r_cpu)
        else:
            dist.all_to_all_single(recv, send, output_split_sizes=self.block.recv_counts,
                                   input_split_sizes=self.block.send_counts, group=self.group)
        return recv.permute(1, 0, 2)


class PartitionedSpatial:
    """K-hop propagation + global mean of ``SGPSpatialEncoder.encode_into`` for the local
    node block: fills ``out[T, n_own, P * feat]`` in place (block 0 already written).

    On GPUs the time axis is cut into ``n_chunks`` pieces and the hop loop is software-pipelined
    over (hop, chunk): the row packing + all_to_all of a chunk run on a communication stream
    while the SpMM of the previous chunk runs on the compute stream, and hop h+1 of a chunk
    starts as soon as hop h of THAT chunk is done (time steps are independent in the
    propagation).  xGMI is point-to-point, so the exchange costs about as much as a hop's
    compute at 8 GPUs; hidden behind it, scaling stays close to the compute curve."""

    def __init__(self, blocks: List[LocalBlock], receptive_field, global_attr, n_total,
                 group=None, ops=HipOps, n_chunks=4, force_collectives=False):
        self.blocks = blocks                       # forward (+ backward) local blocks
        self.k, self.global_attr, self.n_total = receptive_field, global_attr, n_total
        self.group, self.ops = group, ops
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_chunks = max(1, int(n_chunks))
        # run the exchange / reduction even on a one-rank group (self-test of the RCCL plumbing
        # on a single-GPU box: a collective over one rank is valid and moves nothing)
        self._dist = self.world_size > 1 or (force_collectives and dist.is_initialized())
        self._xchg = {}
        self._comm_stream = None
        self._res_stream = None                    # encode_partitioned: reservoir of the next time piece
        # bench.py sets this to a list: (kind, start event, end event) per exchange ("comm", on the
        # communication stream) and per SpMM launch ("hop", on the compute stream)
        self.timeline = None
        self.node_order = None                     # set by make_partitioned_spatial
        # infinity norms of the GLOBAL operators (one per direction): a hop multiplies the bound on |x| by it -- the
        # same figure on every rank, whose halo rows are other ranks' results (set by make_partitioned_spatial)
        self.norm_inf = None

    def num_blocks(self):
        return 1 + len(self.blocks) * self.k + (1 if self.global_attr else 0)

    def _exchange(self, d, j):
        key = (d, j)
        if key not in self._xchg:
            self._xchg[key] = HaloExchange(self.blocks[d], self.group, self.ops)
        return self._xchg[key]

    def _bound_after(self, bound, d):
        if bound is None or self.norm_inf is None:
            return None
        return bound * max(self.norm_inf[d], 1e-30) * (1 + 1e-6)

    def _propagate(self, blk, src, dst, halo, bound):
        if bound is None:
            return self.ops.propagate(blk.op, src, dst, halo)          # (test stand-ins take four arguments)
        return self.ops.propagate(blk.op, src, dst, halo, bound)

    def _hops_serial(self, out, feat, x_bound=None):
        for d, blk in enumerate(self.blocks):
            src = out[:, :, 0:feat]
            bound = x_bound
            for h in range(self.k):
                s = 1 + d * self.k + h
                dst = out[:, :, s * feat:(s + 1) * feat]
                # every rank enters the collective, even one whose block has no halo
                timed = self.timeline is not None and out.is_cuda
                if timed:
                    c0, c1, h1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    c0.record()
                halo = self._exchange(d, 0)(src) if self._dist else None
                if timed:
                    c1.record()
                self._propagate(blk, src, dst, halo if blk.n_halo else None, bound)
                bound = self._bound_after(bound, d)
                if timed:
                    h1.record()
                    self.timeline.append(("comm", c0, c1))
                    self.timeline.append(("hop", c1, h1))
                src = dst

    def _hops_pipelined(self, out, feat, x_bound=None):
        T = out.shape[0]
        nc = min(self.n_chunks, T)
        cuts = [(T * j) // nc for j in range(nc + 1)]
        main = torch.cuda.current_stream(out.device)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=out.device)
        comm = self._comm_stream
        start = torch.cuda.Event()
        start.record(main)                          # block 0 (reservoir states) is complete
        for d, blk in enumerate(self.blocks):
            ready = [start] * nc                    # source slot of chunk j is written
            bound = x_bound
            for h in range(self.k):
                s_src = 0 if h == 0 else 1 + d * self.k + h - 1
                s_dst = 1 + d * self.k + h
                done = []
                for j in range(nc):
                    t0, t1 = cuts[j], cuts[j + 1]
                    src = out[t0:t1, :, s_src * feat:(s_src + 1) * feat]
                    dst = out[t0:t1, :, s_dst * feat:(s_dst + 1) * feat]
                    timed = self.timeline is not None
                    with torch.cuda.stream(comm):
                        comm.wait_event(ready[j])
                        if timed:
                            c0 = torch.cuda.Event(enable_timing=True)
                            c0.record(comm)
                        halo = self._exchange(d, j)(src)
                        got = torch.cuda.Event(enable_timing=timed)
                        got.record(comm)
                    # the receive buffer was allocated on the communication stream and is read on
                    # the compute stream
                    halo.record_stream(main)
                    main.wait_event(got)
                    if timed:
                        h0 = torch.cuda.Event(enable_timing=True)
                        h0.record(main)
                    self._propagate(blk, src, dst, halo if blk.n_halo else None, bound)
                    ev = torch.cuda.Event(enable_timing=timed)
                    ev.record(main)
                    done.append(ev)
                    if timed:
                        self.timeline.append(("comm", c0, got))
                        self.timeline.append(("hop", h0, ev))
                ready = done
                bound = self._bound_after(bound, d)
        # the communication stream's buffers are reused by the next call: let it catch up
        comm.wait_stream(main)

    def encode_into(self, out, feat, col_sums=None, x_bound=None):
        """``col_sums`` [T, feat]: sums over the OWNED rows of slot 0 when the producer has them
        already (``Reservoir.encode_into``); they are all-reduced over the ranks like the sums this
        method would otherwise compute from slot 0.  ``x_bound`` >= max |slot 0| over ALL ranks where the
        caller knows it (bounded reservoir activations): the split-fp16 hop scales its operands by it."""
        if self._dist and out.is_cuda and self.n_chunks > 1 and out.shape[0] >= 8:
            self._hops_pipelined(out, feat, x_bound)
        else:
            self._hops_serial(out, feat, x_bound)
        if self.global_attr:
            p = self.num_blocks() - 1
            sums = col_sums if col_sums is not None else self.ops.node_sums(out[:, :, :feat])
            if self._dist:
                if sums.is_cuda and dist.get_backend(self.group) == "gloo":
                    s_cpu = sums.cpu()
                    dist.all_reduce(s_cpu, op=dist.ReduceOp.SUM, group=self.group)
                    sums.copy_(s_cpu)
                else:
                    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
            self.ops.bcast_rows(sums, 1.0 / self.n_total, out[:, :, p * feat:(p + 1) * feat])
        return out


def encode_partitioned(reservoir, spatial: "PartitionedSpatial", x, out, state=None, pieces=None):
    """Reservoir + K hops (+ global block) of the local node block with the reservoir of time
    piece c + 1 running UNDER the hops and halo exchange of piece c ("reservoir chunk
    c + 1 || SpMM chunk c || halo chunk c").  The recurrence is sequential in time, so the pieces of
    the reservoir follow each other on their own stream with the state carried in ``state``
    [L, n_own, R]; the propagation is independent per time step, so piece c's hops only wait for
    piece c's states.  Same kernels on the same data as reservoir-then-hops: identical results.
    ``x[T, n_own, F]``, ``out[T, n_own, P * D_h]`` on the GPU."""
    T = x.shape[0]
    d_h = reservoir.output_size
    # bounded activations bound the states on every rank alike -- under the premises of
    # sgp_encoder.SGPEncoder._state_bound: leaking rates in [0, 1], a recurrence that starts inside [-1, 1]
    bound = None
    if x.is_cuda and getattr(reservoir, "mode", None) in ("tanh", "self_norm") and \
            all(0.0 <= float(l.alpha) <= 1.0 for l in reservoir.reservoir_layers) and \
            (state is None or hip.is_unit_bounded(state)):
        bound = 1.0
    pieces = spatial.n_chunks if pieces is None else pieces
    pieces = max(1, min(int(pieces), T // 8)) if x.is_cuda else 1
    want_sums = spatial.global_attr and x.is_cuda and reservoir.produces_col_sums(x)
    if pieces <= 1 or not x.is_cuda:
        sums = torch.empty(T, d_h, dtype=torch.float32, device=x.device) if want_sums else None
        reservoir.encode_into(x, out[:, :, :d_h], state, col_sums=sums)
        if bound is not None and state is not None:
            hip.mark_unit_bounded(state)
        return spatial.encode_into(out, d_h, col_sums=sums, x_bound=bound)
    if state is None:
        state = torch.zeros(len(reservoir.reservoir_layers), x.shape[1], reservoir.hidden_size,
                            dtype=torch.float32, device=x.device)
    if bound is not None:
        hip.mark_unit_bounded(state)
    main = torch.cuda.current_stream(x.device)
    if spatial._res_stream is None:
        spatial._res_stream = torch.cuda.Stream(device=x.device)
    side = spatial._res_stream
    side.wait_stream(main)                                  # (x / out / state of the caller)
    cuts = [(T * j) // pieces for j in range(pieces + 1)]
    for j in range(pieces):
        t0, t1 = cuts[j], cuts[j + 1]
        with torch.cuda.stream(side):
            sums = torch.empty(t1 - t0, d_h, dtype=torch.float32, device=x.device) if want_sums else None
            reservoir.encode_into(x[t0:t1], out[t0:t1, :, :d_h], state, col_sums=sums)
            ready = torch.cuda.Event()
            ready.record(side)
        main.wait_event(ready)
        if sums is not None:
            sums.record_stream(main)
        spatial.encode_into(out[t0:t1], d_h, col_sums=sums, x_bound=bound)
    for t in (x, out, state):
        t.record_stream(side)
    if bound is not None:
        hip.mark_unit_bounded(state)                        # (carried on to the caller's next time chunk)
    return out


def choose_numbering(ops_global: List[ShiftOperator], world_size, balance="nnz", locality="auto"):
    """Row bounds of the ranks and, for numberings without locality, the renumbering that gives the cut compact halos:
    ``(ops_global or their permuted forms, bounds, node_order or None)`` (see ``make_partitioned_spatial``)."""
    n = ops_global[0].num_nodes

    def cut(op_list):
        return partition_bounds(n, world_size, op_list[0].rowptr.numpy() if balance == "nnz" else None)

    bounds, node_order = cut(ops_global), None
    if world_size > 1 and locality != "never" and n >= 2 * world_size:
        mid = world_size // 2
        own = max(1, bounds[mid + 1] - bounds[mid])
        plain = halo_rows(ops_global[0], bounds, mid)
        if locality == "always" or 2 * plain > own:
            from .graph import locality_order
            fwd = ops_global[0]
            order = torch.from_numpy(np.ascontiguousarray(
                locality_order(fwd.rowptr.numpy(), fwd.col.numpy(), n))).long()
            cand = [permute_operator(op, order) for op in ops_global]
            cb = cut(cand)
            if locality == "always" or 10 * halo_rows(cand[0], cb, mid) < 7 * plain:
                ops_global, bounds, node_order = cand, cb, order
                if locality == "auto":
                    import warnings
                    warnings.warn("make_partitioned_spatial: the node numbering has no locality; rank r owns "
                                  "spatial.node_order[bounds[r]:bounds[r+1]], not the contiguous range "
                                  "(pass locality='never' to keep the numbering)", stacklevel=3)
    return ops_global, bounds, node_order


@dataclass
class PartitionPlan:
    """Everything the ranks of a node partition need, cut ONCE (``plan_partition``): row bounds, the renumbering (or
    None), the global operators' infinity norms and every rank's local blocks (one per direction)."""
    bounds: List[int]
    node_order: Optional[torch.Tensor]
    norm_inf: List[float]
    n_total: int
    rank_blocks: List[List[LocalBlock]]


def plan_partition(ops_global: List[ShiftOperator], world_size, balance="nnz", locality="auto", exchange="auto"):
    """Cut the global operators for ALL ranks in one process (``multigpu.encode_multi_gpu`` does it in the parent and
    hands every rank its own blocks: the graph preparation, the locality order and the halo lists are computed once
    instead of once per rank)."""
    ops_global, bounds, node_order = choose_numbering(ops_global, world_size, balance, locality)
    rank_blocks = [[] for _ in range(world_size)]
    for op in ops_global:
        halos = [_halo_of(op, bounds[p], bounds[p + 1])[0] for p in range(world_size)]
        for r in range(world_size):
            rank_blocks[r].append(split_operator(op, bounds, r, exchange=exchange, halos=halos))
    return PartitionPlan([int(b) for b in bounds], node_order, [op.norm_inf() for op in ops_global],
                         ops_global[0].num_nodes, rank_blocks)


def spatial_from_plan(plan: PartitionPlan, rank, receptive_field, global_attr, group=None, ops=HipOps, n_chunks=4,
                      force_collectives=False):
    """This rank's ``PartitionedSpatial`` from a ``PartitionPlan`` cut elsewhere."""
    spatial = PartitionedSpatial(plan.rank_blocks[rank], receptive_field, global_attr, plan.n_total, group, ops,
                                 n_chunks=n_chunks, force_collectives=force_collectives)
    spatial.node_order = plan.node_order
    spatial.norm_inf = list(plan.norm_inf)
    return spatial


def make_partitioned_spatial(ops_global: List[ShiftOperator], receptive_field, global_attr,
                             rank=None, world_size=None, group=None, ops=HipOps,
                             balance="nnz", n_chunks=4, force_collectives=False, locality="auto",
                             exchange="auto"):
    """Split the forward (and backward) global operators for this rank.

    Returns ``(spatial, bounds)``; ``spatial.node_order`` is None when rank r owns the global
    nodes ``bounds[r] .. bounds[r+1]``, else an int64 tensor and rank r owns
    ``node_order[bounds[r]:bounds[r+1]]`` (in that order: row i of the rank's tensors is global
    node ``node_order[bounds[r] + i]``).  ``locality``: "auto" renumbers the nodes by
    ``graph.locality_order`` when a contiguous cut of the given numbering would make a rank fetch
    more than half as many halo rows as it owns (a k-NN graph of stations in file order:
    near-full exchange) and the renumbering fetches at least 30 % fewer; "never" keeps the
    numbering; "always" renumbers.
    ``balance``: "nnz" cuts equal edge counts (equal SpMM work), "rows" equal row counts.
    ``exchange``: see ``split_operator`` ("auto": all_gather of full shards instead of the packed
    all_to_all when the ranks together reference more than half of all remote rows)."""
    rank = dist.get_rank(group) if rank is None else rank
    world_size = dist.get_world_size(group) if world_size is None else world_size
    n = ops_global[0].num_nodes
    ops_global, bounds, node_order = choose_numbering(ops_global, world_size, balance, locality)
    blocks = [split_operator(op, bounds, rank, exchange=exchange) for op in ops_global]
    spatial = PartitionedSpatial(blocks, receptive_field, global_attr, n, group, ops,
                                 n_chunks=n_chunks, force_collectives=force_collectives)
    spatial.node_order = node_order
    spatial.norm_inf = [op.norm_inf() for op in ops_global]
    return spatial, bounds
