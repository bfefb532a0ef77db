"""Drop-in multi-GPU entry of the encoder: ``SGPEncoder.forward(..., gpus=N)`` / ``encode_dataset(..., gpus=N)``.

The reference's drivers build ONE encoder in ONE process and call it once on host tensors
(``lib/utils.py:27-31``, ``experiments/run_largescale_sgp.py:214-220``).  With ``gpus=N`` (default: the
environment variable ``SGP_AMD_GPUS``, else 1) that same call

* prepares the graph and cuts the node partition ONCE, in the calling process (``partition.plan_partition``:
  contiguous equal-nnz node blocks, locality renumbering where the numbering has none, every rank's local blocks and
  halo lists) and hands every rank only its own blocks,
* maps the result ``[T, N, D_out]`` as ONE shared file-backed tensor (``torch.from_file(shared=True)`` in
  ``shm_dir``, default /dev/shm, unlinked once mapped): no copy of a 100 GB embedding through a second buffer -- the
  ranks write disjoint rows of it, the caller gets that very mapping back,
* moves the caller's input into shared memory in place (one copy, no clone) and
* starts N ranks -- one process per GPU, ``torch.multiprocessing.spawn``, rendezvous on 127.0.0.1 -- that rebuild the
  caller's encoder from ``encoder.describe()`` (constructor arguments, per-layer leaking rates, weights) and run
  ``RankPipeline``: per time chunk a host gather of the rank's rows into a pinned slot, an asynchronous H2D,
  ``partition.encode_partitioned`` (reservoir of piece c + 1 under the hops + halo exchange of piece c; packed
  all_to_all / all_gather over RCCL), an asynchronous D2H into a pinned slot and a host scatter into the shared
  result -- two slots each way, so the transfers and the host copies of chunks i - 1 and i + 1 run under the
  encoding of chunk i, and nothing on the compute stream waits for the host.

With ``shard_dir`` the ranks write ``.pt`` shard files instead (embeddings larger than host RAM).  When the box shows
fewer GPUs than ranks the ranks share devices over gloo (functional check on a one-GPU box; no scaling meaning).
/dev/shm (or ``shm_dir``) must hold input + result: ``require_shm_space`` checks that BEFORE anything is mapped (a tmpfs
over-commit surfaces as SIGBUS in the middle of a run otherwise); the result's file name is removed as soon as every rank
has mapped it (a killed parent leaks nothing); the process group gets an explicit timeout (``SGP_AMD_DIST_TIMEOUT``
seconds, default 1800).  No hardware scaling curve of this path exists yet (the pool's leases have one GPU: DESIGN.md 5).
"""
import datetime
import json
import os
import socket
import tempfile

import torch


def resolve_gpus(gpus=None):
    """``gpus`` argument -> rank count: None reads SGP_AMD_GPUS (default 1); 0 / 'all' = every visible GPU."""
    if gpus is None:
        gpus = os.environ.get("SGP_AMD_GPUS", "1")
    if isinstance(gpus, str):
        gpus = 0 if gpus.strip().lower() in ("all", "0", "") else int(gpus)
    gpus = int(gpus)
    if gpus < 0:
        raise ValueError(f"gpus must be >= 0, got {gpus}")
    if gpus == 0:
        gpus = max(1, torch.cuda.device_count())
    return gpus


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_rows(bounds, node_order, rank):
    """Global node ids of rank ``rank``'s rows, in the rank's row order (slice when contiguous)."""
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    if node_order is None:
        return slice(lo, hi), hi - lo
    return node_order[lo:hi].clone(), hi - lo


def chunk_steps(T, n_own, f_in, d_out, budget_bytes, floor=8):
    """Time steps per pass of one rank: two input + two embedding slots (+ as much again for halo buffers and
    plans) inside ``budget_bytes``; at least ``floor`` steps."""
    per_step = n_own * (f_in + d_out) * 4
    return int(max(min(T, floor), min(T, budget_bytes // max(1, 4 * per_step))))


class RankPipeline:
    """Time-chunk pipeline of one rank: host rows -> pinned slot -> device -> ``encode(xs, oc)`` -> pinned slot ->
    ``sink(t0, n, rows_tensor)``.  Two slots each way; H2D and D2H on their own streams; the host gathers chunk i + 1
    and scatters chunk i - 1 while the device encodes chunk i (``encode`` only enqueues: nothing on the compute stream
    waits for the host).  ``events``: a list that receives per chunk ``(compute start, compute end, d2h end)`` timing
    events (tests: the D2H of chunk i ends after the compute of chunk i + 1 has started)."""

    def __init__(self, dev, tc, n_own, f_in, d_out, nbuf=2):
        self.dev, self.tc, self.nbuf = dev, tc, nbuf
        self.xin = [torch.empty(tc, n_own, f_in, dtype=torch.float32, device=dev) for _ in range(nbuf)]
        self.buf = [torch.empty(tc, n_own, d_out, dtype=torch.float32, device=dev) for _ in range(nbuf)]
        self.pin_in = [torch.empty(tc, n_own, f_in, dtype=torch.float32, pin_memory=True) for _ in range(nbuf)]
        self.pin_out = [torch.empty(tc, n_own, d_out, dtype=torch.float32, pin_memory=True) for _ in range(nbuf)]
        self.h2d, self.d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def run(self, x, rows, T, encode, sink, events=None):
        tc, nbuf, dev = self.tc, self.nbuf, self.dev
        main = torch.cuda.current_stream(dev)
        starts = list(range(0, T, tc))
        timing = events is not None
        ev_h2d = [None] * nbuf            # the input slot holds its chunk
        ev_done = [None] * nbuf           # compute of the slot's chunk finished
        ev_d2h = [None] * nbuf            # the chunk has left buf[slot]

        def stage_in(i):
            s, t0 = i % nbuf, starts[i]
            n = min(tc, T - t0)
            if ev_h2d[s] is not None:
                ev_h2d[s].synchronize()                          # the slot's previous H2D has read the pinned rows
            src = x[t0:t0 + n]
            if isinstance(rows, slice):
                self.pin_in[s][:n].copy_(src[:, rows])            # host gather (+ dtype cast)
            else:
                torch.index_select(src if src.dtype == torch.float32 else src.float(), 1, rows, out=self.pin_in[s][:n])
            with torch.cuda.stream(self.h2d):
                if ev_done[s] is not None:
                    self.h2d.wait_event(ev_done[s])               # the chunk that used xin[s] is encoded
                self.xin[s][:n].copy_(self.pin_in[s][:n], non_blocking=True)
                ev_h2d[s] = torch.cuda.Event()
                ev_h2d[s].record(self.h2d)

        def drain(i):
            s, t0 = i % nbuf, starts[i]
            ev_d2h[s].synchronize()                              # (the device is busy with the next chunk meanwhile)
            sink(t0, min(tc, T - t0), self.pin_out[s][:min(tc, T - t0)])

        if not starts:
            return
        stage_in(0)
        for i, t0 in enumerate(starts):
            s = i % nbuf
            n = min(tc, T - t0)
            if i + 1 < len(starts):
                stage_in(i + 1)
            main.wait_event(ev_h2d[s])
            if ev_d2h[s] is not None:
                drain(i - nbuf)                                  # chunk i - nbuf leaves pin_out[s] ...
                main.wait_event(ev_d2h[s])                       # ... and has left buf[s]
            if timing:
                c0 = torch.cuda.Event(enable_timing=True)
                c0.record(main)
            encode(self.xin[s][:n], self.buf[s][:n])
            ev_done[s] = torch.cuda.Event(enable_timing=timing)
            ev_done[s].record(main)
            with torch.cuda.stream(self.d2h):
                self.d2h.wait_event(ev_done[s])
                self.pin_out[s][:n].copy_(self.buf[s][:n], non_blocking=True)
                ev_d2h[s] = torch.cuda.Event(enable_timing=timing)
                ev_d2h[s].record(self.d2h)
            if timing:
                events.append((c0, ev_done[s], ev_d2h[s]))
        for i in range(max(0, len(starts) - nbuf), len(starts)):
            drain(i)
        main.wait_stream(self.h2d)
        main.wait_stream(self.d2h)


def dist_timeout():
    """Timeout of the ranks' process group: generous (the first collective waits for the slowest rank's graph plans and
    RCCL's own bring-up over xGMI), explicit rather than the backend's default."""
    return datetime.timedelta(seconds=int(os.environ.get("SGP_AMD_DIST_TIMEOUT", "1800")))


def require_shm_space(shm_dir, needed_bytes, what):
    """Fail BEFORE mapping when ``shm_dir`` cannot hold ``needed_bytes`` (tmpfs hands out pages lazily: an over-committed
    mapping dies with SIGBUS when the ranks touch it)."""
    st = os.statvfs(shm_dir)
    free = st.f_bavail * st.f_frsize
    if free < needed_bytes:
        raise RuntimeError(
            f"sgp_amd multi-GPU encode: {shm_dir} has {free / 2 ** 30:.1f} GiB free but {what} needs "
            f"{needed_bytes / 2 ** 30:.1f} GiB ({needed_bytes} bytes); pass shm_dir= (a larger tmpfs / a fast disk) or "
            f"shard_dir= (the ranks then write shard files and no host tensor is built)")
    return free


def _rank_main(rank, world, port, desc, x, out_file, out_shape, plan_dir, shard_dir, backend, budget):
    """One rank (spawned).  ``x``: shared-memory host tensor of the whole input; ``out_file``: the shared result's
    backing file (None with ``shard_dir``); ``plan_dir``: where the parent left this rank's partition blocks and
    where the rank leaves its report."""
    import torch.distributed as dist
    from . import hip, partition
    from .nn.encoders.sgp_encoder import SGPEncoder
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    hip.require_gpu()
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(rank % n_dev)
    dev = torch.device("cuda", rank % n_dev)
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=dist_timeout())
    try:
        enc = SGPEncoder(**desc["kwargs"])
        enc.load_state_dict(desc["state_dict"])
        for layer, alpha in zip(enc.reservoir.reservoir_layers, desc.get("alphas", [])):
            layer.alpha = float(alpha)                           # (per-layer rates a caller may have edited)
        T, N, F = x.shape
        meta = torch.load(os.path.join(plan_dir, "meta.pt"), weights_only=False)
        blocks = torch.load(os.path.join(plan_dir, f"blocks_r{rank:02d}.pt"), weights_only=False)
        plan = partition.PartitionPlan(meta["bounds"], meta["node_order"], meta["norm_inf"], N,
                                       [blocks if r == rank else None for r in range(world)])
        spatial = partition.spatial_from_plan(plan, rank, enc.sgp_encoder.receptive_field, enc.sgp_encoder.global_attr)
        rows, n_own = rank_rows(plan.bounds, plan.node_order, rank)
        d_out = enc.output_size
        if budget is None:
            free, _ = torch.cuda.mem_get_info()
            budget = int(0.6 * free / max(1, -(-world // n_dev)))       # ranks sharing a device share its memory
        tc = chunk_steps(T, n_own, F, d_out, budget)
        # every rank must cut the time axis alike (the halo exchange is collective)
        tcs = torch.tensor([tc], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tcs, op=dist.ReduceOp.MIN)
        tc = int(tcs.item())
        L, R = len(enc.reservoir.reservoir_layers), enc.reservoir.hidden_size
        state = torch.zeros(L, n_own, R, dtype=torch.float32, device=dev)
        hip.mark_unit_bounded(state)                          # starts at zero (SGPEncoder._state_bound)
        out = None
        if out_file is not None:
            numel = out_shape[0] * out_shape[1] * out_shape[2]
            out = torch.from_file(out_file, shared=True, size=max(numel, 1), dtype=torch.float32)[:numel].view(out_shape)
            # every rank (and the parent) holds its mapping now: the NAME can go -- nothing is left behind in shm_dir
            # if the parent is killed from here on
            dist.barrier()
            if rank == 0:
                try:
                    os.unlink(out_file)
                except OSError:
                    pass
        shards = []
        row_ids = rows if not isinstance(rows, slice) else torch.arange(rows.start, rows.stop)

        def encode(xs, oc):
            partition.encode_partitioned(enc.reservoir, spatial, xs, oc, state)

        def sink(t0, n, emb):
            if shard_dir is not None:
                path = os.path.join(shard_dir, f"embedding_r{rank:02d}_t{t0:08d}.pt")
                torch.save(dict(t0=t0, steps=n, rank=rank, rows=row_ids, embedding=emb.clone()), path)
                shards.append(path)
            elif isinstance(rows, slice):
                out[t0:t0 + n, rows] = emb
            else:
                out[t0:t0 + n].index_copy_(1, rows, emb)

        RankPipeline(dev, tc, n_own, F, d_out).run(x, rows, T, encode, sink)
        torch.cuda.synchronize(dev)
        dist.barrier()
        blk = spatial.blocks[0]
        report = dict(rank=rank, shards=shards)
        if rank == 0:
            report.update(bounds=[int(b) for b in plan.bounds], t_chunk=tc, backend=backend, world=world,
                          reordered=plan.node_order is not None, kernel=blk.op.resolved_kernel(),
                          halo_rows=int(blk.n_halo))
        with open(os.path.join(plan_dir, f"report_r{rank:02d}.json"), "w") as f:      # (a file, not a pipe: no size limit)
            json.dump(report, f)
    finally:
        dist.destroy_process_group()


def default_shm_dir(shm_dir=None):
    return shm_dir or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())


def shared_result(shape, shm_dir=None, extra_bytes=0):
    """A float32 host tensor of ``shape`` backed by a fresh file in ``shm_dir`` (default /dev/shm) that other
    processes can map by name: ``(tensor, path)``.  Rank 0 unlinks the path once every rank has mapped it.
    ``extra_bytes``: what else the caller is about to put into the same filesystem (the shared input)."""
    shm_dir = default_shm_dir(shm_dir)
    n = 1
    for d in shape:
        n *= int(d)
    require_shm_space(shm_dir, 4 * n + extra_bytes, f"the shared result {tuple(shape)} float32"
                      + (f" + {extra_bytes} bytes of shared input" if extra_bytes else ""))
    fd, path = tempfile.mkstemp(prefix="sgp_amd_out_", suffix=".bin", dir=shm_dir)
    os.close(fd)
    t = torch.from_file(path, shared=True, size=max(n, 1), dtype=torch.float32)
    return t[:n].view(*shape), path


def encode_multi_gpu(encoder, x, edge_index, edge_weight, gpus, out=None, shard_dir=None, backend=None,
                     device_budget_bytes=None, info=None, shm_dir=None):
    """Host tensor ``x[T, N, F]`` -> host tensor ``[T, N, D_out]`` (original node order) computed by ``gpus``
    ranks, or -- with ``shard_dir`` -- the list of shard files the ranks wrote (each a dict ``t0, steps,
    rank, rows, embedding[steps, len(rows), D_out]``) and no host tensor at all.  ``info``: a dict that
    receives what rank 0 reports (bounds, time chunk, backend, hop kernel).  ``out``: a caller's tensor to fill
    (one host copy at the end; without it the shared mapping itself is returned)."""
    import shutil
    import torch.multiprocessing as mp
    from . import hip, partition
    hip.require_gpu()
    if not hasattr(encoder, "describe") or type(encoder).__name__ != "SGPEncoder":
        raise NotImplementedError("gpus > 1 serves SGPEncoder (the node-partitioned path of SURVEY.md 8e)")
    if x.is_cuda:
        raise ValueError("gpus > 1 takes the host tensor the reference's drivers hand over (lib/utils.py:24-31)")
    if x.dim() != 3:
        raise ValueError("x must be [T, N, F]")
    world = int(gpus)
    n_dev = torch.cuda.device_count()
    if backend is None:
        backend = "nccl" if n_dev >= world else "gloo"     # ranks that share a device cannot use RCCL
    T, N, _ = x.shape
    d_out = encoder.output_size
    # the input travels as shared memory: moved there in place when it is a plain float32 tensor (one copy, the
    # caller's tensor stays valid), converted first only when it has to be
    xs = x.detach()
    if xs.dtype != torch.float32 or not xs.is_contiguous():
        xs = xs.float().contiguous()
    in_bytes = 0 if xs.is_shared() else xs.numel() * 4          # torch's shared memory lives in /dev/shm
    out_bytes = 0 if shard_dir is not None else T * N * d_out * 4
    if os.path.isdir("/dev/shm") and in_bytes:
        same_fs = shard_dir is None and os.stat(default_shm_dir(shm_dir)).st_dev == os.stat("/dev/shm").st_dev
        require_shm_space("/dev/shm", in_bytes + (out_bytes if same_fs else 0),
                          "the shared input" + (" + result" if same_fs else ""))
    if not xs.is_shared():
        xs.share_memory_()
    if out is not None and (tuple(out.shape) != (T, N, d_out) or out.dtype != torch.float32 or not out.is_contiguous()
                            or out.is_cuda):
        raise ValueError(f"out must be a contiguous float32 host tensor of shape {(T, N, d_out)}")
    work = tempfile.mkdtemp(prefix="sgp_amd_plan_")
    out_path = None
    try:
        # graph preparation and partition: once, here
        ops = encoder.sgp_encoder.operators(N, edge_index, edge_weight)
        plan = partition.plan_partition(ops, world)
        torch.save(dict(bounds=plan.bounds, node_order=plan.node_order, norm_inf=plan.norm_inf),
                   os.path.join(work, "meta.pt"))
        for r in range(world):
            torch.save(plan.rank_blocks[r], os.path.join(work, f"blocks_r{r:02d}.pt"))
        shared = None
        if shard_dir is None:
            shared, out_path = shared_result((T, N, d_out), shm_dir)      # (checks the free space of shm_dir again)
        else:
            os.makedirs(shard_dir, exist_ok=True)
        mp.spawn(_rank_main, args=(world, free_port(), encoder.describe(), xs, out_path, (T, N, d_out), work,
                                   shard_dir, backend, device_budget_bytes), nprocs=world, join=True)
        report, shard_lists = {}, {}
        for r in range(world):
            with open(os.path.join(work, f"report_r{r:02d}.json")) as f:
                m = json.load(f)
            shard_lists[r] = m.pop("shards")
            if r == 0:
                report = m
        if info is not None:
            info.update(report)
        if shard_dir is not None:
            return [p for r in sorted(shard_lists) for p in shard_lists[r]]
        if out is not None:
            out.copy_(shared)
            return out
        return shared
    finally:
        shutil.rmtree(work, ignore_errors=True)
        if out_path is not None and os.path.exists(out_path):
            os.unlink(out_path)                               # the mapping (and the caller's tensor) outlives the name
