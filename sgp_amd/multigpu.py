"""Drop-in multi-GPU entry of the encoder: ``SGPEncoder.forward(..., gpus=N)`` / ``encode_dataset(..., gpus=N)``.

The reference's drivers build ONE encoder in ONE process and call it once on host tensors
(``lib/utils.py:27-31``, ``experiments/run_largescale_sgp.py:214-220``).  With ``gpus=N`` (default: the
environment variable ``SGP_AMD_GPUS``, else 1) that same call starts N ranks -- one process per GPU,
``torch.multiprocessing.spawn``, rendezvous on 127.0.0.1 -- and each rank

* rebuilds the caller's encoder from ``encoder.describe()`` (same constructor arguments, same weights),
* cuts the graph with ``partition.make_partitioned_spatial`` (contiguous equal-nnz node blocks, locality
  renumbering where the numbering has none, packed all_to_all / all_gather halo exchange per hop over RCCL),
* encodes its node block with ``partition.encode_partitioned`` in time chunks that fit its device, the
  reservoir state carried on the device, and
* writes its rows of every chunk into ONE shared-memory host tensor ``[T, N, D_out]`` in the ORIGINAL node
  order (or, with ``shard_dir``, into its own ``.pt`` shard files: embeddings larger than host RAM).

The caller gets that tensor back, exactly as from the single-GPU call.  When the box shows fewer GPUs than
ranks the ranks share devices over gloo (functional check on a one-GPU box; no scaling meaning).
"""
import os
import socket

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
    """Time steps per pass of one rank: input + embedding chunk (+ as much again for halo buffers and
    plans) inside ``budget_bytes``; at least ``floor`` steps."""
    per_step = n_own * (f_in + d_out) * 4
    return int(max(min(T, floor), min(T, budget_bytes // max(1, 2 * per_step))))


def _rank_main(rank, world, port, desc, x, edge_index, edge_weight, out, shard_dir, backend, budget, q):
    """One rank (spawned).  ``x`` / ``out``: shared-memory host tensors of the whole problem."""
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
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        enc = SGPEncoder(**desc["kwargs"])
        enc.load_state_dict(desc["state_dict"])
        T, N, F = x.shape
        ops = enc.sgp_encoder.operators(N, edge_index, edge_weight)
        spatial, bounds = partition.make_partitioned_spatial(ops, enc.sgp_encoder.receptive_field,
                                                             enc.sgp_encoder.global_attr)
        rows, n_own = rank_rows(bounds, spatial.node_order, rank)
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
        state._sgp_unit_bounded = True                          # starts at zero (SGPEncoder._state_bound)
        buf = torch.empty(tc, n_own, d_out, dtype=torch.float32, device=dev)
        shards = []
        for t0 in range(0, T, tc):
            n = min(tc, T - t0)
            xs = x[t0:t0 + n][:, rows].to(dev, non_blocking=False).float().contiguous()
            oc = buf[:n]
            partition.encode_partitioned(enc.reservoir, spatial, xs, oc, state)
            torch.cuda.synchronize(dev)
            if shard_dir is not None:
                path = os.path.join(shard_dir, f"embedding_r{rank:02d}_t{t0:08d}.pt")
                torch.save(dict(t0=t0, steps=n, rank=rank, rows=rows if not isinstance(rows, slice)
                                else torch.arange(rows.start, rows.stop), embedding=oc.cpu()), path)
                shards.append(path)
            elif isinstance(rows, slice):
                out[t0:t0 + n, rows] = oc.cpu()
            else:
                out[t0:t0 + n].index_copy_(1, rows, oc.cpu())
        dist.barrier()
        if rank == 0:
            blk = spatial.blocks[0]
            q.put(dict(bounds=[int(b) for b in bounds], t_chunk=tc, backend=backend, world=world,
                       reordered=spatial.node_order is not None,
                       kernel=blk.op.resolved_kernel(), halo_rows=int(blk.n_halo)))
        if shard_dir is not None:
            q.put(dict(rank=rank, shards=shards))
    finally:
        dist.destroy_process_group()


def encode_multi_gpu(encoder, x, edge_index, edge_weight, gpus, out=None, shard_dir=None, backend=None,
                     device_budget_bytes=None, info=None):
    """Host tensor ``x[T, N, F]`` -> host tensor ``[T, N, D_out]`` (original node order) computed by ``gpus``
    ranks, or -- with ``shard_dir`` -- the list of shard files the ranks wrote (each a dict ``t0, steps,
    rank, rows, embedding[steps, len(rows), D_out]``) and no host tensor at all.  ``info``: a dict that
    receives what rank 0 reports (bounds, time chunk, backend, hop kernel)."""
    import torch.multiprocessing as mp
    from . import hip
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
    xs = x.detach().float().contiguous()
    if not xs.is_shared():
        xs = xs.clone().share_memory_()
    if shard_dir is None:
        if out is None:
            out = torch.empty(T, N, d_out, dtype=torch.float32)
        elif tuple(out.shape) != (T, N, d_out) or out.dtype != torch.float32 or not out.is_contiguous() or out.is_cuda:
            raise ValueError(f"out must be a contiguous float32 host tensor of shape {(T, N, d_out)}")
        shared = out if out.is_shared() else out.share_memory_()
    else:
        os.makedirs(shard_dir, exist_ok=True)
        shared = None
    ei = torch.as_tensor(edge_index).cpu() if not hasattr(edge_index, "csr") else edge_index
    ew = None if edge_weight is None else torch.as_tensor(edge_weight).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    mp.spawn(_rank_main, args=(world, port, encoder.describe(), xs, ei, ew, shared, shard_dir, backend,
                               device_budget_bytes, q), nprocs=world, join=True)
    report, shard_lists = {}, {}
    while not q.empty():
        m = q.get()
        if "shards" in m:
            shard_lists[m["rank"]] = m["shards"]
        else:
            report = m
    if info is not None:
        info.update(report)
    if shard_dir is not None:
        return [p for r in sorted(shard_lists) for p in shard_lists[r]]
    return shared
