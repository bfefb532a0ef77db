import torch
from torch import nn

from ... import hip, tune
from ..reservoir import Reservoir
from ._args import add_reservoir_args, add_spatial_args
from .sgp_spatial_encoder import SGPSpatialEncoder


class _RegisteredSink:
    """Destination of the pipelined host path: an ORDINARY (pageable) host tensor whose pages are
    registered with the HIP runtime block by block in a helper thread, so that the D2H copies go
    straight into it at PCIe speed -- no pinned bounce buffer, no host memcpy.  What remains is
    the kernel's first-touch cost of fresh pages (11-13 GB/s measured on the MI355X host,
    tools/probe_hostmem.py): the floor of ANY way of producing into new host memory.  The
    registration is dropped when the encode is done; the tensor is then plain memory again
    (the reference's drivers fork DataLoader workers that inherit it copy-on-write)."""
    BLOCK = 256 << 20

    def __init__(self, out):
        import threading
        self.out = out
        self.rt = torch.cuda.cudart()
        page = 4096
        lo = out.data_ptr() // page * page
        hi = -(-(out.data_ptr() + out.numel() * out.element_size()) // page) * page
        self.base, self.blocks = out.data_ptr(), []
        self.todo = [(a, min(self.BLOCK, hi - a)) for a in range(lo, hi, self.BLOCK)]
        self.done_bytes = 0                     # bytes of ``out`` (from its start) that are registered
        self.failed = False
        self.cv = threading.Condition()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        for addr, size in self.todo:
            ok = False
            try:
                ok = int(self.rt.cudaHostRegister(addr, size, 0)) == 0
            except Exception:
                ok = False
            with self.cv:
                if ok:
                    self.blocks.append(addr)
                    self.done_bytes = addr + size - self.base
                else:
                    self.failed = True
                self.cv.notify_all()
            if not ok:
                return

    def wait(self, end_byte):
        """True once bytes [0, end_byte) of the tensor are registered; False if registration is
        not available (the caller then goes through its pinned slot)."""
        with self.cv:
            while self.done_bytes < end_byte and not self.failed:
                self.cv.wait()
            return self.done_bytes >= end_byte

    def pieces(self, b0, b1):
        """[b0, b1) (bytes from the tensor's start) cut at the block boundaries: one asynchronous
        copy must stay inside ONE registered range."""
        first = self.todo[0][0] - self.base                 # <= 0: start of block 0
        cuts = [b0]
        k = (b0 - first) // self.BLOCK + 1
        while first + k * self.BLOCK < b1:
            cuts.append(first + k * self.BLOCK)
            k += 1
        cuts.append(b1)
        return list(zip(cuts[:-1], cuts[1:]))

    def close(self):
        self.thread.join()
        for addr in self.blocks:
            try:
                self.rt.cudaHostUnregister(addr)
            except Exception:
                pass
        self.blocks = []


class SGPEncoder(nn.Module):
    """SGP's training-free spatiotemporal encoder, ``lib/nn/encoders/sgp_encoder.py:9-51``:
    reservoir over time, then K-hop graph-shift propagation over nodes.

    The device path allocates the final ``[T, N, D_out]`` tensor once; the reservoir
    writes its states straight into block 0, every hop reads one block and writes the next
    and the global-mean block is filled last -- the ``stack``/``rearrange``/``cat`` copies of
    the reference (reservoir.py:178-183, sgp_spatial_encoder.py:35) never happen.
    """

    def __init__(self,
                 input_size,
                 reservoir_size,
                 reservoir_layers,
                 leaking_rate,
                 spectral_radius,
                 density,
                 input_scaling,
                 receptive_field,
                 bidirectional,
                 alpha_decay,
                 global_attr,
                 add_self_loops=False,
                 undirected=False,
                 reservoir_activation='tanh'):
        super(SGPEncoder, self).__init__()
        self.reservoir = Reservoir(input_size=input_size,
                                   hidden_size=reservoir_size,
                                   input_scaling=input_scaling,
                                   num_layers=reservoir_layers,
                                   leaking_rate=leaking_rate,
                                   spectral_radius=spectral_radius,
                                   density=density,
                                   activation=reservoir_activation,
                                   alpha_decay=alpha_decay)
        self.sgp_encoder = SGPSpatialEncoder(
            receptive_field=receptive_field,
            bidirectional=bidirectional,
            undirected=undirected,
            add_self_loops=add_self_loops,
            global_attr=global_attr)
        self._side_streams = {}

    @property
    def output_size(self):
        return self.sgp_encoder.num_blocks() * self.reservoir.output_size

    # Small graphs: the reservoir is a serial chain of T steps on ceil(N / 16) CUs (PEMS-BAY: 21 of
    # 256) and the hops are bandwidth-bound on all of them.  Up to `overlap_tiles` node tiles, and when
    # the hops are worth it (their estimated time >= a tenth of the chain's), the time axis is cut
    # into up to `overlap_chunks` pieces and the hops (+ global mean) of piece i run on a second stream under
    # the reservoir of piece i + 1 (state carried on the device: bit-identical to one pass).
    # PEMS-BAY shape: 107 -> 93 ms per pass in round 2 (the chain runs ~13 % slower beside the hops -- the clock under
    # load, not its neighbours: its own compute units and no row wait changed nothing).
    # Round 5 (chain 0.76 us per step, hops 33 of the 39 ms beside it): 8 pieces 49.0 ms, 12: 45.2, 16: 45.0, 24: 48.3 -- the tail is the last piece's hops.
    overlap_tiles = 128
    overlap_chunks = 16
    overlap_chunks_time_parallel = 1   # time chunks of that pipeline when the reservoir runs as time pieces (SGP_TUNE=overlap_chunks_tp)
    overlap_masked_tiles = 32      # the chain gets its own compute units up to this many node tiles (hip.cu_masked_streams)

    def _overlap_pieces(self, T, N):
        if (N + 15) // 16 > self.overlap_tiles:
            return 1
        L, R = len(self.reservoir.reservoir_layers), self.reservoir.hidden_size
        chain_us = (0.27 + 3.0e-5 * R * R) * L                # per step: 0.39 us at R = 64, 0.76 at 128 (measured)
        hop_us = (self.sgp_encoder.num_blocks() - 1) * N * L * R * 8 / 4e6   # bytes of the hop blocks at ~4 TB/s
        share = hop_us / chain_us
        # hops as long as the chain (PEMS-BAY shape, 0.9): 16 pieces; a tenth of it (METR-LA shape, 0.14): 4 pieces take
        # 14.2 ms against 14.8 in one piece, 14.5 in 8 and 15.0 in 16 -- every piece costs a launch gap of the chain
        pieces = self.overlap_chunks if share >= 0.5 else 8 if share >= 0.25 else 4 if share >= 0.1 else 1
        pieces = tune.get("overlap_chunks", pieces, int)
        while pieces > 1 and T < 64 * pieces:                  # a sequence too short for the count takes the next smaller one
            pieces //= 2
        return max(1, pieces)

    def _state_bound(self, state=None):
        """Upper bound of |reservoir state| where one holds: a leaky average ``(1 - a) h + a act(.)`` of values in
        [-1, 1] stays in [-1, 1] (tanh; self_norm rows have unit 2-norm) PROVIDED every layer's leaking rate lies in
        [0, 1] (the reference accepts any float, reservoir.py:109-123) and the recurrence starts inside the
        interval -- from zero, or from a state this encoder produced itself under the same premises (marked
        ``_sgp_unit_bounded`` by ``encode_device``).  Everything else -- relu / identity, a leaking rate outside
        [0, 1], a state handed in by the caller -- returns None: the hop measures its operand."""
        if self.reservoir.mode not in ("tanh", "self_norm"):
            return None
        if not all(0.0 <= float(l.alpha) <= 1.0 for l in self.reservoir.reservoir_layers):
            return None
        if state is not None and not hip.is_unit_bounded(state):
            return None
        return 1.0

    def encode_device(self, x, ops, out=None, state=None, timeline=None):
        """x[T, N, F] CUDA float32 -> out[T, N, D_out] on the same device.  ``state`` [L, N, R]:
        reservoir state carried across calls (updated in place); ``timeline``: see
        ``sgp_preprocessing.propagate_into``."""
        T, N, _ = x.shape
        d_h = self.reservoir.output_size
        if out is None:
            out = torch.empty(T, N, self.output_size, dtype=torch.float32, device=x.device)
        chunks = self._overlap_pieces(T, N)
        time_parallel = chunks > 1 and self.reservoir.time_parallel(T, N, x.device)
        if time_parallel:
            # the reservoir runs as time pieces on the whole chip: few, long time chunks (the chain of chunk c + 1 under
            # the hops of chunk c), ordinary streams -- the chain is no longer a handful of workgroups to fence off
            chunks = max(1, min(tune.get("overlap_chunks_tp", self.overlap_chunks_time_parallel, int), chunks))
            while chunks > 1 and not self.reservoir.time_parallel(T // chunks, N, x.device):
                chunks //= 2
        x_bound = self._state_bound(state)
        # global_attr: the column sums of the states come from the reservoir kernel where it has them
        # in registers (fused stacked kernel); the other kernels keep the fused mean + broadcast pass
        want_sums = self.sgp_encoder.global_attr and self.reservoir.produces_col_sums(x)
        if chunks <= 1:
            sums = torch.empty(T, d_h, dtype=torch.float32, device=x.device) if want_sums else None
            self.reservoir.encode_into(x, out[:, :, :d_h], state, col_sums=sums)
            self.sgp_encoder.encode_into(out, d_h, ops, timeline, col_sums=sums, x_bound=x_bound)
            if state is not None and x_bound is not None:
                hip.mark_unit_bounded(state)                  # (carried on: encode_streamed, the time pieces below)
            return out
        if state is None:
            state = torch.zeros(len(self.reservoir.reservoir_layers), N, self.reservoir.hidden_size,
                                dtype=torch.float32, device=x.device)
            hip.mark_unit_bounded(state)
        main = torch.cuda.current_stream(x.device)
        key = str(x.device)
        if key not in self._side_streams:
            self._side_streams[key] = torch.cuda.Stream(device=x.device)
        side, chain = self._side_streams[key], main
        # The chain (one wave per SIMD on ceil(N / 16) CUs) and the hops (every CU) on DISJOINT compute units: the small
        # kernels between two pieces of the chain (weight packing) no longer queue behind hop workgroups (65 -> 5 us
        # each; PEMS-BAY shape 51.2 -> 48.6 ms per pass).  Up to 32 tiles.
        tiles = (N + 15) // 16
        if tiles <= self.overlap_masked_tiles and tune.get("overlap_cu_mask", 1, int) and not time_parallel:
            pair = hip.cu_masked_streams(x.device, (tiles + 7) // 8 * 8)
            if pair is not None:
                chain, side = pair
        side.wait_stream(main)                                  # (out / earlier work of the caller)
        if chain is not main:
            chain.wait_stream(main)
        for j in range(chunks):
            t0, t1 = T * j // chunks, T * (j + 1) // chunks
            with torch.cuda.stream(chain):
                sums = torch.empty(t1 - t0, d_h, dtype=torch.float32, device=x.device) if want_sums else None
                self.reservoir.encode_into(x[t0:t1], out[t0:t1, :, :d_h], state, col_sums=sums)
                ready = torch.cuda.Event()
                ready.record(chain)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                self.sgp_encoder.encode_into(out[t0:t1], d_h, ops, timeline, col_sums=sums,
                                             x_bound=x_bound)
                if sums is not None:
                    sums.record_stream(side)
        main.wait_stream(side)
        if chain is not main:
            main.wait_stream(chain)
        return out

    # Device-memory budget for one pass (bytes); None = 80 % of what is free right now.  Host
    # inputs whose input + embedding exceed it are encoded in time chunks (see encode_streamed).
    max_device_bytes = None

    def _budget(self):
        if self.max_device_bytes is not None:
            return int(self.max_device_bytes)
        free, _ = torch.cuda.mem_get_info()
        return int(0.8 * free)

    # Host inputs larger than this many bytes (input + embedding) take the pipelined path even
    # when they would fit the device: PCIe then runs under the compute instead of after it.
    stream_threshold_bytes = 256 << 20
    stream_chunk_bytes = 1 << 30          # embedding bytes per time chunk of the pipelined path

    def encode_streamed(self, x, ops, t_chunk, out=None):
        """Host tensor x[T, N, F] -> host tensor [T, N, D_out], ``t_chunk`` steps at a time,
        transfers overlapped with the compute (the drivers hand over host tensors,
        lib/utils.py:24-31): two device buffers per direction, H2D of chunk i+1 (through a pinned
        slot) and D2H of chunk i-1 on their own streams while chunk i is encoded; the D2H goes
        straight into the result tensor, whose pages a helper thread registers with the runtime
        ahead of the copies (``_RegisteredSink``; pinned bounce slots + a host memcpy if the
        runtime refuses).  The recurrence is
        carried across chunks in a device-resident state ``[L, N, R]`` and the propagation is
        independent per time step, so the result is bit-identical to a single pass.  This is
        also how embeddings larger than the 288 GB of HBM (BASELINE config C5: 629 GB) or than
        the free memory are produced.  The returned tensor is ordinary (pageable) host memory:
        the reference's drivers fork DataLoader workers that inherit it copy-on-write.

        ``out``: a caller-supplied contiguous float32 host tensor ``[T, N, D_out]`` to fill instead
        of a fresh one.  A FRESH result tensor costs this host a page fault + zeroing per page
        (11-13 GB/s) -- more than the PCIe transfer; a tensor that is re-used between
        calls (or was touched before) skips that, and a pinned one (``pin_memory=True``) is written
        by the D2H copies directly, without the registration thread."""
        hip.require_gpu()
        T, N, F = x.shape
        dev = torch.device("cuda", torch.cuda.current_device())
        L, R = len(self.reservoir.reservoir_layers), self.reservoir.hidden_size
        d_h, D = L * R, self.output_size
        tc = max(1, min(int(t_chunk), T))
        starts = list(range(0, T, tc))
        if out is None:
            out = torch.empty(T, N, D, dtype=torch.float32)
        elif (out.is_cuda or out.dtype != torch.float32 or tuple(out.shape) != (T, N, D)
              or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous float32 host tensor of shape {(T, N, D)}")
        if T == 0:
            return out
        out_pinned = out.is_pinned()
        state = torch.zeros(L, N, R, dtype=torch.float32, device=dev)
        hip.mark_unit_bounded(state)                          # starts at zero (see _state_bound)
        nbuf = 2 if len(starts) > 1 else 1
        xin = [torch.empty(tc, N, F, dtype=torch.float32, device=dev) for _ in range(nbuf)]
        buf = [torch.empty(tc, N, D, dtype=torch.float32, device=dev) for _ in range(nbuf)]
        x_pinned = x.is_pinned() and x.dtype == torch.float32
        pin_in = None if x_pinned else [torch.empty(tc, N, F, dtype=torch.float32, pin_memory=True)
                                        for _ in range(nbuf)]
        sink = _RegisteredSink(out) if self.register_output and not out_pinned else None
        pin_out = [None] * nbuf                                  # pinned bounce slots: only if needed
        main = torch.cuda.current_stream(dev)
        h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ev_h2d = [torch.cuda.Event() for _ in range(nbuf)]      # input slot holds its chunk
        ev_done = [torch.cuda.Event() for _ in range(nbuf)]     # compute of the slot's chunk finished
        ev_d2h = [torch.cuda.Event() for _ in range(nbuf)]      # the chunk has left buf[slot]
        used = [False] * nbuf
        bounced = [False] * nbuf                                 # chunk sits in pin_out[slot], not in out
        row_bytes = N * D * 4
        out_flat = out.view(-1)

        def stage_in(i):
            s, t0 = i % nbuf, starts[i]
            n = min(tc, T - t0)
            if x_pinned:
                src = x[t0:t0 + n]
            else:
                if used[s]:
                    ev_h2d[s].synchronize()                      # the slot's previous H2D has read it
                pin_in[s][:n].copy_(x[t0:t0 + n])                # host memcpy (+ dtype cast)
                src = pin_in[s][:n]
            with torch.cuda.stream(h2d):
                if used[s]:
                    h2d.wait_event(ev_done[s])                   # the chunk that used xin[s] is encoded
                xin[s][:n].copy_(src, non_blocking=True)
                ev_h2d[s].record(h2d)

        def send_out(i):
            """D2H of chunk i: straight into ``out`` once its pages are registered, else through a
            pinned slot that ``drain`` copies out."""
            s, t0 = i % nbuf, starts[i]
            n = min(tc, T - t0)
            direct = out_pinned or (sink is not None and sink.wait((t0 + n) * row_bytes))
            if not direct and pin_out[s] is None:
                pin_out[s] = torch.empty(tc, N, D, dtype=torch.float32, pin_memory=True)
            bounced[s] = not direct
            with torch.cuda.stream(d2h):
                d2h.wait_event(ev_done[s])
                if direct:
                    src = buf[s][:n].reshape(-1)
                    e0 = t0 * (row_bytes // 4)
                    cuts = [(t0 * row_bytes, (t0 + n) * row_bytes)] if out_pinned else \
                        sink.pieces(t0 * row_bytes, (t0 + n) * row_bytes)
                    for a, b in cuts:
                        out_flat[a // 4:b // 4].copy_(src[a // 4 - e0:b // 4 - e0], non_blocking=True)
                else:
                    pin_out[s][:n].copy_(buf[s][:n], non_blocking=True)
                ev_d2h[s].record(d2h)

        def drain(i):
            s, t0 = i % nbuf, starts[i]
            n = min(tc, T - t0)
            if bounced[s]:
                ev_d2h[s].synchronize()
                out[t0:t0 + n].copy_(pin_out[s][:n])             # host memcpy into pageable memory
                bounced[s] = False

        try:
            stage_in(0)
            for i, t0 in enumerate(starts):
                s = i % nbuf
                n = min(tc, T - t0)
                if i + 1 < len(starts):
                    stage_in(i + 1)
                main.wait_event(ev_h2d[s])
                if used[s]:
                    drain(i - nbuf)                              # (no-op on the direct path)
                    main.wait_event(ev_d2h[s])                   # buf[s] has been copied out
                oc = buf[s][:n]
                self.encode_device(xin[s][:n], ops, out=oc, state=state)
                ev_done[s].record(main)
                send_out(i)
                used[s] = True
            for i in range(max(0, len(starts) - nbuf), len(starts)):
                drain(i)
            d2h.synchronize()
            main.wait_stream(h2d)
            main.wait_stream(d2h)
        finally:
            if sink is not None:
                torch.cuda.synchronize(dev)
                sink.close()
        return out

    def encode_to_shards(self, x, ops, shard_dir, shard_steps):
        """Host tensor x[T, N, F] -> time shards on disk, ``shard_steps`` steps each, never holding more than
        one shard on the host: embeddings larger than host RAM (configuration C5 is 629 GB,
        ``experiments/run_largescale_sgp.py:208-212``).  The reservoir state is carried on the device, the
        propagation is independent per step: the shards are bit-identical to slices of one pass.  Returns a
        ``sgp_amd.datasets.ShardedEmbedding`` (``load_steps(t0, t1)`` reads any time range back)."""
        import os
        from ...datasets.sharded import ShardedEmbedding
        hip.require_gpu()
        T, N, F = x.shape
        os.makedirs(shard_dir, exist_ok=True)
        dev = torch.device("cuda", torch.cuda.current_device())
        L, R = len(self.reservoir.reservoir_layers), self.reservoir.hidden_size
        D = self.output_size
        per_step = N * (F + D) * 4
        ts = max(1, min(int(shard_steps), T, self._budget() // max(1, 2 * per_step)))
        state = torch.zeros(L, N, R, dtype=torch.float32, device=dev)
        hip.mark_unit_bounded(state)                          # starts at zero (see _state_bound)
        buf = torch.empty(ts, N, D, dtype=torch.float32, device=dev)
        pin = torch.empty(ts, N, D, dtype=torch.float32, pin_memory=True)
        paths = []
        for t0 in range(0, T, ts):
            n = min(ts, T - t0)
            xs = x[t0:t0 + n].float().to(dev)
            self.encode_device(xs if xs.stride(2) == 1 else xs.contiguous(), ops, out=buf[:n], state=state)
            pin[:n].copy_(buf[:n])                                # (synchronous: the shard is complete on the host)
            path = os.path.join(shard_dir, f"embedding_t{t0:08d}.pt")
            torch.save(dict(t0=t0, steps=n, rows=None, embedding=pin[:n].clone()), path)
            paths.append(path)
        ShardedEmbedding.write_index(shard_dir, paths, (T, N, D))
        return ShardedEmbedding(paths, T, N, D)

    # D2H straight into the (registered) result tensor; False = pinned bounce slots + host memcpy
    register_output = True

    def forward(self, x, edge_index, edge_weight, return_device=False, out=None, gpus=None, shard_dir=None,
                shard_steps=None):
        # x : [t n f]; ``return_device=True`` keeps the embedding of a host input on the GPU
        # (the next row f1 consumes it there: sgp_amd.datasets.IIDDataset); ``out``: host tensor
        # [t, n, d_out] to fill (host inputs only; see encode_streamed); ``gpus``: number of GPUs the
        # graph is node-partitioned over (None: SGP_AMD_GPUS, default 1; 0 / "all": every visible GPU) --
        # the call stays a single-process call on a host tensor, the ranks are started and joined inside
        # (sgp_amd/multigpu.py)
        # ``shard_dir``: write the embedding as shard files instead of returning one host tensor (embeddings
        # larger than host RAM) -- time shards of ``shard_steps`` steps on one GPU, time x node-block shards of
        # the ranks with ``gpus`` > 1; the call then returns a ``sgp_amd.datasets.ShardedEmbedding``
        from ... import multigpu
        n_gpus = multigpu.resolve_gpus(gpus)
        if n_gpus > 1:
            if x.is_cuda or return_device:
                raise ValueError("gpus > 1 takes a host tensor and returns a host tensor in the original node "
                                 "order (device shards live in the rank processes)")
            if shard_dir is not None:
                from ...datasets.sharded import ShardedEmbedding
                paths = multigpu.encode_multi_gpu(self, x, edge_index, edge_weight, n_gpus, shard_dir=shard_dir)
                shape = (x.shape[0], x.shape[1], self.output_size)
                ShardedEmbedding.write_index(shard_dir, paths, shape)
                return ShardedEmbedding(paths, *shape)
            return multigpu.encode_multi_gpu(self, x, edge_index, edge_weight, n_gpus, out=out)
        if shard_dir is not None:
            if x.is_cuda or return_device or out is not None:
                raise ValueError("shard_dir takes a host tensor and writes shard files")
            ops = self.sgp_encoder.operators(x.size(-2), edge_index, edge_weight)
            return self.encode_to_shards(x, ops, shard_dir, shard_steps or 256)
        dev = x.device
        ops = self.sgp_encoder.operators(x.size(-2), edge_index, edge_weight)
        xg = x.float()
        if not xg.is_cuda:
            hip.require_gpu()
            T, N, F = xg.shape
            per_step = N * (F + self.output_size) * 4
            budget = self._budget()
            if return_device:
                if T * per_step > budget:
                    raise RuntimeError("return_device=True: the embedding does not fit the device")
                return self.encode_device(xg.cuda(), ops)
            if T * per_step > min(budget, self.stream_threshold_bytes) and T > 1:
                # two input + two output buffers on the device; chunks of >= 32 steps keep the
                # propagation kernels at their full-size rates
                t_chunk = max(1, min(budget // (2 * per_step),
                                     max(32, self.stream_chunk_bytes // max(1, N * self.output_size * 4))))
                return self.encode_streamed(xg, ops, t_chunk, out=out)
            if out is not None:
                return self.encode_streamed(xg, ops, T, out=out)
            xg = xg.cuda()
        if out is not None:
            raise ValueError("out= is for host inputs (a device input returns a device tensor)")
        if xg.stride(2) != 1:
            xg = xg.contiguous()
        return self.encode_device(xg, ops).to(dev)

    def describe(self):
        """What makes an embedding re-derivable: constructor arguments, the
        per-layer leaking rates and every weight tensor."""
        res = self.reservoir
        sp = self.sgp_encoder
        return dict(
            encoder="SGPEncoder",
            kwargs=dict(input_size=res.input_size, reservoir_size=res.hidden_size,
                        reservoir_layers=res.num_layers, leaking_rate=res.leaking_rate,
                        spectral_radius=res.spectral_radius, density=res.density,
                        input_scaling=res.input_scaling, receptive_field=sp.receptive_field,
                        bidirectional=sp.bidirectional, alpha_decay=res.alpha_decay,
                        global_attr=sp.global_attr, add_self_loops=sp.add_self_loops,
                        undirected=sp.undirected, reservoir_activation=res.mode),
            alphas=[float(l.alpha) for l in res.reservoir_layers],
            state_dict={k: v.detach().cpu().clone() for k, v in self.state_dict().items()})

    @staticmethod
    def add_model_specific_args(parser):
        add_reservoir_args(parser)
        add_spatial_args(parser)
        return parser
