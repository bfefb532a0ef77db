import torch
from torch import nn

from ... import hip
from ..reservoir import Reservoir
from ._args import add_reservoir_args, add_spatial_args
from .sgp_spatial_encoder import SGPSpatialEncoder


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

    @property
    def output_size(self):
        return self.sgp_encoder.num_blocks() * self.reservoir.output_size

    def encode_device(self, x, ops, out=None):
        """x[T, N, F] CUDA float32 -> out[T, N, D_out] on the same device."""
        T, N, _ = x.shape
        d_h = self.reservoir.output_size
        if out is None:
            out = torch.empty(T, N, self.output_size, dtype=torch.float32, device=x.device)
        self.reservoir.encode_into(x, out[:, :, :d_h])
        self.sgp_encoder.encode_into(out, d_h, ops)
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

    def encode_streamed(self, x, ops, t_chunk):
        """Host tensor x[T, N, F] -> host tensor [T, N, D_out], ``t_chunk`` steps at a time,
        transfers overlapped with the compute (the drivers hand over host tensors,
        lib/utils.py:24-31): two device buffers and two pinned host buffers per direction, H2D
        of chunk i+1 and D2H of chunk i-1 on their own streams while chunk i is encoded, the
        host thread copying chunk i-1 out of its pinned slot meanwhile.  The recurrence is
        carried across chunks in a device-resident state ``[L, N, R]`` and the propagation is
        independent per time step, so the result is bit-identical to a single pass.  This is
        also how embeddings larger than the 288 GB of HBM (BASELINE config C5: 629 GB) or than
        the free memory are produced.  The returned tensor is ordinary (pageable) host memory:
        the reference's drivers fork DataLoader workers that inherit it copy-on-write."""
        hip.require_gpu()
        T, N, F = x.shape
        dev = torch.device("cuda", torch.cuda.current_device())
        L, R = len(self.reservoir.reservoir_layers), self.reservoir.hidden_size
        d_h, D = L * R, self.output_size
        tc = max(1, min(int(t_chunk), T))
        starts = list(range(0, T, tc))
        out = torch.empty(T, N, D, dtype=torch.float32)
        if T == 0:
            return out
        state = torch.zeros(L, N, R, dtype=torch.float32, device=dev)
        nbuf = 2 if len(starts) > 1 else 1
        xin = [torch.empty(tc, N, F, dtype=torch.float32, device=dev) for _ in range(nbuf)]
        buf = [torch.empty(tc, N, D, dtype=torch.float32, device=dev) for _ in range(nbuf)]
        x_pinned = x.is_pinned() and x.dtype == torch.float32
        pin_in = None if x_pinned else [torch.empty(tc, N, F, dtype=torch.float32, pin_memory=True)
                                        for _ in range(nbuf)]
        pin_out = [torch.empty(tc, N, D, dtype=torch.float32, pin_memory=True) for _ in range(nbuf)]
        main = torch.cuda.current_stream(dev)
        h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ev_h2d = [torch.cuda.Event() for _ in range(nbuf)]      # input slot holds its chunk
        ev_done = [torch.cuda.Event() for _ in range(nbuf)]     # compute of the slot's chunk finished
        ev_d2h = [torch.cuda.Event() for _ in range(nbuf)]      # pinned output slot holds its chunk
        used = [False] * nbuf

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

        def drain(i):
            s, t0 = i % nbuf, starts[i]
            n = min(tc, T - t0)
            ev_d2h[s].synchronize()
            out[t0:t0 + n].copy_(pin_out[s][:n])                 # host memcpy into pageable memory

        stage_in(0)
        for i, t0 in enumerate(starts):
            s = i % nbuf
            n = min(tc, T - t0)
            if i + 1 < len(starts):
                stage_in(i + 1)
            main.wait_event(ev_h2d[s])
            if used[s]:
                main.wait_event(ev_d2h[s])                       # buf[s] has been copied out
            oc = buf[s][:n]
            self.reservoir.encode_into(xin[s][:n], oc[:, :, :d_h], state)
            self.sgp_encoder.encode_into(oc, d_h, ops)
            ev_done[s].record(main)
            if i >= 1:
                drain(i - 1)                                     # under the compute of chunk i
            with torch.cuda.stream(d2h):
                d2h.wait_event(ev_done[s])
                pin_out[s][:n].copy_(oc, non_blocking=True)
                ev_d2h[s].record(d2h)
            used[s] = True
        drain(len(starts) - 1)
        main.wait_stream(h2d)
        main.wait_stream(d2h)
        return out

    def forward(self, x, edge_index, edge_weight, return_device=False):
        # x : [t n f]; ``return_device=True`` keeps the embedding of a host input on the GPU
        # (the next row f1 consumes it there: sgp_amd.datasets.IIDDataset)
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
                return self.encode_streamed(xg, ops, t_chunk)
            xg = xg.cuda()
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
