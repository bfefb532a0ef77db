"""First layer of the SGP decoder on the GPU -- "next" row f4 of SURVEY.md 8f.

``SGPModel.input_encoder`` of the reference (``lib/nn/models/sgp_model.py:41-52``) is
``Rearrange('b n f -> b f n')``, ``nn.Conv1d(input_size, out_channels, kernel_size=1,
groups=order)``, ``Rearrange('b f n -> b n f')``, activation, ``Dropout``: a block-diagonal linear
map that lets every one of the ``order`` blocks of the embedding (``[H | A H | ... ]``, the layout
the encoder writes) feed its own ``out_channels / order`` units.  Here it is one HIP kernel
(``sgp_grouped_linear_f32``, fp32 MFMA) that reads the block layout in place, optionally fused
with the IID gather of row f1 so that a training batch goes from the embedding in HBM straight to
the first hidden activations.

Parameters keep the reference's names and shapes (``weight [out_channels, input_size / order, 1]``,
``bias [out_channels]``, initialised by ``nn.Conv1d`` itself, so ``load_state_dict`` of the
reference's ``input_encoder.1`` works).

The reference trains this layer (Lightning optimiser loop around ``sgp_model.py:41-52``), so it has a
backward pass: ``_GroupedLinearFn`` is an ``autograd.Function`` whose pieces are HIP kernels too --
``dz = dy * dropout * act'(z)``, ``dx`` = the same forward kernel on ``dz`` with the transposed
grouped weight, ``dW`` on the fp32 matrix cores with the rows as contraction index, ``db`` = column
sums of ``dz``.  ``Dropout(p)`` (``sgp_model.py:50``) is a Philox4x32-10 mask keyed by a per-call seed
drawn from torch's generator and recomputed, not stored, by the backward pass; like ``nn.Dropout`` it
is the identity in ``eval()`` mode.  (The reference's CPU / CUDA dropout streams differ from each
other too: the mask is not part of the parity contract, its rate, scale and forward / backward
consistency are -- tests/test_gpu_parity.py.)
"""
import torch
from torch import nn

from ... import hip


class _GroupedLinearFn(torch.autograd.Function):
    """y = dropout(act(grouped_linear(rows))) with rows either ``x2[K, groups*ic]`` or gathered from
    ``source[step, node]``; gradients for x2 (not for a gathered source: the embedding is data),
    weight and bias."""

    @staticmethod
    def forward(ctx, x2, weight, bias, source, step, node, groups, activation, p, seed, cached=None):
        w2 = weight.reshape(weight.shape[0], -1)
        oc, ic = w2.shape[0] // groups, w2.shape[1]
        dev = x2.device if x2 is not None else source.device
        if cached is not None:                      # the module's per-(version, device) copies: no repacking per call
            wd, packed, bd = cached
        else:
            wd = w2.detach().to(dev, torch.float32)
            packed = hip.grouped_linear_pack(wd, groups)
            bd = bias.detach().to(dev, torch.float32).contiguous()
        y, pre = hip.grouped_linear(x2, packed, bd, groups, ic, oc, activation, step_index=step,
                                    node_index=node, source=source, want_pre=True, dropout_p=p, seed=seed)
        ctx.save_for_backward(x2 if x2 is not None else source, wd, pre, step, node)
        ctx.cfg = (groups, ic, oc, activation, p, seed, x2 is None, weight.shape, weight.device, bias.device)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, wd, pre, step, node = ctx.saved_tensors
        groups, ic, oc, activation, p, seed, sampled, wshape, wdev, bdev = ctx.cfg
        dz = hip.grouped_linear_dact(dy, pre, activation, dropout_p=p, seed=seed)
        dx = dw = db = None
        if ctx.needs_input_grad[0] and not sampled:
            wt = hip.grouped_linear_pack(hip.grouped_linear_transpose(wd, groups), groups)
            zero = torch.zeros(groups * ic, dtype=torch.float32, device=dz.device)
            dx = hip.grouped_linear(dz, wt, zero, groups, oc, ic, None)
        if ctx.needs_input_grad[1]:
            if sampled:
                dw = hip.grouped_linear_wgrad(None, dz, groups, ic, oc, step_index=step, node_index=node,
                                              source=rows)
            else:
                dw = hip.grouped_linear_wgrad(rows, dz, groups, ic, oc)
            dw = dw.reshape(wshape).to(wdev)
        if ctx.needs_input_grad[2]:
            db = hip.node_sums(dz[None])[0].to(bdev)
        return dx, dw, db, None, None, None, None, None, None, None, None


class SGPInputEncoder(nn.Module):
    def __init__(self, input_size, order, hidden_size, activation="silu", dropout=0.):
        super().__init__()
        if input_size % order:
            raise ValueError("in_channels must be divisible by groups")      # nn.Conv1d's own check
        if activation not in hip.GL_ACT_CODES:
            raise ValueError(f"Activation '{activation}' not valid.")
        if not 0. <= float(dropout) <= 1.:                                   # nn.Dropout's own range
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {dropout}")
        self.dropout = float(dropout)
        self.input_size, self.order = int(input_size), int(order)
        self.out_channels = hidden_size - hidden_size % order               # sgp_model.py:41
        if self.out_channels <= 0:
            raise ValueError("hidden_size must be at least order")
        self.activation = activation
        conv = nn.Conv1d(in_channels=input_size, out_channels=self.out_channels, kernel_size=1,
                         groups=order)                                       # same init, same RNG use
        self.weight, self.bias = conv.weight, conv.bias
        self._packed = None

    def _device_params(self, device, with_weight=False):
        """Packed weights + bias on ``device``, rebuilt only when a parameter changed (``_version``) -- also for
        the autograd path (``with_weight``: plus the plain fp32 weight the backward pass transposes)."""
        key = (self.weight._version, self.bias._version, str(device))
        if self._packed is None or self._packed[0] != key:
            w = self.weight.detach().reshape(self.weight.shape[0], -1).to(device, torch.float32)
            self._packed = (key, hip.grouped_linear_pack(w, self.order),
                            self.bias.detach().to(device, torch.float32).contiguous(), w)
        if with_weight:
            return self._packed[3], self._packed[1], self._packed[2]
        return self._packed[1], self._packed[2]

    @property
    def _ic(self):
        return self.input_size // self.order

    @property
    def _oc(self):
        return self.out_channels // self.order

    def forward(self, x):
        """x[b, n, f] (or [b, s, n, f]: the last step is used, sgp_model.py:96) -> [b, n, out].  Inference
        should run under ``torch.no_grad()``: with grad mode on and trainable parameters every call goes
        through the autograd function (an extra [rows, out] pre-activation buffer)."""
        x = x[:, -1] if x.dim() == 4 else x
        if x.dim() != 3 or x.shape[-1] != self.input_size:
            raise ValueError(f"expected [b, n, {self.input_size}], got {tuple(x.shape)}")
        on_cpu = not x.is_cuda
        if on_cpu:
            hip.require_gpu()
            x = x.cuda()
        x = x.float()
        rows = x.reshape(-1, self.input_size)
        if rows.stride(1) != 1:
            rows = rows.contiguous()
        if self._needs_graph(rows):
            y = _GroupedLinearFn.apply(rows, self.weight, self.bias, None, None, None, self.order,
                                       self.activation, *self._dropout_args(),
                                       self._device_params(x.device, with_weight=True))
            if self.training and self.dropout >= 1.:
                y = y * 0.                                                   # nn.Dropout(p=1): all zeros
        else:
            packed, bias = self._device_params(x.device)
            y = hip.grouped_linear(rows, packed, bias, self.order, self._ic, self._oc, self.activation)
        y = y.reshape(x.shape[0], x.shape[1], self.out_channels)
        return y.cpu() if on_cpu else y

    def _dropout_args(self):
        """(p, seed): a fresh 63-bit seed from torch's default generator per training-mode call."""
        if not (self.training and 0. < self.dropout < 1.):                   # (p = 1 is applied by the caller)
            return 0., 0
        return self.dropout, int(torch.randint(0, 2 ** 62, (1,)).item())

    def _needs_graph(self, rows=None):
        grad = torch.is_grad_enabled() and (self.weight.requires_grad or self.bias.requires_grad or
                                            (rows is not None and rows.requires_grad))
        return grad or (self.training and self.dropout > 0.)

    def forward_sampled(self, embedding, step_index, node_index):
        """Rows ``embedding[step_index[k], node_index[k], :]`` -> [K, 1, out_channels] without
        materialising the gathered batch (f1 + f4 in one launch)."""
        hip.require_gpu()
        if embedding.dim() != 3 or embedding.shape[-1] != self.input_size or not embedding.is_cuda:
            raise ValueError("embedding must be a CUDA tensor [T, N, input_size]")
        st = step_index.to(embedding.device, torch.int32)
        nd = node_index.to(embedding.device, torch.int32)
        if self._needs_graph():
            y = _GroupedLinearFn.apply(None, self.weight, self.bias, embedding.detach(), st, nd, self.order,
                                       self.activation, *self._dropout_args(),
                                       self._device_params(embedding.device, with_weight=True))
            if self.training and self.dropout >= 1.:
                y = y * 0.
        else:
            packed, bias = self._device_params(embedding.device)
            y = hip.grouped_linear(None, packed, bias, self.order, self._ic, self._oc, self.activation,
                                   step_index=st, node_index=nd, source=embedding)
        return y[:, None, :]
