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
reference's ``input_encoder.1`` works).  Forward only: training the decoder is outside this
package's path (out of scope).
"""
import torch
from torch import nn

from ... import hip


class SGPInputEncoder(nn.Module):
    def __init__(self, input_size, order, hidden_size, activation="silu", dropout=0.):
        super().__init__()
        if input_size % order:
            raise ValueError("in_channels must be divisible by groups")      # nn.Conv1d's own check
        if activation not in hip.GL_ACT_CODES:
            raise ValueError(f"Activation '{activation}' not valid.")
        if dropout:
            raise NotImplementedError("forward-only layer: dropout must be 0")
        self.input_size, self.order = int(input_size), int(order)
        self.out_channels = hidden_size - hidden_size % order               # sgp_model.py:41
        if self.out_channels <= 0:
            raise ValueError("hidden_size must be at least order")
        self.activation = activation
        conv = nn.Conv1d(in_channels=input_size, out_channels=self.out_channels, kernel_size=1,
                         groups=order)                                       # same init, same RNG use
        self.weight, self.bias = conv.weight, conv.bias
        self._packed = None

    def _device_params(self, device):
        key = (self.weight._version, self.bias._version, str(device))
        if self._packed is None or self._packed[0] != key:
            w = self.weight.detach().to(device, torch.float32)
            self._packed = (key, hip.grouped_linear_pack(w, self.order),
                            self.bias.detach().to(device, torch.float32).contiguous())
        return self._packed[1], self._packed[2]

    @property
    def _ic(self):
        return self.input_size // self.order

    @property
    def _oc(self):
        return self.out_channels // self.order

    def forward(self, x):
        """x[b, n, f] (or [b, s, n, f]: the last step is used, sgp_model.py:96) -> [b, n, out]."""
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
        packed, bias = self._device_params(x.device)
        y = hip.grouped_linear(rows, packed, bias, self.order, self._ic, self._oc, self.activation)
        y = y.reshape(x.shape[0], x.shape[1], self.out_channels)
        return y.cpu() if on_cpu else y

    def forward_sampled(self, embedding, step_index, node_index):
        """Rows ``embedding[step_index[k], node_index[k], :]`` -> [K, 1, out_channels] without
        materialising the gathered batch (f1 + f4 in one launch)."""
        hip.require_gpu()
        if embedding.dim() != 3 or embedding.shape[-1] != self.input_size or not embedding.is_cuda:
            raise ValueError("embedding must be a CUDA tensor [T, N, input_size]")
        packed, bias = self._device_params(embedding.device)
        st = step_index.to(embedding.device, torch.int32)
        nd = node_index.to(embedding.device, torch.int32)
        y = hip.grouped_linear(None, packed, bias, self.order, self._ic, self._oc, self.activation,
                               step_index=st, node_index=nd, source=embedding)
        return y[:, None, :]
