"""Golden vectors for the decoder's first layer ("next" row f4 of SURVEY.md 8f) -- container only.

TEST INFRASTRUCTURE.  Imports the UNMODIFIED ``lib/nn/models/sgp_model.py`` under the shim
(``oracle/ref_shim.py``; the names the file imports for the parts of the model that are NOT on
this row -- ``StaticGraphEmbedding``, ``LinearReadout``, ``MLP``, ``ResidualMLP`` -- are stubbed
with placeholders), builds the reference's ``SGPModel`` and records input / parameters / output
of its ``input_encoder`` (``sgp_model.py:41-52``: Rearrange, grouped ``Conv1d(kernel_size=1,
groups=order)``, Rearrange, activation, Dropout(0)) and, for the backward pass of that trained layer,
the gradients ``autograd`` gives the reference module for a recorded cotangent ``gy``: ``gx`` (input),
``gw`` / ``gb`` (``input_encoder.1.weight.grad`` / ``.bias.grad``).

    python oracle/make_golden_decoder.py    # writes tests/golden/g8_decoder_*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _Placeholder(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def load_sgp_model():
    ref_shim.load_reference()
    base = types.ModuleType("tsl.nn.base")
    base.StaticGraphEmbedding = _Placeholder
    sys.modules["tsl.nn.base"] = base
    dec = types.ModuleType("tsl.nn.blocks.decoders")
    dec.LinearReadout = _Placeholder
    sys.modules["tsl.nn.blocks.decoders"] = dec
    enc = sys.modules["tsl.nn.blocks.encoders"]
    enc.MLP, enc.ResidualMLP = _Placeholder, _Placeholder
    path = os.path.join(ref_shim.REFERENCE_ROOT, "lib", "nn", "models", "sgp_model.py")
    spec = importlib.util.spec_from_file_location("ref_sgp_model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SGPModel


def main():
    SGPModel = load_sgp_model()
    g = torch.Generator().manual_seed(4242)
    cases = [
        # name, input_size (= order * per-group width), order, hidden, activation, x shape
        ("la_silu", 192, 3, 32, "silu", (37, 1, 192)),
        ("bay_silu", 1280, 10, 256, "silu", (50, 1, 1280)),
        ("pv_relu", 512, 4, 64, "relu", (8, 5, 512)),
        ("odd_width", 60, 5, 33, "silu", (19, 3, 60)),
        ("window4d", 256, 4, 128, "silu", (6, 2, 7, 256)),
    ]
    for idx, (name, f, order, hidden, act, shape) in enumerate(cases):
        torch.manual_seed(800 + idx)
        model = SGPModel(input_size=f, order=order, n_nodes=shape[-2], hidden_size=hidden,
                         mlp_size=16, output_size=1, n_layers=1, horizon=1,
                         positional_encoding=False, activation=act)
        x = torch.randn(*shape, generator=g)
        xin = x[:, -1] if x.ndim == 4 else x                      # sgp_model.py:96
        with torch.no_grad():
            y = model.input_encoder(xin)
            y64 = model.input_encoder.double()(xin.double())
        model.input_encoder.float()
        conv = model.input_encoder[1]
        # backward of the reference module itself: loss = <y, gy>
        xg = xin.clone().requires_grad_(True)
        yg = model.input_encoder(xg)
        gy = torch.randn(*yg.shape, generator=torch.Generator().manual_seed(900 + idx))
        model.input_encoder.zero_grad()
        yg.backward(gy)
        grads = dict(gy=gy.numpy(), gx=xg.grad.numpy(), gw=conv.weight.grad.numpy().copy(),
                     gb=conv.bias.grad.numpy().copy())
        np.savez_compressed(os.path.join(GOLDEN, f"g8_decoder_{name}.npz"),
                            x=x.numpy(), y=y.numpy(), y64=y64.numpy(),
                            weight=conv.weight.detach().numpy(), bias=conv.bias.detach().numpy(),
                            cfg=np.array([f, order, hidden], dtype=np.int64), activation=np.array(act),
                            seed=np.int64(800 + idx), **grads)
        print("wrote g8_decoder_" + name, tuple(x.shape), "->", tuple(y.shape), tuple(conv.weight.shape))


if __name__ == "__main__":
    main()
