"""Golden vectors for the IID sampler ("next" row f1 of SURVEY.md 8f) -- container only.

TEST INFRASTRUCTURE.  Runs the UNMODIFIED ``lib/datasets/iid_dataset.py`` of the reference under the
import shim (``oracle/ref_shim.py`` plus the three ``tsl.data`` names the file imports, stubbed
here: ``Data`` = attribute dictionary, ``ScalerModule`` = parameter holder, ``WINDOW``/``HORIZON``
= the enum values of ``tsl/data/utils.py:14-21``) and records what ``IIDDataset.sample`` returns
for a hand-made dataset object: the drawn indices and every gathered tensor.

    python oracle/make_golden_iid.py        # writes tests/golden/g7_iid_*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


class AttrDict(dict):
    """``sample.input[key] = v`` and ``sample.input.node_index = v`` (tsl.data.Data views)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class Data:
    def __init__(self):
        self.input, self.target = AttrDict(), AttrDict()
        self.pattern, self.transform = {}, {}


class ScalerModule:
    def __init__(self, **params):
        self.params = params


def load_iid_dataset():
    ref_shim.load_reference()
    tsl_data = sys.modules["tsl.data"]
    tsl_data.Data = Data
    prep = types.ModuleType("tsl.data.preprocessing")
    prep.ScalerModule = ScalerModule
    sys.modules["tsl.data.preprocessing"] = prep
    utils = types.ModuleType("tsl.data.utils")
    utils.WINDOW, utils.HORIZON = "window", "horizon"
    utils.outer_pattern = lambda patterns: " ".join(sorted({d for p in patterns for d in p.split()},
                                                           reverse=True))
    sys.modules["tsl.data.utils"] = utils
    typing_mod = types.ModuleType("tsl.typing")
    typing_mod.TensArray = object
    sys.modules["tsl.typing"] = typing_mod
    path = os.path.join(ref_shim.REFERENCE_ROOT, "lib", "datasets", "iid_dataset.py")
    spec = importlib.util.spec_from_file_location("ref_iid_dataset", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.IIDDataset


class Entry:
    def __init__(self, key, preprocess):
        self.keys, self.preprocess = [key], preprocess


class InputMap(dict):
    def by_synch_mode(self, mode):
        assert mode == "window"
        return self


class Scaler:
    """Stand-in for a fitted tsl scaler: (x - bias) / scale, parameters broadcast over nodes."""
    def __init__(self, bias, scale):
        self.bias, self.scale = bias, scale

    def params(self):
        return dict(bias=self.bias, scale=self.scale)

    def transform(self, x):
        return (x - self.bias) / self.scale


class FakeSelf:
    """The attributes ``IIDDataset.sample`` (iid_dataset.py:57-99) reads."""


def main():
    IID = load_iid_dataset()
    g = torch.Generator().manual_seed(77)
    cases = [
        dict(name="plain", T=60, N=17, F=40, C=1, horizon=3, delay=0, lag=1, n=64, scaler=False, exo=False),
        dict(name="exo_scaled", T=48, N=9, F=24, C=2, horizon=6, delay=1, lag=2, n=33, scaler=True, exo=True),
        dict(name="h1", T=20, N=5, F=8, C=1, horizon=1, delay=0, lag=1, n=7, scaler=True, exo=False),
    ]
    for idx, c in enumerate(cases):
        T, N, F, C = c["T"], c["N"], c["F"], c["C"]
        emb = torch.randn(T, N, F, generator=g)
        y = torch.randn(T, N, C, generator=g)
        u = torch.randn(T, 3, generator=g)
        fs = FakeSelf()
        fs.n_steps, fs.n_nodes = T, N
        fs.horizon, fs.delay, fs.horizon_lag = c["horizon"], c["delay"], c["lag"]
        fs.x, fs.y, fs.u = emb, y, u
        fs.patterns = dict(x="t n f", y="t n f", u="t f")
        fs.input_map = InputMap(x=Entry("x", True))
        if c["exo"]:
            fs.input_map["u"] = Entry("u", False)
        fs.target_map = dict(y=Entry("y", True))
        bias, scale = torch.randn(1, N, C, generator=g), torch.rand(1, N, C, generator=g) + .5
        fs.scalers = dict(y=Scaler(bias, scale)) if c["scaler"] else {}
        seed = 900 + idx
        torch.manual_seed(seed)
        out = IID.sample(fs, c["n"])
        # the indices the reference drew (same global RNG stream, same order)
        torch.manual_seed(seed)
        step_index = torch.randint(0, T - c["horizon"], (c["n"],))
        node_index = torch.randint(0, N, (c["n"],))
        assert torch.equal(out.input.node_index, node_index[:, None])
        arrays = dict(emb=emb.numpy(), y=y.numpy(), u=u.numpy(), seed=np.int64(seed),
                      cfg=np.array([c["horizon"], c["delay"], c["lag"], c["n"]], dtype=np.int64),
                      has_scaler=np.bool_(c["scaler"]), has_exo=np.bool_(c["exo"]),
                      bias=bias.numpy(), scale=scale.numpy(),
                      step_index=step_index.numpy(), node_index=node_index.numpy(),
                      out_x=out.input["x"].numpy(), out_y=out.target["y"].numpy(),
                      out_node_index=out.input.node_index.numpy())
        if c["exo"]:
            arrays["out_u"] = out.input["u"].numpy()
        if c["scaler"]:
            tr = out.transform["y"].params
            arrays["tr_bias"], arrays["tr_scale"] = tr["bias"].numpy(), tr["scale"].numpy()
        np.savez_compressed(os.path.join(GOLDEN, f"g7_iid_{c['name']}.npz"), **arrays)
        print("wrote g7_iid_" + c["name"], tuple(out.input["x"].shape), tuple(out.target["y"].shape))


if __name__ == "__main__":
    main()
