"""IID (step, node) sampling of a device-resident embedding -- "next" row f1 of SURVEY.md 8f.

Mirror of ``IIDDataset.sample`` (reference ``lib/datasets/iid_dataset.py:57-99``, driven by
``IIDLoader``, ``lib/dataloader/iid_dataloader.py:25-46``): a training batch is ``N`` random
``(t, n)`` rows of the encoded sequence plus the matching horizon rows of the target.  The
reference indexes host tensors (and first clones the 36-630 GB embedding in ``add_exogenous``);
here the embedding stays where the encoder wrote it, in HBM, and a batch is ONE gather kernel
launch per tensor (``sgp_gather_rows_f32``): 4096 rows x D_out floats, HBM-bound.

Semantics kept from the reference: index distribution and RNG order (``torch.randint`` for the
steps, then for the nodes, from the global CPU generator, :58-59), output shapes (``[N 1 1 f]``
inputs, ``[N h 1 f]`` targets, graph-level ``[t f]`` tensors give ``[N 1 f]`` / ``[N h 1 f]``),
the horizon grid ``t + delay + 1 .. t + horizon`` step ``horizon_lag`` (:78-80), scalers applied
AFTER the gather (:73-74, :93-94) and handed back under ``transform`` with a leading batch axis
(:70-72), ``input.node_index`` = ``node_index[:, None]`` (:98).
"""
from typing import Dict, Optional

import torch

from .. import hip


class _Entry:
    def __init__(self, tensor, pattern, scaler, preprocess):
        self.tensor, self.pattern, self.scaler, self.preprocess = tensor, pattern, scaler, preprocess


class IIDSampler:
    def __init__(self, n_steps: int, n_nodes: int, horizon: int, delay: int = 0,
                 horizon_lag: int = 1, device: Optional[torch.device] = None):
        self.n_steps, self.n_nodes = int(n_steps), int(n_nodes)
        self.horizon, self.delay, self.horizon_lag = int(horizon), int(delay), int(horizon_lag)
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self.inputs: Dict[str, _Entry] = {}
        self.targets: Dict[str, _Entry] = {}

    # ---- registration (the reference's input_map / target_map entries) -----------------------
    def _resident(self, tensor, pattern):
        hip.require_gpu()
        dims = pattern.split()
        if "t" not in dims or dims[0] != "t":
            raise ValueError(f"pattern {pattern!r}: the step axis must come first")
        want = 3 if "n" in dims else 2
        if tensor.dim() != want:
            raise ValueError(f"pattern {pattern!r} needs a {want}-D tensor, got {tuple(tensor.shape)}")
        if tensor.shape[0] != self.n_steps or ("n" in dims and tensor.shape[1] != self.n_nodes):
            raise ValueError("tensor does not match n_steps / n_nodes")
        t = tensor.to(self.device, torch.float32)
        return t if t.stride(-1) == 1 else t.contiguous()

    def add_input(self, key, tensor, pattern="t n f", scaler=None, preprocess=True):
        self.inputs[key] = _Entry(self._resident(tensor, pattern), pattern, scaler, preprocess)

    def add_target(self, key, tensor, pattern="t n f", scaler=None, preprocess=True):
        self.targets[key] = _Entry(self._resident(tensor, pattern), pattern, scaler, preprocess)

    # ---- sampling ------------------------------------------------------------------------------
    def draw(self, n):
        """(step_index, node_index) exactly as the reference draws them (:58-59)."""
        step_index = torch.randint(0, self.n_steps - self.horizon, (n,))
        node_index = torch.randint(0, self.n_nodes, (n,))
        return step_index, node_index

    def _rows(self, e, steps, nodes):
        """[K, f] rows (steps[k], nodes[k]) of a registered tensor through the HIP gather."""
        x = e.tensor if e.tensor.dim() == 3 else e.tensor[:, None, :]
        if e.tensor.dim() == 2:
            nodes = torch.zeros_like(nodes)
        return hip.gather_rows(x, steps, nodes)

    def sample(self, n, step_index=None, node_index=None):
        if step_index is None:
            step_index, node_index = self.draw(n)
        st = step_index.to(self.device, torch.int32)
        nd = node_index.to(self.device, torch.int32)
        out = dict(input={}, target={}, transform={}, pattern={})
        for key, e in self.inputs.items():
            rows = self._rows(e, st, nd)                                  # [N, f]
            tens = rows[:, None, None, :] if e.tensor.dim() == 3 else rows[:, None, :]
            if e.scaler is not None:
                out["transform"][key] = {k: p[None] for k, p in e.scaler.params().items()}
                if e.preprocess:
                    tens = e.scaler.transform(tens)
            out["input"][key] = tens
            out["pattern"][key] = e.pattern
        offs = torch.arange(self.delay + 1, self.horizon + 1, self.horizon_lag, device=self.device,
                            dtype=torch.int32)
        hor = (st[:, None] + offs[None, :]).reshape(-1).contiguous()      # [N * h]
        h = offs.numel()
        nd_h = nd[:, None].expand(-1, h).reshape(-1).contiguous()
        for key, e in self.targets.items():
            rows = self._rows(e, hor, nd_h)                               # [N * h, f]
            tens = rows.reshape(n, h, 1, rows.shape[-1])
            if e.scaler is not None:
                out["transform"][key] = {k: p[None] for k, p in e.scaler.params().items()}
                if e.preprocess:
                    tens = e.scaler.transform(tens)
            out["target"][key] = tens
            out["pattern"][key] = e.pattern
        out["input"]["node_index"] = node_index[:, None]
        return out
