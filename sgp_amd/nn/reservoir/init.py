"""Random initialisation of an echo-state layer.

What the seed -> weights contract fixes (golden vectors ``tests/golden/g4_seed_*.npz`` recorded from
the reference, ``lib/nn/reservoir/reservoir.py:54-75``) is the ORDER in which the global torch
generator is consumed and the arithmetic applied to each draw; this module states both once, for
``ReservoirLayer`` and ``GESNLayer`` alike:

  1. input matrix   U(-1, 1) of shape [units, inputs], times ``in_scaling``
  2. bias           U(-1, 1) of shape [units], times ``bias_scale``        (skipped without a bias)
  3. recurrent      U(-1, 1) of shape [units, units]
  4. sparsity       ``randperm(units^2)``: its first floor(units^2 (1 - density)) entries are the flat
                    (row-major) positions set to zero                       (skipped when density >= 1)
  5. the recurrent matrix is rescaled to the requested spectral radius (largest |eigenvalue|,
     ``torch.linalg.eigvals`` on the host; no random numbers)
"""
import torch


def _uniform_pm1(*shape):
    return torch.empty(*shape, dtype=torch.float32).uniform_(-1, 1)


def draw_reservoir_weights(input_size, hidden_size, density=1., spectral_radius=0.9, in_scaling=1.,
                           bias_scale=1., with_bias=True):
    """-> (w_in [units, inputs], bias [units] or None, w_rec [units, units]) as float32 CPU tensors,
    consuming the global torch RNG in the order listed in the module docstring."""
    units = int(hidden_size)
    w_in = _uniform_pm1(units, int(input_size)) * in_scaling
    bias = _uniform_pm1(units) * bias_scale if with_bias else None
    w_rec = _uniform_pm1(units, units)
    if density < 1:
        cells = units * units
        dropped = torch.randperm(cells)[:int(cells * (1 - density))]
        w_rec.view(-1)[dropped] = 0.
    radius = torch.linalg.eigvals(w_rec).abs().max()
    w_rec = w_rec * (spectral_radius / radius)
    return w_in, bias, w_rec
