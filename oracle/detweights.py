"""Deterministic synthetic weights / noise shared by the golden generator, the
tests and bench.py (no trained checkpoints exist: README.md:18-26 of the
reference points at Google Drive).  Values depend only on (seed, key order,
shapes) and on torch's CPU generator, which is identical in the build container
and on the GPU box (same image).

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math

import torch

_randn = torch.randn      # bound early: make_goldens patches torch.randn while the reference runs


def det_params(spec, seed=0):
    """spec: iterable of (key, shape) in state_dict order -> {key: fp32 tensor}."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for key, shape in spec:
        shape = tuple(int(s) for s in shape)
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("rotary_emb.freqs"):
            d = 2 * shape[0]
            P[key] = 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))
        elif key.endswith("relative_attention_bias.weight"):
            P[key] = 0.5 * _randn(shape, generator=g)
        elif leaf in ("g", "gamma") or key.endswith("norm.weight"):
            P[key] = 1.0 + 0.1 * _randn(shape, generator=g)
        elif leaf == "bias":
            P[key] = 0.05 * _randn(shape, generator=g)
        else:
            fan_in = max(1, math.prod(shape[1:]))
            P[key] = _randn(shape, generator=g) / math.sqrt(fan_in)
    return P


def det_noise(shape, seed):
    """noise(i) callable for the samplers: draw i is randn(seed + i)."""
    def noise(i):
        return _randn(shape, generator=torch.Generator().manual_seed(seed + i))
    return noise


def det_tensor(shape, seed, scale=1.0, lo=None, hi=None):
    x = scale * _randn(shape, generator=torch.Generator().manual_seed(seed))
    if lo is not None:
        x = x.clamp(lo, hi)
    return x
