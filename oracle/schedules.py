"""Oracle: DDPM schedule tables (fp64 on the host, cast to fp32 at the end).

Restates
  * cosine / linear schedules   1D/model/model_utils.py:142-158
                                (identical in tokamak/model/model_utils.py)
  * sigmoid schedule            2d/ddpm/diffusion_2d.py:95-108
  * derived tables              1D/model/diffusion.py:111-156,
                                2d/ddpm/diffusion_2d.py:137-191

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math

import torch


def betas_linear(T):
    s = 1000.0 / T
    return torch.linspace(s * 1e-4, s * 2e-2, T, dtype=torch.float64)


def betas_cosine(T, s=0.008):
    x = torch.linspace(0, T, T + 1, dtype=torch.float64)
    ac = torch.cos(((x / T) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def betas_sigmoid(T, start=-3, end=3, tau=1):
    t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
    # NB: the reference builds v_start/v_end from python floats -> fp32 tensors
    v0 = torch.tensor(start / tau).sigmoid()
    v1 = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0)
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


_BETAS = {"linear": betas_linear, "cosine": betas_cosine, "sigmoid": betas_sigmoid}


def make_tables(kind, T=1000):
    """dict of fp32 [T] tensors with the reference's buffer names."""
    b = _BETAS[kind](T)
    a = 1.0 - b
    ac = torch.cumprod(a, dim=0)
    acp = torch.cat([torch.ones(1, dtype=ac.dtype), ac[:-1]])
    pv = b * (1.0 - acp) / (1.0 - ac)
    tabs = {
        "betas": b,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": acp,
        "sqrt_alphas_cumprod": ac.sqrt(),
        "sqrt_one_minus_alphas_cumprod": (1.0 - ac).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - ac).log(),
        "sqrt_recip_alphas_cumprod": (1.0 / ac).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / ac - 1).sqrt(),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": pv.clamp(min=1e-20).log(),
        "posterior_mean_coef1": b * acp.sqrt() / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - acp) * a.sqrt() / (1.0 - ac),
    }
    return {k: v.to(torch.float32) for k, v in tabs.items()}
