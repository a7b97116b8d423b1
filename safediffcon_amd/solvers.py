"""Evaluation rollouts: drop-ins for the solver calls that follow sampling.

  burgers_numeric_solve_free   1D/data/generate_burgers.py:207-299
  control_trajectories         1D/utils/metrics.py:42-65
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check


def burgers_numeric_solve_free(u0, f, visc, T, dt=1e-4, num_t=10, mode=None):
    """u0 (N,s), f (N,Nt,s) on the HIP device -> trajectory (N, Nt+1, s).  Same signature / semantics as the reference."""
    if mode == "const":
        raise ValueError
    assert f.size(1) == num_t, "check number of time interval"
    if not u0.is_cuda:
        raise RuntimeError("safediffcon_amd.solvers runs on MI355X only (no CPU fallback)")
    N, s = u0.shape[0], u0.size(-1)
    Nt = f.size(1)
    assert f.shape[0] == N
    delta_x = 1.0 / (s + 1)
    steps = math.ceil(T / dt)
    record_time = math.floor(steps / Nt)
    # coefficients rounded to fp32 from float64 like torch.FloatTensor(np.stack(D.data) / (2*delta_x)) does
    ct = float(np.float32(1.0 / (2 * delta_x)))
    d = (visc * np.array([1.0, -2.0, 1.0]) / delta_x ** 2).astype(np.float32)
    u0c = u0.detach().reshape(N, s).to(torch.float32).contiguous()
    fc = f.detach().reshape(N, Nt, s).to(torch.float32).contiguous()
    traj = torch.empty(N, Nt + 1, s, dtype=torch.float32, device=u0.device)
    stream = torch.cuda.current_stream(u0.device).cuda_stream
    check(_lib.get_lib().sdc_burgers_rollout(u0c.data_ptr(), fc.data_ptr(), traj.data_ptr(), N, s, Nt, steps, record_time,
                                             float(np.float32(dt)), ct, float(d[0]), float(d[1]), float(d[2]), stream),
          "sdc_burgers_rollout")
    return traj


def control_trajectories(diffused, nt):
    """Roll the sampled control force through the solver: diffused (B, C, padded_time, space) unscaled -> (B, nt, space)."""
    u = diffused[:, 0, :nt, :]
    fc = diffused[:, 1, :nt - 1, :]
    return burgers_numeric_solve_free(u[:, 0, :], fc, visc=0.01, T=1.0, dt=1e-4, num_t=10)
