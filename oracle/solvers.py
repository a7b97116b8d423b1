"""Oracle: Burgers' finite-difference rollout, restating burgers_numeric_solve_free
(1D/data/generate_burgers.py:207-299): same fp32 operation order, plain torch on the CPU.

TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def burgers_rollout(u0, f, visc=0.01, T=1.0, dt=1e-4):
    N, s = u0.shape
    Nt = f.shape[1]
    dx = 1.0 / (s + 1)
    steps = math.ceil(T / dt)
    rec = math.floor(steps / Nt)
    ct = torch.tensor(np.array([-1.0, 1.0]) / (2 * dx), dtype=torch.float32)
    dc = torch.tensor(visc * np.array([1.0, -2.0, 1.0]) / dx ** 2, dtype=torch.float32)
    u = F.pad(u0.reshape(N, s), (1, 1))
    fp = F.pad(f.reshape(N, Nt, s), (1, 1))
    sol = torch.zeros(N, Nt, s)
    c, fidx = 0, -1
    for j in range(steps):
        u = F.pad(u[:, 1:-1], (1, 1))
        us = u ** 2
        tr = torch.zeros_like(u)
        tr[:, 1:-1] = us[:, :-2] * ct[0] + us[:, 2:] * ct[1]
        df = torch.zeros_like(u)
        df[:, 1:-1] = (u[:, :-2] * dc[0] + u[:, 1:-1] * dc[1]) + u[:, 2:] * dc[2]
        if j % rec == 0:
            fidx += 1
        u = u + dt * (-(1 / 2) * tr + df + fp[:, fidx, :])
        if (j + 1) % rec == 0:
            sol[:, c] = u[:, 1:-1]
            c += 1
    return torch.cat((u0.reshape(N, 1, s), sol), dim=1)
