#!/usr/bin/env python3
"""Full-length (1000-step) guided smoke trajectories at production width, B=2, identical Philox noise, in the default conv
mode (4: Winograd over D, H, W) and in the direct fp32 mode (0: k-ordered FMA chains, the arithmetic closest to the
reference's): how far rounding-order differences drift over a whole reverse process.  usage: python tools/drift_probe.py [T]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safediffcon_amd as sdc  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
torch.manual_seed(0)
net = sdc.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7).to(dev)
init = (torch.rand(2, 64, 64) * 0.2).to(dev)
control = (torch.randn(2, 32, 2, 64, 64) * 0.3).to(dev)
outs = {}
for prec in (4, 3, 0):
    net.precision = prec
    gs = sdc.GaussianDiffusionSmoke(net, image_size=64, frames=32, timesteps=T, standard_fixed_ratio=100.0).to(dev)
    torch.manual_seed(7)
    t0 = time.perf_counter()
    outs[prec] = gs.sample(batch_size=2, design_fn=sdc.SmokeGuidance(0.01, 0.9, 0.1), init=init, control=control).cpu()
    print(f"precision {prec}: {time.perf_counter() - t0:.1f} s, finite {bool(torch.isfinite(outs[prec]).all())}, "
          f"|x|max {outs[prec].abs().max():.3f}", flush=True)
for a, b in ((4, 0), (3, 0), (4, 3)):
    d = (outs[a] - outs[b]).abs()
    print(f"[measured] {T}-step trajectories, precision {a} vs {b}: max|diff| {d.max():.3e}  mean|diff| {d.mean():.3e}  "
          f"MSE {(d ** 2).mean():.3e}")
