#!/usr/bin/env python3
"""FULL-schedule (T = 1000) guided DDPM trajectories at C4 WIDTH (Unet3D_with_Conv3D dim 64, 32 frames of 64 x 64), injected
noise: the HIP sampler against the oracle's loop + functional net executed by PyTorch-ROCm eager on the same device (held to
the CPU oracle in tests/test_gpu_strawman.py).  Too long for the test suite (the eager side runs ~0.3 s per step); run once per
round, log under profiles/.  Reference: 2d/ddpm/diffusion_2d.py:288-322.  usage: python tools/c4_t1000_parity.py [B] [T]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safediffcon_amd as sdc  # noqa: E402
from oracle import nets as onets  # noqa: E402
from oracle import samplers as osam  # noqa: E402
from oracle import schedules as osched  # noqa: E402
from oracle.detweights import det_noise, det_params, det_tensor  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
net = sdc.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7)
P = det_params([(k, tuple(v.shape)) for k, v in net.state_dict().items()], 31)
net.load_state_dict(P)
net.to(DEV)
gs = sdc.GaussianDiffusionSmoke(net, image_size=64, frames=32, timesteps=T, standard_fixed_ratio=100.0).to(DEV)
init = det_tensor((B, 64, 64), 43, 0.2).abs()
noise = det_noise((B, 32, 7, 64, 64), 7000)
t0 = time.time()
out = gs.sample(batch_size=B, design_fn=sdc.SmokeGuidance(0.01, 0.9, -5.0), init=init.to(DEV), noise=noise)
torch.cuda.synchronize()
t_hip = time.time() - t0
free = gs.sample(batch_size=B, design_fn=None, init=init.to(DEV), noise=noise)
Pg = {k: v.to(DEV) for k, v in P.items()}
tabs = osched.make_tables("sigmoid", T)
t0 = time.time()
last = [t0]


def nz(s):
    if time.time() - last[0] > 60:
        print(f"  oracle at step {s} ({time.time() - t0:.0f} s)", flush=True)
        last[0] = time.time()
    return noise(s).to(DEV)


ref = osam.sample_smoke(lambda a, b: onets.unet_smoke(Pg, a, b.to(a.device), dim=64, dim_mults=(1, 2, 4)), tabs, B, nz,
                        init=init.to(DEV), design_fn=osam.smoke_guidance(0.01, 0.9, -5.0), ratio=100.0, shape=(32, 7, 64, 64))
torch.cuda.synchronize()
t_ref = time.time() - t0
d = (out - ref).float()
err = d.abs().max().item()
mse = (d ** 2).flatten(1).mean(1).max().item()
print(f"C4 width, T = {T} guided DDPM, B = {B}: HIP sampler {t_hip:.1f} s, eager-GPU oracle {t_ref:.1f} s")
print(f"max|err| {err:.3e}   worst per-trajectory MSE {mse:.3e}   |ref|max {ref.abs().max().item():.3f}   "
      f"guided vs unguided max|diff| {(out - free).abs().max().item():.3e}")
ok = torch.isfinite(out).all().item() and err < 1e-3 and mse <= 1e-5
print("PASS" if ok else "FAIL", "(gate: north star eps-MSE <= 1e-5; element-wise < 1e-3)")
sys.exit(0 if ok else 1)
