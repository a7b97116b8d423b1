#!/usr/bin/env python3
"""which python lines of a fine-tuning step launch torch's own kernels (copies, adds, cats ...):
python tools/ft_aten_ops.py [c2|c3] -- torch.profiler with stacks, aten ops grouped by the innermost safediffcon_amd frame"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safediffcon_amd as sdc  # noqa: E402
from oracle.detweights import det_tensor  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
B = 64
if wl == "c2":
    net = sdc.Unet2D(dim=64, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1).to(dev)
    gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=1000, temporal=True, use_conv2d=True,
                                      is_condition_u0=True, is_condition_uT=True, condition_idx=10).to(dev)
    shape = (3, 16, 128)
else:
    net = sdc.Unet1D(dim=256, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1).to(dev)
    gd = sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=1000).to(dev)
    shape = (12, 128)
state = det_tensor((B, *shape), 9, 0.3).to(dev)
w = torch.ones(B, device=dev)
t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3)).to(dev)
noise = det_tensor((B, *shape), 10).to(dev)


def step():
    net.zero_grad(set_to_none=True)
    loss = (w * gd.p_losses(state, t, noise=noise, mean=False)).mean()
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::sum", "aten::clone", "aten::fill_", "aten::zero_",
         "aten::index", "aten::sub", "aten::div", "aten::mean", "aten::neg", "aten::sqrt", "aten::pow")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    where = "(autograd engine / no python frame)"
    frames = [fr for fr in ev.stack if "safediffcon_amd/" in fr]
    if frames:
        where = " <- ".join(fr.split("safediffcon_amd/")[-1] for fr in frames[:2])
    elif ev.stack:
        where = "(no repo frame) " + ev.stack[0][-60:]
    cnt[(ev.name, where)] += 1
for (name, where), n in cnt.most_common(45):
    print(f"{n:5d}  {name:14s} {where}")
