#!/usr/bin/env python3
"""time the 7x7x7 stem conv of the smoke net (Cin 7 -> 64 at 32x64x64).  usage: stem_probe.py [B]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safediffcon_amd.engine import Plan, as5
import stages
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
plan = Plan(dev, precision=3)
xs = torch.randn(B, 32, 7, 64, 64, device=dev)           # frame-major state
x5 = xs.permute(0, 2, 1, 3, 4)
w = torch.randn(64, 7, 7, 7, 7, device=dev) * 0.05
b = torch.randn(64, device=dev)
out = plan.conv(x5, plan.conv_weight(w), b, 64, (7, 7, 7), pad=(3, 3, 3))
s = torch.cuda.current_stream().cuda_stream
plan.run(s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    plan.run(s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
wk = stages.classify(plan.lib, plan.calls[0][0], plan.calls[0][1])
ref = torch.nn.functional.conv3d(x5[:1].double(), w.double(), b.double(), padding=3)
err = (out[:1].double() - ref).abs().max().item() / ref.abs().max().item()
print(f"stem B={B}: {wk['kernel']} {ms:.3f} ms  {wk['flops'] / ms / 1e9:.1f} TF/s  rel err vs fp64 {err:.2e}")
