#!/usr/bin/env python3
"""Run the fused temporal-attention block N times.  usage: ta_probe.py B H W [reps]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan
B, H, W = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = "cuda:0"
plan = Plan(dev)
x = torch.randn(B, 64, 32, H, W, device=dev)
g = torch.ones(64, device=dev)
wqkv = plan.conv_weight(torch.randn(384, 64, 1, device=dev) * 0.1)
wo = plan.conv_weight(torch.randn(64, 128, 1, device=dev) * 0.1)
rot = torch.randn(32 * 16 * 2, device=dev)
bias = torch.randn(4 * 32 * 32, device=dev)
y = plan.tattn_block(x, g, wqkv, wo, rot, bias)
s = torch.cuda.current_stream().cuda_stream
plan.run(s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    plan.run(s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
toks = B * 32 * H * W
fl = toks * (2.0 * 64 * 384 + 2 * 2 * 32 * 32 * 4 + 2.0 * 128 * 64)
print(f"tattn_block B={B} {H}x{W}: {ms:.3f} ms, {fl / ms / 1e9:.1f} TFLOP/s, x = {x.numel() * 4 / 1e6:.0f} MB")
