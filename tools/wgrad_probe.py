#!/usr/bin/env python3
"""time sdc_conv_wgrad on one conv shape: python tools/wgrad_probe.py B Cin Cout D H W kD kH kW [reps]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd import grad_ops  # noqa: E402

B, Cin, Cout, D, H, W, kD, kH, kW = (int(v) for v in sys.argv[1:10])
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 5
dev = "cuda:0"
g = torch.randn(B, Cout, D, H, W, device=dev)
x = torch.randn(B, Cin, D, H, W, device=dev)
k, p = (kD, kH, kW), (kD // 2, kH // 2, kW // 2)
grad_ops.conv_wgrad(g, x, k, (1, 1, 1), p)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    grad_ops.conv_wgrad(g, x, k, (1, 1, 1), p)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gf = 2.0 * B * D * H * W * Cin * Cout * kD * kH * kW / 1e9
print(f"wgrad B={B} {Cin}->{Cout} {D}x{H}x{W} k={k}: {ms:.3f} ms  {gf / ms:.1f} TFLOP/s (direct form, incl. the split reduction)", flush=True)
