#!/usr/bin/env python3
"""A/B timing of the 3x3 / 3x3x3 conv shapes of C4 and C2 in the fp32 conv modes (0 direct, 2 Winograd along W, 3 Winograd
over H and W), one process, same box.  usage: python tools/wg_probe.py [B3d] [B2d]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safediffcon_amd.engine import Plan, as5  # noqa: E402
import stages  # noqa: E402

B3 = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B2 = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
shapes = [("c4 L0 64->64", 3, B3, 64, 0, 64, (32, 64, 64)), ("c4 L0 64+64->64", 3, B3, 64, 64, 64, (32, 64, 64)),
          ("c4 L1 128->128", 3, B3, 128, 0, 128, (32, 32, 32)), ("c4 L1 64->128", 3, B3, 64, 0, 128, (32, 32, 32)),
          ("c4 L2 256->256", 3, B3, 256, 0, 256, (32, 16, 16)), ("c4 L2 256+256->128", 3, B3, 256, 256, 128, (32, 16, 16)),
          ("c2 L0 64->64", 2, B2, 64, 0, 64, (16, 128)), ("c2 L1 128->128", 2, B2, 128, 0, 128, (8, 64)),
          ("c2 L2 256->256", 2, B2, 256, 0, 256, (4, 32)), ("c2 L3 512->512", 2, B2, 512, 0, 512, (2, 16))]
modes = [int(m) for m in os.environ.get("MODES", "2,3").split(",")]
if os.environ.get("SHAPES"):      # e.g. SHAPES=0,2: only these rows of the table
    shapes = [shapes[int(i)] for i in os.environ["SHAPES"].split(",")]
s = torch.cuda.current_stream().cuda_stream
for name, nd, B, c0, c1, co, sp in shapes:
    x = torch.randn(B, c0, *sp, device=dev)
    x1 = torch.randn(B, c1, *sp, device=dev) if c1 else None
    w = torch.randn(co, c0 + c1, *([3] * nd), device=dev) * 0.05
    b = torch.randn(co, device=dev)
    row = [f"{name:22s} B={B:3d}"]
    for prec in modes:
        plan = Plan(dev, precision=prec)
        k3, p3 = (1,) * (3 - nd) + (3,) * nd, (0,) * (3 - nd) + (1,) * nd
        plan.conv(as5(x), plan.conv_weight(w), b, co, k3, x1=None if x1 is None else as5(x1), pad=p3, gn_groups=8)
        plan.run(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            plan.run(s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        wk = stages.classify(plan.lib, plan.calls[0][0], plan.calls[0][1])
        row.append(f"p{prec} {wk['kernel'][:26]:26s} {ms:8.4f} ms eff {wk['flops'] / ms / 1e9:6.1f} issued {wk['issued'] / ms / 1e9:6.1f}")
    print(" | ".join(row), flush=True)
