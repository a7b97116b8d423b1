#!/usr/bin/env python3
"""Fixed per-workgroup cost of conv_wg2_kernel: the same tensor through the 3x3x3 conv (K = 3 Cin) and the 1x3x3 conv
(K = Cin), precision 3.  usage: python tools/kd_probe.py [B]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safediffcon_amd.engine import Plan, as5  # noqa: E402
import stages  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
s = torch.cuda.current_stream().cuda_stream
for name, c0, co, sp in [("L0 64->64", 64, 64, (32, 64, 64)), ("L0 128->64", 128, 64, (32, 64, 64)), ("L1 128->128", 128, 128, (32, 32, 32)),
                         ("L2 256->256", 256, 256, (32, 16, 16))]:
    x = torch.randn(B, c0, *sp, device=dev)
    row = [f"{name:14s}"]
    for kd in (3, 1):
        w = torch.randn(co, c0, kd, 3, 3, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        plan = Plan(dev, precision=3)
        plan.conv(as5(x), plan.conv_weight(w), b, co, (kd, 3, 3), pad=(kd // 2, 1, 1), gn_groups=8)
        plan.run(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            plan.run(s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        wk = stages.classify(plan.lib, plan.calls[0][0], plan.calls[0][1])
        row.append(f"kD={kd} {wk['kernel'][:22]:22s} {ms:8.4f} ms issued {wk['issued'] / ms / 1e9:6.1f} TF/s")
    print(" | ".join(row), flush=True)
