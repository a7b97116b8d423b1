#!/usr/bin/env python3
"""timing of the 1x1x1 conv shapes of the C4 step (B = 64 by default): python tools/pw_probe.py [B]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan, as5  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
shapes = [("to_qkv 128->384 L1", 128, 384, (32, 32, 32), False), ("to_out 128->128 L1 +res", 128, 128, (32, 32, 32), True),
          ("to_qkv 256->384 L2", 256, 384, (32, 16, 16), False), ("to_out 128->256 L2 +res", 128, 256, (32, 16, 16), True),
          ("res_conv 64->128 L1", 64, 128, (32, 32, 32), False), ("res_conv 128->256 L2", 128, 256, (32, 16, 16), False),
          ("res_conv 512->128 L1 (2 inputs)", 256, 128, (32, 32, 32), False),
          ("res_conv 128->64 L0", 128, 64, (32, 64, 64), False), ("final 1x1 64->64 L0", 64, 64, (32, 64, 64), False)]
s = torch.cuda.current_stream().cuda_stream
tot = 0.0
for name, cin, cout, sp, res in shapes:
    x = torch.randn(B, cin, *sp, device=dev)
    w = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.05
    r = torch.randn(B, cout, *sp, device=dev) if res else None
    plan = Plan(dev, precision=4)
    plan.conv(as5(x), plan.conv_weight(w), None, cout, (1, 1, 1), residual=r)
    plan.run(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        plan.run(s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tot += ms
    fl = 2.0 * B * sp[0] * sp[1] * sp[2] * cin * cout
    print(f"{name:34s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF/s", flush=True)
print(f"total {tot:.3f} ms")
