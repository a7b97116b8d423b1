#!/usr/bin/env python3
"""One 3x3x3 conv + GroupNorm statistics launch (C4 level-0 shape by default) in precision 4 / 3, a few repetitions:
the workload for PMC passes on the F(2x2x2,3x3x3) kernel.  usage: python tools/one_wg3.py [B] [cin] [cout] [D H W]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan, as5  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sp = tuple(int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (32, 64, 64)
dev = "cuda:0"
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, cin, *sp, device=dev)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
plan = Plan(dev, precision=int(os.environ.get("SDC_PRECISION", "4")))
plan.conv(as5(x), plan.conv_weight(w), b, cout, (3, 3, 3), pad=(1, 1, 1), gn_groups=8)
for _ in range(3):
    plan.run(s)
torch.cuda.synchronize()
print("done")
