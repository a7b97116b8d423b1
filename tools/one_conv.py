#!/usr/bin/env python3
"""Launch one conv shape N times (for PMC / trace passes).  usage: one_conv.py Cin Cout k H W B [reps]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan, as5
cin, cout, k, H, W, B = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = "cuda:0"
plan = Plan(dev, precision=int(os.environ.get('SDC_PRECISION', '0')))
x = torch.randn(B, cin, H, W, device=dev)
w = torch.randn(cout, cin, k, k, device=dev) * 0.05
b = torch.randn(cout, device=dev)
out = plan.conv(as5(x), plan.conv_weight(w), b, cout, (1, k, k), pad=(0, k // 2, k // 2))
s = torch.cuda.current_stream().cuda_stream
plan.run(s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    plan.run(s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"done {tuple(out.shape)} {ms:.4f} ms {2.0 * B * H * W * cout * cin * k * k / ms / 1e9:.1f} TF/s")
