#!/usr/bin/env python3
"""Workload for the FETCH_SIZE / WRITE_SIZE PMC passes: (1) calibration launches with a KNOWN byte count in the
same access shapes the conv uses (dword-per-lane loads: sdc_act over n floats reads 4n and writes 4n bytes),
(2) the dominant conv of the C2 bench (64->64 3x3 at (256,64,16,128), conv_wg_kernel<64,512,...>), a few launches each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan, as5
dev = "cuda:0"
plan = Plan(dev, precision=2)                # bench default: Winograd F(2,3) along W for the 3-tap convs
n = 64 * 1024 * 1024                       # 256 MiB in, 256 MiB out: beyond the 256 MiB Infinity Cache together
xa = torch.randn(n, device=dev); ya = torch.empty(n, device=dev)
plan.act(xa, 0, out=ya)
B, cin, cout = 256, 64, 64
x = torch.randn(B, cin, 16, 128, device=dev)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
out = plan.conv(as5(x), plan.conv_weight(w), b, cout, (1, 3, 3), pad=(0, 1, 1))
s = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    plan.run(s)
torch.cuda.synchronize()
print("act bytes read/written:", 4 * n, 4 * n, "conv algorithmic bytes:", 4 * (x.numel() + out.numel() + w.numel()))
