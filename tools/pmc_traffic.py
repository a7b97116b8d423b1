#!/usr/bin/env python3
"""Workload for the FETCH_SIZE / WRITE_SIZE / SQ PMC passes: (1) calibration launches with a KNOWN byte count in the
same access shapes the conv uses (dword-per-lane loads: sdc_act over n floats reads 4n and writes 4n bytes),
(2) the dominant conv of the bench workload, a few launches:
      c4 (default): 64->64 3x3x3 at (64,64,32,64,64) + GroupNorm statistics in the epilogue (the C4 level-0 ResnetBlock conv)
      c2:           64->64 3x3   at (256,64,16,128)
(3) c4 only: the fused temporal-attention block at width 64 on the same tensor.
usage: python3 tools/pmc_traffic.py [c4|c2] [batch]   (under rocprofv3 --kernel-trace --pmc ... -- python3 tools/pmc_traffic.py)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan, as5
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = "cuda:0"
prec = int(os.environ.get("SDC_PRECISION", "4"))
plan = Plan(dev, precision=prec)
n = 64 * 1024 * 1024                       # 256 MiB in, 256 MiB out: beyond the 256 MiB Infinity Cache together
xa = torch.randn(n, device=dev); ya = torch.empty(n, device=dev)
plan.act(xa, 0, out=ya)
if wl == "c2":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    x = torch.randn(B, 64, 16, 128, device=dev)
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    k, pad = (1, 3, 3), (0, 1, 1)
else:
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    x = torch.randn(B, 64, 32, 64, 64, device=dev)
    w = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.03
    k, pad = (3, 3, 3), (1, 1, 1)
b = torch.randn(64, device=dev)
out = plan.conv(as5(x), plan.conv_weight(w), b, 64, k, pad=pad, gn_groups=8 if wl != "c2" else 1)
if wl != "c2":
    g = torch.ones(64, device=dev)
    wqkv = plan.conv_weight(torch.randn(384, 64, 1, device=dev) * 0.1)
    wo = plan.conv_weight(torch.randn(64, 128, 1, device=dev) * 0.1)
    rot = torch.randn(32 * 16 * 2, device=dev)
    bias = torch.randn(4 * 32 * 32, device=dev)
    plan.tattn_block(x, g, wqkv, wo, rot, bias)
    gm, bt = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    plan.gn_silu(out, gm, bt, 8)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    plan.run(s)
torch.cuda.synchronize()
print("act bytes read/written:", 4 * n, 4 * n, "conv algorithmic bytes:", 4 * (x.numel() + out.numel() + w.numel()),
      "x bytes:", 4 * x.numel())
