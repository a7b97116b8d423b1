#!/usr/bin/env python3
"""Run one fused LinearAttention block N times (for rocprofv3 kernel stats).  usage: la_probe.py B C n [F] [reps]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan
B, Cc, n = (int(v) for v in sys.argv[1:4])
Fr = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda:0"
plan = Plan(dev)
x = torch.randn(B, Cc, Fr, n, device=dev)
g = torch.ones(Cc, device=dev)
wqkv = plan.conv_weight(torch.randn(384, Cc, 1, device=dev) * 0.2)
wo = plan.conv_weight(torch.randn(Cc, 128, 1, device=dev) * 0.2)
bo = torch.zeros(Cc, device=dev)
y = plan.linattn_block(x, g, wqkv, wo, bo, g, B, Fr, n, (Cc * Fr * n, Fr * n, n), 0, 0)
s = torch.cuda.current_stream().cuda_stream
plan.run(s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    plan.run(s)
e1.record()
torch.cuda.synchronize()
print(f"linattn_block B={B} C={Cc} n={n} F={Fr}: {e0.elapsed_time(e1) / reps:.4f} ms, x = {x.numel() * 4 / 1e6:.0f} MB")
