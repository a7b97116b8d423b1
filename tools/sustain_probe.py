#!/usr/bin/env python3
"""One conv shape run back to back for seconds with the shader clock / socket power sampled beside it: does the kernel alone, held
long enough for the power management to settle, run at its burst time (tools/wg_probe.py: 5 launches) or at its in-step time?
usage: python tools/sustain_probe.py [shape index of tools/wg_probe.py's table] [seconds]"""
import _libsel  # noqa: F401,E402
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safediffcon_amd.engine import Plan, as5  # noqa: E402
import bench  # noqa: E402

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
B = 64
shapes = [("c4 L0 64->64", 64, 0, 64, (32, 64, 64)), ("c4 L0 64+64->64", 64, 64, 64, (32, 64, 64)), ("c4 L1 128->128", 128, 0, 128, (32, 32, 32)),
          ("c4 L1 64->128", 64, 0, 128, (32, 32, 32)), ("c4 L2 256->256", 256, 0, 256, (32, 16, 16)), ("c4 L2 256+256->128", 256, 256, 128, (32, 16, 16))]
name, c0, c1, co, sp = shapes[idx]
dev = "cuda:0"
x = torch.randn(B, c0, *sp, device=dev)
x1 = torch.randn(B, c1, *sp, device=dev) if c1 else None
w = torch.randn(co, c0 + c1, 3, 3, 3, device=dev) * 0.05
b = torch.randn(co, device=dev)
plan = Plan(dev, precision=4)
plan.conv(as5(x), plan.conv_weight(w), b, co, (3, 3, 3), x1=None if x1 is None else as5(x1), pad=(1, 1, 1), gn_groups=8)
s = torch.cuda.current_stream().cuda_stream
plan.run(s)
torch.cuda.synchronize()
sensors = bench.GpuSensors(0)
for label, n in (("burst", 5), ("sustained", None)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if n is None:
        n = max(10, int(secs / (last_ms * 1e-3)))
        sensors.start()
    torch.cuda.synchronize()
    time.sleep(0.5 if label == "burst" else 0.0)
    e0.record()
    for _ in range(n):
        plan.run(s)
    e1.record()
    torch.cuda.synchronize()
    last_ms = e0.elapsed_time(e1) / n
    extra = sensors.stop() if label == "sustained" else None
    print(f"{name} B={B}: {label:9s} {n:5d} launches  {last_ms:.4f} ms per launch" + (f"  clock {extra['sclk_mhz_median']} MHz (min {extra['sclk_mhz_min']})  power {extra['power_w_mean']} W" if extra else ""), flush=True)
