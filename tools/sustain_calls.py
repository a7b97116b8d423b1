#!/usr/bin/env python3
"""Which kernels of a C4 step are power-limited: every kernel instance of one smoke U-Net forward (B = 64) held alone for ~1 s,
launched back to back, with the shader clock / socket power sampled beside it (the heaviest call of each kernel instance).
usage: python tools/sustain_calls.py [batch] [seconds per kernel] [min ms: skip instances lighter than this per forward]"""
import _libsel  # noqa: F401,E402
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import safediffcon_amd as sdc  # noqa: E402
from safediffcon_amd import _lib  # noqa: E402
import stages  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
dev = "cuda:0"
torch.manual_seed(0)
net = sdc.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7).to(dev)
shape = (B, 32, 7, 64, 64)
x = torch.randn(shape, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
with torch.no_grad():
    net(x, t)
ent = net.entry(shape, B)
lib = _lib.get_lib()
plan = ent["plan"]
stream = torch.cuda.current_stream().cuda_stream
# the heaviest call of every kernel instance (one timed launch each)
best = {}
for fn, args in plan.calls:
    w = stages.classify(lib, fn, args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(*args, stream)
    e0.record()
    fn(*args, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    k = best.setdefault(w["kernel"], dict(ms=0.0, call=None, total=0.0, n=0, w=None))
    k["total"] += ms
    k["n"] += 1
    if ms > k["ms"]:
        k.update(ms=ms, call=(fn, args), w=w)
sensors = bench.GpuSensors(0)
print(f"| kernel (heaviest instance of a C4 forward, B = {B}) | launches per forward | burst ms | sustained ms | clock MHz (min) | power W | issued TFLOP/s or GB/s sustained |")
print("|---|---|---|---|---|---|---|")
for name, k in sorted(best.items(), key=lambda kv: -kv[1]["total"]):
    if k["total"] < min_ms:
        continue
    fn, args = k["call"]
    n = max(5, int(secs / (k["ms"] * 1e-3)))
    torch.cuda.synchronize()
    time.sleep(0.3)
    sensors.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(*args, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    s = sensors.stop() or {}
    w = k["w"]
    rate = f"{w['issued'] / ms / 1e9:.1f} TFLOP/s" if w["issued"] else f"{w['bytes'] / ms / 1e6:.0f} GB/s"
    print(f"| {name} | {k['n']} | {k['ms']:.3f} | {ms:.3f} | {s.get('sclk_mhz_median')} ({s.get('sclk_mhz_min')}) | {s.get('power_w_mean')} | {rate} |", flush=True)
