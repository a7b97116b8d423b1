#!/usr/bin/env python3
"""time sdc_chan_norm on the C4 attention pre-norm tensors and check it against torch.  usage: cn_probe.py"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan
dev = "cuda:0"
for (B, C, F, H) in ((64, 128, 32, 32), (64, 256, 32, 16), (64, 512, 32, 8), (64, 64, 32, 64)):
    plan = Plan(dev)
    x = torch.randn(B, C, F, H, H, device=dev) + 0.5
    r = torch.randn_like(x)
    g = torch.rand(C, device=dev) + 0.5
    y0 = plan.chan_norm(x, g, 0)
    y1 = plan.chan_norm(x, g, 0, residual=r)
    s = torch.cuda.current_stream().cuda_stream
    plan.run(s)
    mean = x.mean(1, keepdim=True)
    var = x.var(1, unbiased=False, keepdim=True)
    ref = (x - mean) / (var + 1e-5).sqrt() * g.view(1, C, 1, 1, 1)
    e0, e1 = (y0 - ref).abs().max().item(), (y1 - ref - r).abs().max().item()
    for idx, name in ((0, "ln"), (1, "ln+res")):
        fn, args = plan.calls[idx]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn(*args, s)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        by = x.numel() * 4 * (3 if idx else 2)
        print(f"chan_norm {name} B={B} C={C} S={F*H*H}: {ms:.3f} ms  {by / ms / 1e9:.2f} TB/s   max|err| {e0:.2e} {e1:.2e}")
