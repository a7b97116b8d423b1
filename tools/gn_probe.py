#!/usr/bin/env python3
"""time sdc_gn_apply on the C4 level-0 tensor.  usage: gn_probe.py [B]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
plan = Plan(dev)
x = torch.randn(B, 64, 32, 64, 64, device=dev)
r = torch.randn(B, 64, 32, 64, 64, device=dev)
g, b = torch.ones(64, device=dev), torch.zeros(64, device=dev)
plan.gn_silu(x, g, b, 8)
plan.gn_silu(x, g, b, 8, residual=r)
s = torch.cuda.current_stream().cuda_stream
plan.run(s)
for idx, name in ((1, "apply"), (3, "apply+res")):
    fn, args = plan.calls[idx]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(*args, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    by = x.numel() * 4 * (3 if idx == 3 else 2)
    print(f"gn_{name} B={B}: {ms:.3f} ms  {by / ms / 1e9:.2f} TB/s")
