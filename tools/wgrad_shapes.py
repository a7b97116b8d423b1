#!/usr/bin/env python3
"""print the weight-gradient calls (shapes, strides, time) of one fine-tuning step: python tools/wgrad_shapes.py [c2|c3|c4] [batch]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from safediffcon_amd import grad_ops  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"c2": 64, "c3": 64, "c4": 4}[wl]
orig = grad_ops.conv_wgrad
calls = []


def traced(g, x, k, stride=(1, 1, 1), pad=(0, 0, 0), up=(1, 1, 1), bias=True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(g, x, k, stride, pad, up, bias)
    e1.record()
    calls.append((e0, e1, tuple(g.shape), tuple(g.stride()), tuple(x.shape), tuple(x.stride()), tuple(k), tuple(stride), tuple(up)))
    return out


grad_ops.conv_wgrad = traced
import safediffcon_amd.autograd as ag  # noqa: E402
ag.grad_ops.conv_wgrad = traced
bench.finetune_step(wl, B, 0, torch.device("cuda:0"), steps=1, eager=False)
torch.cuda.synchronize()
rows = [(e0.elapsed_time(e1), *rest) for e0, e1, *rest in calls]
n = len(rows) // 2                                   # warm-up + 1 timed step: keep the second half
flt = os.environ.get('WG_K')
sel = [r for r in rows[n:] if flt is None or 'x'.join(map(str, r[5])) == flt]
print(f'{len(sel)} calls, {sum(r[0] for r in sel):.2f} ms')
for ms, gs, gst, xs, xst, k, st, up in sorted(sel, key=lambda r: -r[0])[:int(os.environ.get('WG_TOP', '12'))]:
    print(f"{ms:7.3f} ms  g {gs} {gst}  x {xs} {xst}  k {k} s {st} up {up}", flush=True)
