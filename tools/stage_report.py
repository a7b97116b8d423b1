#!/usr/bin/env python3
"""Per-stage roofline report of one U-Net forward (HIP events around every recorded call, grouped by stage).
For each stage: time, algorithmic FLOP and bytes (SURVEY 8d definitions: read input + write output + weights,
fp32), the MFMA FLOPs actually issued (Winograd forms issue fewer), achieved TFLOP/s and GB/s against the MI355X peaks.
usage: python tools/stage_report.py [burgers|tokamak|smoke] [B] [dim] > profiles/<name>.md"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import safediffcon_amd as sdc  # noqa: E402
from safediffcon_amd import _lib  # noqa: E402
import stages  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "burgers"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
lib = _lib.get_lib()
if which == "burgers":
    net, shape = sdc.Unet2D(dim=dim, channels=3, resnet_block_groups=1).to(dev), (B, 3, 16, 128)
elif which == "tokamak":
    net, shape = sdc.Unet1D(dim=dim, channels=12, resnet_block_groups=1).to(dev), (B, 12, 128)
else:
    net, shape = sdc.Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7).to(dev), (B, 32, 7, 64, 64)
if os.environ.get("SDC_PRECISION"):
    net.precision = int(os.environ["SDC_PRECISION"])
ent = net.entry(shape, B)
net.bind_cond(ent, None)
stream = torch.cuda.current_stream(dev).cuda_stream
st, kern = stages.time_plan(ent["plan"], lib, stream, reps=3)
print(stages.markdown(f"{which} U-Net forward, B={B}, dim={dim}, precision={net.precision}", st))
print("\n### by kernel template instance\n")
print("| kernel | launches | ms | MFMA-issued TFLOP/s | effective (direct-form) TFLOP/s | GB/s |")
print("|---|---|---|---|---|---|")
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"]):
    s = v["ms"] * 1e-3
    print(f"| {k} | {v['launches']} | {v['ms']:.3f} | {v['issued'] / s / 1e12:.1f} | {v['flops'] / s / 1e12:.1f} | {v['bytes'] / s / 1e9:.0f} |")
