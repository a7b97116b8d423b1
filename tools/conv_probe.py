#!/usr/bin/env python3
"""Per-call timing of every sdc_conv launch in a U-Net plan (HIP events), for kernel tuning.
usage: python tools/conv_probe.py [burgers|tokamak|smoke] [B] [dim]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safediffcon_amd as sdc  # noqa: E402
from safediffcon_amd import _lib  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stages  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "burgers"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
lib = _lib.get_lib()
if which == "burgers":
    net = sdc.Unet2D(dim=dim, channels=3, resnet_block_groups=1).to(dev)
    shape = (B, 3, 16, 128)
elif which == "tokamak":
    net = sdc.Unet1D(dim=dim, channels=12, resnet_block_groups=1).to(dev)
    shape = (B, 12, 128)
else:
    net = sdc.Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7).to(dev)
    shape = (B, 32, 7, 64, 64)
net.precision = int(os.environ.get('SDC_PRECISION', '4'))
ent = net.entry(shape, B)
net.bind_cond(ent, None)
stream = torch.cuda.current_stream(dev).cuda_stream
e0, e1 = C.c_void_p(), C.c_void_p()
lib.sdc_event_create(C.byref(e0)); lib.sdc_event_create(C.byref(e1))
tot = {}
print(f"{'instance':34s} {'Cin':>5s} {'Cout':>5s} {'k':>7s} {'out':>14s} {'ms':>8s} {'TF/s':>7s}")
for fn, args in ent["plan"].calls:
    name = fn.__name__
    reps = 5
    fn(*args, stream)
    lib.sdc_event_record(e0, stream)
    for _ in range(reps):
        fn(*args, stream)
    lib.sdc_event_record(e1, stream)
    ms = C.c_float()
    lib.sdc_event_elapsed_ms(e0, e1, C.byref(ms))
    t = ms.value / reps
    tot[name] = tot.get(name, 0.0) + t
    if fn is lib.sdc_conv or fn is lib.sdc_conv_gn:
        d = args[0]._obj
        w = stages.classify(lib, fn, args)
        print(f"{w['kernel']:38s} {d.Cin0 + d.Cin1:5d} {d.Cout:5d} {d.kD}x{d.kH}x{d.kW:<3d} "
              f"{d.oD}x{d.oH}x{d.oW:<6d} {t:8.4f} {w['flops'] / t / 1e9:7.1f} (issued {w['issued'] / t / 1e9:6.1f})")
print({k: round(v, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()), 3))
