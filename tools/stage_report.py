#!/usr/bin/env python3
"""Per-stage roofline report of one U-Net forward (HIP events around every recorded call, grouped by stage).
For each stage: time, algorithmic FLOP and bytes (SURVEY 8d definitions: read input + write output + weights,
fp32), achieved TFLOP/s and GB/s against the MI355X peaks.
usage: python tools/stage_report.py [burgers|tokamak|smoke] [B] [dim] > profiles/<name>.md"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safediffcon_amd as sdc  # noqa: E402
from safediffcon_amd import _lib  # noqa: E402

PEAK_TF, PEAK_GBS = 157.3, 8000.0
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
ent = net.entry(shape, B)
net.bind_cond(ent, None)
stream = torch.cuda.current_stream(dev).cuda_stream
e0, e1 = C.c_void_p(), C.c_void_p()
lib.sdc_event_create(C.byref(e0)); lib.sdc_event_create(C.byref(e1))


def work(fn, a):
    """(stage name, flops, bytes)"""
    n = fn.__name__
    if fn is lib.sdc_conv or fn is lib.sdc_conv_gn:
        d = a[0]._obj
        P = d.B * d.oD * d.oH * d.oW
        cin, taps = d.Cin0 + d.Cin1, d.kD * d.kH * d.kW
        nin = d.B * cin * d.iD * d.iH * d.iW
        by = 4.0 * (nin + P * d.Cout * (2 if a[5] else 1) + taps * cin * d.Cout)

        kind = f"conv {d.kD}x{d.kH}x{d.kW}" + (" (up/transposed)" if d.uH > 1 else "") + (" s2" if d.sH > 1 else "")
        if d.precision == 2 and d.kW == 3 and d.sW == 1:
            kind += " [Winograd F(2,3) along W; FLOPs = direct-form]"
        if fn is lib.sdc_conv_gn:
            kind += " + GroupNorm statistics in the epilogue"
        return kind, 2.0 * P * d.Cout * cin * taps, by
    if fn is lib.sdc_gn_finalize:
        return "groupnorm stats (finalize of the conv-epilogue sums)", 0.0, 0.0
    if fn is lib.sdc_gn_stats:
        Bb, Cc, S = a[2], a[3], a[5]
        return "groupnorm stats", 3.0 * Bb * Cc * S, 4.0 * Bb * Cc * S
    if fn is lib.sdc_gn_apply:
        Bb, Cc, S = a[11], a[12], a[14]
        return "groupnorm apply+SiLU(+res)", 8.0 * Bb * Cc * S, 4.0 * Bb * Cc * S * (3 if a[9] else 2)
    if fn is lib.sdc_chan_norm:
        Bb, Cc, S = a[4], a[5], a[6]
        return "channel LN/RMS(+res)", 8.0 * Bb * Cc * S, 4.0 * Bb * Cc * S * (3 if a[2] else 2)
    if fn is lib.sdc_linattn:
        outer, inner, heads, nn = a[3], a[4], a[5], a[6]
        seqs = outer * inner * heads
        return "linear attention core", seqs * nn * (2 * 2 * 32 * 32 + 10 * 32), 4.0 * seqs * nn * 32 * 4
    if fn is lib.sdc_linattn_block:
        outer, inner, Cc, nn = a[8], a[9], a[10], a[11]
        toks = outer * inner * nn
        # reference-equivalent work: qkv 1x1 (C -> 384), attention core (4 heads x two 32x32 products), out 1x1 (128 -> C),
        # two channel norms; bytes: x read once, y written once (what the fused block is priced against)
        return "fused LinearAttention block (norm+qkv+core+out+norm+res)", toks * (2.0 * Cc * 384 + 4 * 2 * 2 * 32 * 32 + 2.0 * 128 * Cc + 16 * Cc), 8.0 * toks * Cc
    if fn is lib.sdc_tattn_block:
        outer, inner, Cc, ntok = a[7], a[8], a[9], a[10]
        toks = outer * inner * ntok
        return ("fused temporal-attention block (norm+qkv+rotary/bias attention+out+res)",
                toks * (2.0 * Cc * 384 + 4 * 2 * 2 * ntok * 32 + 2.0 * 128 * Cc + 8 * Cc), 8.0 * toks * Cc)
    if fn is lib.sdc_attn:
        outer, inner, heads, nt = a[4], a[5], a[6], a[7]
        seqs = outer * inner * heads
        st = a[11]
        return ("temporal attention core" if st != 1 else "softmax attention core"), seqs * 4.0 * nt * nt * 32, 4.0 * seqs * nt * 32 * 4
    return n, 0.0, 0.0


tot = {}
for fn, args in ent["plan"].calls:
    reps = 3
    fn(*args, stream)
    lib.sdc_event_record(e0, stream)
    for _ in range(reps):
        fn(*args, stream)
    lib.sdc_event_record(e1, stream)
    ms = C.c_float()
    lib.sdc_event_elapsed_ms(e0, e1, C.byref(ms))
    name, fl, by = work(fn, args)
    t = tot.setdefault(name, [0, 0.0, 0.0, 0.0])
    t[0] += 1; t[1] += ms.value / reps; t[2] += fl; t[3] += by
total_ms = sum(v[1] for v in tot.values())
print(f"## {which} U-Net forward, B={B}, dim={dim}: {total_ms:.2f} ms (sum of stage times, HIP events, MI355X)\n")
print("| stage | launches | ms | share | GFLOP | TFLOP/s | %fp32-MFMA peak (157.3) | MB (algorithmic) | GB/s | %HBM peak (8 TB/s) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, (n, ms, fl, by) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    tf = fl / (ms * 1e-3) / 1e12 if ms else 0
    gbs = by / (ms * 1e-3) / 1e9 if ms else 0
    print(f"| {k} | {n} | {ms:.3f} | {100 * ms / total_ms:.1f}% | {fl / 1e9:.1f} | {tf:.1f} | {100 * tf / PEAK_TF:.1f}% | {by / 1e6:.0f} | {gbs:.0f} | {100 * gbs / PEAK_GBS:.1f}% |")
