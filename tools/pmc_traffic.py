#!/usr/bin/env python3
"""Workload for the FETCH_SIZE / WRITE_SIZE / SQ PMC passes (rocprofv3 --kernel-trace --pmc ... -- python3 tools/pmc_traffic.py):
 (1) a calibration launch with a KNOWN byte count (sdc_act over n floats reads 4n and writes 4n bytes, dword per lane),
 (2) the kernels of the C4 step the bench prices, a few launches each, at the C4 shapes (B = 64 by default):
       conv_wg3s<64> 64->64  3x3x3 at (B,64,32,64,64)   + GroupNorm statistics          (level-0 ResnetBlock conv)
       conv_wg3s<32> 64->128 3x3x3 at (B,64,32,32,32)
       conv_wg3<32>  128->128 3x3x3 at (B,128,32,32,32)
       conv_wg3<16>  256->256 3x3x3 at (B,256,32,16,16)
       gn_apply      GroupNorm apply + SiLU on the level-0 tensor, in place
       ta_block      fused temporal-attention block at width 64 on the level-0 tensor
       conv_pw       1x1x1 conv 128->384 at (B,128,32,32,32) (to_qkv of the width-128 temporal attention)
       gn_pw_out     the last GroupNorm apply inside final_conv 64->7 (round 6)
 It writes the algorithmic bytes of every case to <out dir>/pmc_cases.json for tools/pmc_to_json.py.
usage: python3 tools/pmc_traffic.py [batch] [cases.json]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.engine import Plan, as5  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cases_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pmc_cases.json"
dev = "cuda:0"
plan = Plan(dev, precision=4)
n = 64 * 1024 * 1024                       # 256 MiB in, 256 MiB out: beyond the 256 MiB Infinity Cache together
xa = torch.randn(n, device=dev)
ya = torch.empty(n, device=dev)
plan.act(xa, 0, out=ya)
cases = {"act_kernel": dict(read=4 * n, write=4 * n, shape=f"sdc_act over {n} floats")}


def conv3(cin, cout, sp, tag):
    x = torch.randn(B, cin, *sp, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.03
    b = torch.randn(cout, device=dev)
    out = plan.conv(as5(x), plan.conv_weight(w), b, cout, (3, 3, 3), pad=(1, 1, 1), gn_groups=8)
    cases[tag] = dict(algorithmic=4 * (x.numel() + out.numel() + w.numel()), shape=f"{cin}->{cout} 3x3x3 at ({B},{cin},{','.join(map(str, sp))}) + GN statistics")
    return x, out


x0, out0 = conv3(64, 64, (32, 64, 64), "conv_wg3s_kernel<64")          # round 5: rows of 64 and 32 run the two-workgroups-per-CU form
conv3(128, 128, (32, 32, 32), "conv_wg3s_kernel<32")                  # (one shape per instance: the counters are averaged per kernel name)
conv3(256, 256, (32, 16, 16), "conv_wg3_kernel<16")
gm, bt = torch.ones(64, device=dev), torch.zeros(64, device=dev)
plan.gn_silu(out0, gm, bt, 8)
cases["gn_apply_kernel"] = dict(algorithmic=8 * out0.numel(), shape=f"GroupNorm apply + SiLU in place on ({B},64,32,64,64)")
g = torch.ones(64, device=dev)
wqkv = plan.conv_weight(torch.randn(384, 64, 1, device=dev) * 0.1)
wo = plan.conv_weight(torch.randn(64, 128, 1, device=dev) * 0.1)
rot = torch.randn(32 * 16 * 2, device=dev)
bias = torch.randn(4 * 32 * 32, device=dev)
plan.tattn_block(x0, g, wqkv, wo, rot, bias)
cases["ta_block_kernel"] = dict(algorithmic=8 * x0.numel(), shape=f"fused temporal-attention block, width 64, on ({B},64,32,64,64)")
x1 = torch.randn(B, 128, 32, 32, 32, device=dev)
w1 = torch.randn(384, 128, 1, 1, 1, device=dev) * 0.1
o1 = plan.conv(as5(x1), plan.conv_weight(w1), None, 384, (1, 1, 1))
cases["conv_pw_kernel<128"] = dict(algorithmic=4 * (x1.numel() + o1.numel() + w1.numel()), shape=f"1x1x1 conv 128->384 at ({B},128,32,32,32)")
# round 6: the last ResnetBlock's GroupNorm apply + SiLU + residual inside the 1x1x1 output conv (64 -> 7, frame-major eps)
hf = torch.randn(B, 64, 32, 64, 64, device=dev)
rf = torch.randn(B, 64, 32, 64, 64, device=dev)
stf = plan.gn_stats_deferred(hf, 8)
wf, bf = torch.randn(7 * 64, device=dev) * 0.1, torch.randn(7, device=dev)
ef = torch.empty(B, 32, 7, 64, 64, device=dev).permute(0, 2, 1, 3, 4)
assert plan.gn_pointwise_out(hf, (stf, gm, bt, 8, rf), wf, bf, ef) is not None
cases["gn_pw_out_kernel"] = dict(algorithmic=4 * (hf.numel() + rf.numel() + ef.numel()),
                                 shape=f"GroupNorm apply + SiLU + residual inside final_conv 64->7 on ({B},64,32,64,64)")
# the dominant kernels of the other two single-GPU workloads, at their own batch sizes
x2 = torch.randn(256, 64, 16, 128, device=dev)
w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
o2 = plan.conv(as5(x2), plan.conv_weight(w2), torch.randn(64, device=dev), 64, (1, 3, 3), pad=(0, 1, 1), gn_groups=1)
cases["conv_wg2s_kernel<128"] = dict(algorithmic=4 * (x2.numel() + o2.numel() + w2.numel()), workload="c2",
                                    shape="C2: 64->64 3x3 at (256,64,16,128) + GN statistics (Burgers level 0)")
x3 = torch.randn(128, 256, 128, device=dev)
w3 = torch.randn(256, 256, 3, device=dev) * 0.03
o3 = plan.conv(as5(x3), plan.conv_weight(w3), torch.randn(256, device=dev), 256, (1, 1, 3), pad=(0, 0, 1))
cases["conv_wg_kernel<128, 128, 4, 2, 16, 512"] = dict(algorithmic=4 * (x3.numel() + o3.numel() + w3.numel()), workload="c3",
                                                      shape="C3: 256->256 Conv1d k3 at (128,256,128) (tokamak level 0), F(2,3)")
x4 = torch.randn(128, 2048, 16, device=dev)
w4 = torch.randn(2048, 2048, 3, device=dev) * 0.01
plan5 = Plan(dev, precision=5)
o4 = plan5.conv(as5(x4), plan5.conv_weight(w4), torch.randn(2048, device=dev), 2048, (1, 1, 3), pad=(0, 0, 1))
cases["conv_f43_kernel"] = dict(algorithmic=4 * (x4.numel() + o4.numel() + w4.numel()), workload="c3",
                                shape="C3 (precision 5): 2048->2048 Conv1d k3 at (128,2048,16) (tokamak mid level), F(4,3)")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    plan.run(s)
    plan5.run(s)
torch.cuda.synchronize()
os.makedirs(os.path.dirname(cases_path) or ".", exist_ok=True)
json.dump(dict(batch=B, cases=cases), open(cases_path, "w"), indent=1)
print("cases:", list(cases))
