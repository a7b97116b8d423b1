#!/usr/bin/env python3
"""Host-side cost of the un-captured / graph-replayed Unet2D.__call__ path.  usage: call_profile.py"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safediffcon_amd as sdc
dev = "cuda:0"
net = sdc.Unet2D(dim=64, channels=3, resnet_block_groups=1).to(dev)
x = torch.randn(256, 3, 16, 128, device=dev)
t = torch.randint(0, 1000, (256,), device=dev)
tc = t.cpu()
for _ in range(3):
    net(x, t)
torch.cuda.synchronize()
def bench(label, fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{label}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
bench("net(x, t_gpu)", lambda: net(x, t))
bench("net(x, t_cpu)", lambda: net(x, tc))
ent = net.entry(tuple(x.shape), 256)
lib = ent["plan"].lib
side = net._side
bench("graph launch only", lambda: lib.sdc_graph_launch(ent["graph"], side.cuda_stream))
s = torch.cuda.current_stream().cuda_stream
bench("call list only", lambda: (ent["cond"].run(s), ent["plan"].run(s)))
cur = torch.cuda.current_stream()
def a():
    side.wait_stream(cur); lib.sdc_graph_launch(ent["graph"], side.cuda_stream); cur.wait_stream(side)
bench("wait + graph + wait", a)
def b():
    a(); return ent["eps"].clone()
bench("  + clone", b)
def c():
    ent["x"].copy_(x); return b()
bench("  + x copy", c)
emb_host = sdc.unet.sinusoid_table(tc, net.dim).reshape(ent["emb"].shape)
def d():
    ent["emb"].copy_(emb_host); return c()
bench("  + emb H2D copy (pageable)", d)
pinned = emb_host.pin_memory()
def e():
    ent["emb"].copy_(pinned, non_blocking=True); return c()
bench("  + emb H2D copy (pinned, non_blocking)", e)
def f():
    sdc.unet.sinusoid_table(tc, net.dim); return c()
bench("  + host sinusoid only", f)
def g():
    ent["emb"].copy_(emb_host.clone()); return c()
bench("  + emb H2D copy from a fresh host tensor", g)
def h():
    ent["emb"].copy_(sdc.unet.sinusoid_table(tc, net.dim).reshape(ent["emb"].shape), non_blocking=False); return c()
bench("  + emb H2D copy from sinusoid_table()", h)
bench("net(x, t_cpu) again", lambda: net(x, tc))
net.forward_graph = False
bench("net(x, t_cpu) no graph", lambda: net(x, tc))
