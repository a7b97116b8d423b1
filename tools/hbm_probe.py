#!/usr/bin/env python3
"""What a plain device-to-device stream reaches on this GPU (read + write bytes per second), beside gn_apply:
torch copy_ / add of 2 GiB fp32 tensors.  usage: python tools/hbm_probe.py"""
import torch

dev = "cuda:0"
n = 512 * 1024 * 1024
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
z = torch.randn(n, device=dev)


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn, by in (("copy_ (read + write)", lambda: y.copy_(x), 8 * n), ("add out= (2 reads + write)", lambda: torch.add(x, z, out=y), 12 * n),
                     ("mul_ in place (read + write)", lambda: x.mul_(1.0001), 8 * n), ("fill_ (write)", lambda: y.fill_(1.0), 4 * n),
                     ("sum (read)", lambda: x.sum(), 4 * n)):
    ms = timed(fn)
    print(f"{name:32s} {ms:8.3f} ms  {by / ms / 1e9:7.2f} TB/s")
