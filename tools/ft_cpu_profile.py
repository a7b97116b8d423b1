#!/usr/bin/env python3
"""where the host time of one fine-tuning step goes: python tools/ft_cpu_profile.py [c2|c3|c4] [batch]  (cProfile, top by own time)"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"c2": 64, "c3": 64, "c4": 4}[wl]
bench.finetune_step(wl, B, 0, torch.device("cuda:0"), steps=2, eager=False)      # warm: plans, arena
pr = cProfile.Profile()
pr.enable()
out = bench.finetune_step(wl, B, 0, torch.device("cuda:0"), steps=4, eager=False)
pr.disable()
print(out["hip_ms"], "ms per step (under cProfile)")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
