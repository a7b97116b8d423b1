#!/usr/bin/env python3
"""time the KSTAR rollout (sdc_kstar_rollout) for a batch of control sequences: python tools/kstar_time.py [B ...]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd import kstar  # noqa: E402

w = kstar.unflatten_weights(dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "kstar_weights.npz"))))
model = kstar.KSTARModel(w, "cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [128, 512, 2048]:
    lo, hi = torch.tensor(kstar.LOW_ACTION), torch.tensor(kstar.HIGH_ACTION)
    acts = (lo + (hi - lo) * torch.rand(B, 121, 9)).float().cuda()
    model.rollout(acts)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        model.rollout(acts)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 3 * 1e3
    print(f"B={B}: {ms:.2f} ms per rollout of 122 rows ({B / ms * 1e3:.0f} trajectories/s)", flush=True)
