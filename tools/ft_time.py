#!/usr/bin/env python3
"""time one fine-tuning step (bench.finetune_step) of a workload: python tools/ft_time.py [c2|c3|c4] [batch]"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if os.environ.get("FT_NO_OVERLAP") == "1":          # A/B of the wgrad / dgrad fork in ConvFn.backward
    import safediffcon_amd.autograd as _ag
    _ag.OVERLAP_WGRAD = False
if os.environ.get("FT_NO_SPLIT") == "1":
    import safediffcon_amd.autograd as _ag2
    _ag2.SPLIT_SMALL_GRIDS = False
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"c2": 64, "c3": 64, "c4": 4}[wl]
print(json.dumps(bench.finetune_step(wl, B, 0, torch.device("cuda:0"), eager=os.environ.get("FT_NO_EAGER") != "1")))
