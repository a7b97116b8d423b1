import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDC_STAMP"] = "1"
from safediffcon_amd.engine import Plan, as5
from safediffcon_amd import _lib
cin, cout, k, H, W, B = (int(v) for v in sys.argv[1:7])
dev = "cuda:0"
plan = Plan(dev, precision=2)
x = torch.randn(B, cin, H, W, device=dev); w = torch.randn(cout, cin, k, k, device=dev) * 0.05; b = torch.randn(cout, device=dev)
out = plan.conv(as5(x), plan.conv_weight(w), b, cout, (1, k, k), pad=(0, k // 2, k // 2))
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): plan.run(s)
torch.cuda.synchronize()
lib = C.CDLL(_lib.get_lib()._name)
lib.sdc_dbg_ptr.restype = C.c_void_p
p = lib.sdc_dbg_ptr()
n = 1024 * 4
buf = (C.c_ulonglong * n)()
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy(buf, C.c_void_p(p), n * 8, 2)
import numpy as np
a = np.array(buf[:n], dtype=np.float64).reshape(-1, 4)
print("blocks", len(a), "first half (stores) / second half (loads) / barrier wait / main total:", a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean())
