"""-m gpu: the oracle's functional U-Nets executed by PyTorch-ROCm eager on the same MI355X (MIOpen / rocBLAS / aten
kernels) -- the "hipified reference" strawman SURVEY 8(d) asks to be timed beside the HIP path.  It doubles as a
full-batch parity check: at B = 256 the CPU oracle is affordable for a few samples only, the eager GPU run covers
every trajectory of the batch (eps-MSE gate 1e-5, the two fp32 implementations agree to ~1e-11)."""
import time

import pytest
import torch

import safediffcon_amd as sdc
from oracle import nets as onets
from oracle.detweights import det_params, det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _spec(net):
    return [(k, tuple(v.shape)) for k, v in net.state_dict().items()]


def _time(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def test_c2_forward_vs_torch_rocm_eager():
    net = sdc.Unet2D(dim=64, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    P = det_params(_spec(net), 11)
    net.load_state_dict(P)
    net.to(DEV)
    Pg = {k: v.to(DEV) for k, v in P.items()}
    B = 256
    x, t = det_tensor((B, 3, 16, 128), 12).to(DEV), torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(13)).to(DEV)
    with torch.no_grad():
        ref = onets.unet_burgers(Pg, x, t, dim=64)
        eps = net(x, t)
        mse = ((eps - ref) ** 2).mean().item()
        per_sample = ((eps - ref) ** 2).flatten(1).mean(1).max().item()
        ms_eager = _time(lambda: onets.unet_burgers(Pg, x, t, dim=64), 5)
        ms_call = _time(lambda: net(x, t), 5)
        # the sampler's unit of work: one guided denoising step (U-Net + guidance + update) as a hipGraph replay
        gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=1000, temporal=True, use_conv2d=True,
                                          is_condition_u0=True, is_condition_uT=True, condition_idx=10).to(DEV)
        u0, uT = det_tensor((B, 128), 41, 0.1).to(DEV), det_tensor((B, 128), 42, 0.1).to(DEV)
        S = gd.sample(batch_size=B, u_init=u0, u_final=uT, nablaJ=sdc.BurgersGuidance(0.01, 500.0, 0.05), enable_grad=False,
                      _prepare=True)
        with torch.cuda.stream(torch.cuda.Stream(DEV)):      # graph capture needs a non-default stream
            S.init()
            ms_hip = _time(S.step, 20)
            S.close()
    print(f"C2 B=256: torch-ROCm eager U-Net forward {ms_eager:.1f} ms | HIP: drop-in net(x, t) call {ms_call:.1f} ms, "
          f"whole guided denoising step as hipGraph {ms_hip:.1f} ms ({ms_eager / ms_hip:.2f}x vs the eager forward alone); "
          f"eps-MSE {mse:.2e}, worst trajectory {per_sample:.2e}")
    assert mse <= 1e-9 and per_sample <= 1e-8          # gate is 1e-5; two fp32 implementations agree far tighter
    assert ms_hip < ms_eager


def test_c4_forward_vs_torch_rocm_eager():
    net = sdc.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7)
    P = det_params(_spec(net), 31)
    net.load_state_dict(P)
    net.to(DEV)
    Pg = {k: v.to(DEV) for k, v in P.items()}
    B = 4
    x, t = det_tensor((B, 32, 7, 64, 64), 32).to(DEV), torch.tensor([5, 300, 700, 995], device=DEV)
    with torch.no_grad():
        ref = onets.unet_smoke(Pg, x, t, dim=64, dim_mults=(1, 2, 4))
        eps = net(x, t)
        mse = ((eps - ref) ** 2).mean().item()
        ms_eager = _time(lambda: onets.unet_smoke(Pg, x, t, dim=64, dim_mults=(1, 2, 4)), 2)
        ms_hip = _time(lambda: net(x, t), 2)
    print(f"C4 U-Net forward B=4: torch-ROCm eager {ms_eager:.1f} ms, HIP path {ms_hip:.1f} ms ({ms_eager / ms_hip:.1f}x); eps-MSE {mse:.2e}")
    assert mse <= 1e-9
    assert ms_hip < ms_eager
