#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (read-only at
/root/reference) on seeded inputs.  Build-container only: the reference never
travels to the GPU box; the fixtures (data: inputs + expected outputs) do.

    python oracle/make_goldens.py            # all three trees (one subprocess each)
    python oracle/make_goldens.py burgers    # one tree

What is pinned:
  * schedule tables (cosine / sigmoid) bit-exact
  * U-Net epsilon at tiny configs + a handful of intermediate stage outputs
  * T=8 p_sample_loop trajectories, guided and calibration variants, noise injected
    by patching torch.randn / torch.randn_like while the reference loop runs
  * guidance-gradient, weight-normalisation and conformal-quantile KATs
  * one FULL-SCHEDULE (T = 1000) guided DDPM trajectory per tree (final state), gen_long
  * fine-tuning loss (p_losses, per sample) and gradient digests of every parameter, gen_grad
Weights are NOT stored: both sides rebuild them with oracle.detweights.det_params
from the (key, shape) list stored in the fixture.
"""
import contextlib
import os
import subprocess
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.environ.get("SDC_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")     # (override: regeneration checks)
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle.detweights import det_params, det_noise, det_tensor  # noqa: E402


@contextlib.contextmanager
def injected_noise(noise):
    """Route the reference's torch.randn / randn_like draws to noise(i)."""
    state = {"i": 0}
    real_randn, real_like = torch.randn, torch.randn_like

    def _next(shape):
        z = noise(state["i"])
        assert tuple(z.shape) == tuple(shape), (z.shape, shape)
        state["i"] += 1
        return z

    def randn(*shape, **kw):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        return _next(shape)

    torch.randn, torch.randn_like = randn, (lambda x, **kw: _next(x.shape))
    try:
        yield state
    finally:
        torch.randn, torch.randn_like = real_randn, real_like


def spec_of(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def load_det(module, seed):
    spec = spec_of(module)
    P = det_params(spec, seed)
    module.load_state_dict(P)
    module.eval()
    return spec


def spec_arrays(spec):
    return {"spec_keys": np.array([k for k, _ in spec]),
            "spec_shapes": np.array([",".join(map(str, s)) for _, s in spec])}


def grab_stages(module, names):
    got, hooks = {}, []
    mods = dict(module.named_modules())
    for n in names:
        hooks.append(mods[n].register_forward_hook(
            lambda m, i, o, n=n: got.__setitem__(n, o.detach().clone())))
    return got, hooks


def stub_modules(*names):
    for n in names:
        if n not in sys.modules:
            import importlib.machinery
            m = types.ModuleType(n)
            m.__spec__ = importlib.machinery.ModuleSpec(n, None)
            sys.modules[n] = m


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ----------------------------------------------------------------------------

def gen_burgers():
    sys.path.insert(0, os.path.join(REF, "1D"))
    from model.unet import Unet2D
    from model.diffusion import GaussianDiffusion

    # schedules (1000 steps)
    gd = GaussianDiffusion(Unet2D(dim=8, channels=3, resnet_block_groups=1), seq_length=(16, 128),
                           temporal=True, use_conv2d=True)
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2"]
    save("schedule_cosine", **{n: getattr(gd, n) for n in names})
    gl = GaussianDiffusion(Unet2D(dim=8, channels=3, resnet_block_groups=1), seq_length=(16, 128),
                           temporal=True, use_conv2d=True, beta_schedule="linear")
    save("schedule_linear", **{n: getattr(gl, n) for n in names})

    # U-Net KAT
    dim = 8
    net = Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    spec = load_det(net, seed=100)
    x = det_tensor((2, 3, 16, 128), 101)
    t = torch.tensor([7, 512])
    stages = ["init_conv", "downs.0.0.block1", "downs.0.0", "downs.0.2", "downs.0.3", "downs.3.3",
              "mid_block1", "mid_attn", "ups.0.3", "ups.3.2", "final_res_block"]
    got, hooks = grab_stages(net, stages)
    with torch.no_grad():
        eps = net(x, t)
        temb = net.time_mlp(t)
    for h in hooks:
        h.remove()
    save("burgers_unet", dim=dim, x=x, t=t, eps=eps, temb=temb, **spec_arrays(spec),
         **{"stage:" + k: v for k, v in got.items()})

    # guidance / conformal helpers of the reference (need stubs for absent pip packages
    # that utils.common pulls in at import time; none is used by the functions called here)
    stub_modules("h5py", "tensorboardX", "ema_pytorch", "IPython")
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["ema_pytorch"].EMA = object
    sys.modules["IPython"].embed = None
    ref_guid = None
    try:
        from utils.guidance import get_finetune_guidance, calculate_guidance
        from inference.guidance import get_weight, normalize_weights
        from inference.conformal import ConformalCalculator
        ref_guid = True
    except Exception as e:  # noqa: BLE001
        print("!! 1D guidance import failed, KAT skipped:", repr(e))

    cfg = types.SimpleNamespace(use_max_safety=True, u_bound=0.8, guidance_weights={"w_score": 500.0}, InfFT_Q=None)
    if ref_guid:
        for tag, ums in (("mean", True), ("amax", False)):
            cfg.use_max_safety = ums
            xs = det_tensor((6, 3, 16, 128), 110, scale=0.1)
            xs[:3, 2] += 0.07            # make the hinge active for half the batch
            Q = 0.01
            x_ = xs.clone().requires_grad_()
            grad = get_finetune_guidance(cfg, x_, Q)
            J = calculate_guidance(xs, Q, cfg)
            w = get_weight(xs, Q, cfg)
            save(f"burgers_guidance_{tag}", x=xs, Q=Q, w_score=500.0, u_bound=0.8, grad=grad, J=J, weight=w)
        # normalize_weights edge cases + quantile
        cases = {
            "plain": torch.tensor([0.5, 1.5, 0.25, 3.0, 0.0, 1.0]),
            "inf": torch.tensor([0.5, float("inf"), 0.25, 3.0, float("inf"), 1.0]),
            "zero": torch.zeros(6),
        }
        out = {}
        for k, v in cases.items():
            out["nw_in_" + k] = v.clone()
            out["nw_out_" + k] = normalize_weights(v.clone())
        sc = det_tensor((16,), 120).abs()
        for a in (0.98, 0.9, 0.5, 0.05):
            out[f"q_{a}"] = ConformalCalculator.calculate_quantile(None, sc, None, None, a)
        out["q_scores"] = sc
        save("burgers_conformal", **out)

    # trajectories, T = 8
    T = 8
    for tag, kw in (("guided", dict(guidance=True)), ("calib", dict(guidance=False))):
        net = Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        load_det(net, seed=100)
        gd = GaussianDiffusion(net, seq_length=(16, 128), timesteps=T, temporal=True, use_conv2d=True,
                               is_condition_u0=True, is_condition_uT=True, condition_idx=10,
                               train_on_padded_locations=False)
        B = 2
        u0 = det_tensor((B, 128), 130, 0.1, -0.1, 0.3)
        uT = det_tensor((B, 128), 131, 0.1, -0.1, 0.3)
        noise = det_noise((B, 3, 16, 128), 1000)
        Q = 0.01
        cfg.use_max_safety = True
        cfg.u_bound = 0.05   # tiny bound so the hinge is active on random nets
        if kw["guidance"]:
            if ref_guid:
                nablaJ = lambda x: get_finetune_guidance(cfg, x, Q)  # noqa: E731
            else:
                from oracle.samplers import burgers_guidance
                nablaJ = burgers_guidance(Q, 500.0, 0.05)
            with injected_noise(noise) as st:
                out = gd.sample(batch_size=B, clip_denoised=True, u_init=u0, u_final=uT, guidance_u0=True,
                                nablaJ=nablaJ, J_scheduler=lambda t: 1.0, w_scheduler=None, enable_grad=False)
            save("burgers_traj_guided", out=out, u0=u0, uT=uT, Q=Q, w_score=500.0, u_bound=0.05, T=T,
                 draws=st["i"], noise_seed=1000, dim=dim, weight_seed=100)
        else:
            wgt = det_tensor((B, 16, 128), 132, 0.05)
            with injected_noise(noise) as st:
                out = gd.sample(batch_size=B, clip_denoised=True, guidance_u0=False, u_init=u0, u_final=uT,
                                w_groundtruth=wgt, nablaJ=None, J_scheduler=None, w_scheduler=None,
                                enable_grad=False)
            save("burgers_traj_calib", out=out, u0=u0, uT=uT, w_gt=wgt, T=T, draws=st["i"], noise_seed=1000,
                 dim=dim, weight_seed=100)
    if ref_guid:
        _ddim_burgers(Unet2D, GaussianDiffusion, cfg, get_finetune_guidance, dim)
    # evaluation rollout: the reference's finite-difference solver on seeded inputs
    try:
        from data.generate_burgers import burgers_numeric_solve_free
        u0 = det_tensor((3, 128), 140, 0.5)
        ff = det_tensor((3, 10, 128), 141, 1.0)
        traj = burgers_numeric_solve_free(u0, ff, visc=0.01, T=1.0, dt=1e-4, num_t=10)
        save("burgers_rollout", u0=u0, f=ff, traj=traj)
    except Exception as e:  # noqa: BLE001
        print("!! burgers solver import failed:", repr(e))


def _ddim_burgers(Unet2D, GaussianDiffusion, cfg, get_finetune_guidance, dim):
    T, S, B = 20, 5, 2
    net = Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    load_det(net, seed=100)
    gd = GaussianDiffusion(net, seq_length=(16, 128), timesteps=T, sampling_timesteps=S, ddim_sampling_eta=1.0,
                           temporal=True, use_conv2d=True, is_condition_u0=True, is_condition_uT=True, condition_idx=10,
                           train_on_padded_locations=False)
    u0 = det_tensor((B, 128), 130, 0.1, -0.1, 0.3)
    uT = det_tensor((B, 128), 131, 0.1, -0.1, 0.3)
    noise = det_noise((B, 3, 16, 128), 1500)
    Q = 0.01
    cfg.use_max_safety, cfg.u_bound = True, 0.05
    with injected_noise(noise) as st:
        out = gd.sample(batch_size=B, clip_denoised=True, u_init=u0, u_final=uT, guidance_u0=True,
                        nablaJ=lambda x: get_finetune_guidance(cfg, x, Q), J_scheduler=lambda t: 1.0, w_scheduler=None,
                        enable_grad=False)
    save("burgers_ddim_guided", out=out, u0=u0, uT=uT, Q=Q, w_score=500.0, u_bound=0.05, T=T, S=S, eta=1.0, draws=st["i"],
         noise_seed=1500, dim=dim, weight_seed=100)
    wgt = det_tensor((B, 16, 128), 132, 0.05)
    with injected_noise(noise) as st:
        out = gd.sample(batch_size=B, clip_denoised=True, guidance_u0=False, u_init=u0, u_final=uT, w_groundtruth=wgt,
                        nablaJ=None, J_scheduler=None, w_scheduler=None, enable_grad=False)
    save("burgers_ddim_calib", out=out, u0=u0, uT=uT, w_gt=wgt, T=T, S=S, eta=1.0, draws=st["i"], noise_seed=1500, dim=dim,
         weight_seed=100)


def gen_tokamak():
    sys.path.insert(0, os.path.join(REF, "tokamak"))
    from model.unet import Unet1D
    from model.diffusion import GaussianDiffusion

    dim = 8
    net = Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
    spec = load_det(net, seed=200)
    x = det_tensor((2, 12, 128), 201)
    t = torch.tensor([3, 900])
    stages = ["init_conv", "downs.0.0", "downs.0.2", "downs.0.3", "downs.3.3", "mid_attn", "ups.0.3", "ups.3.2",
              "final_res_block"]
    got, hooks = grab_stages(net, stages)
    with torch.no_grad():
        eps = net(x, t)
    for h in hooks:
        h.remove()
    save("tokamak_unet", dim=dim, x=x, t=t, eps=eps, **spec_arrays(spec), **{"stage:" + k: v for k, v in got.items()})

    # guidance: reference module needs the HF dataset for its target; restate-free route:
    # import calculate_loss through GradientGuidance.__new__ (skip __init__) and set fields.
    ref_guid = False
    stub_modules("h5py", "tensorboardX", "ema_pytorch", "IPython")
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["ema_pytorch"].EMA = object
    sys.modules["IPython"].embed = None
    # utils.metrics imports the TensorFlow KSTAR simulator (evaluation only, TF absent): stub that one module
    stub_modules("kstar_solver")
    sys.modules["kstar_solver"].KSTARSolver = object
    try:
        from utils.guidance import GradientGuidance, calculate_weight, normalize_weights
        from inference.conformal import ConformalCalculator
        ref_guid = True
    except Exception as e:  # noqa: BLE001
        print("!! tokamak guidance import failed, KAT skipped:", repr(e))

    B, nt = 4, 122
    target = det_tensor((B, 3, nt), 210, 0.3) + 1.0
    def make_guid(w_obj, w_safe, scaler, thr, Q):
        g = GradientGuidance.__new__(GradientGuidance)
        g.w_obj, g.w_safe, g.guidance_scaler, g.Q, g.safety_threshold, g.nt = w_obj, w_safe, scaler, Q, thr, nt
        g.state_target = target
        return g
    if ref_guid:
        xs = det_tensor((B, 12, 128), 211, 0.3) + 0.5
        for tag, (wo, ws, sc, thr, Q) in {"safe": (0.0, 1.0, 0.01, 4.98, 0.0), "mixed": (0.7, 0.3, 0.5, 3.6, 0.1)}.items():
            g = make_guid(wo, ws, sc, thr, Q)
            grad = g(xs)
            J = g.calculate_loss(xs)
            w = calculate_weight(xs, target, nt, Q, thr, wo, ws, sc)
            save(f"tokamak_guidance_{tag}", x=xs, target=target, w_obj=wo, w_safe=ws, scaler=sc, thr=thr, Q=Q,
                 grad=grad, J=J, weight=w)

    T = 8
    net = Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
    load_det(net, seed=200)
    gd = GaussianDiffusion(net, seq_length=128, nt=nt, timesteps=T, use_conv2d=False, temporal=False,
                           guidance_u0=True, is_condition_u0=True, is_condition_uT=True)
    B = 2
    u0 = det_tensor((B, 3), 220, 0.1) + 0.6
    uT = det_tensor((B, 2, nt), 221, 0.1) + 0.6
    noise = det_noise((B, 12, 128), 2000)
    target2 = target[:B]
    if ref_guid:
        g = make_guid(0.0, 1.0, 0.01, 4.98, 0.0)
        g.state_target = target2
        nablaJ = g
    else:
        from oracle.samplers import tokamak_guidance
        nablaJ = tokamak_guidance(target2, nt, 0.0, 4.98, 0.0, 1.0, 0.01)
    with injected_noise(noise) as st:
        out = gd.sample(batch_size=B, clip_denoised=True, guidance_u0=True, u_init=u0, u_final=uT, nablaJ=nablaJ,
                        J_scheduler=lambda t: 1.0, w_scheduler=None, enable_grad=False)
    save("tokamak_traj_guided", out=out, u0=u0, uT=uT, target=target2, T=T, draws=st["i"], noise_seed=2000,
         dim=dim, weight_seed=200, w_obj=0.0, w_safe=1.0, scaler=0.01, thr=4.98, Q=0.0)
    # unguided calibration-style (guidance_u0=False, no w_groundtruth: that path crashes in the reference)
    with injected_noise(noise) as st:
        out = gd.sample(batch_size=B, clip_denoised=True, guidance_u0=False, u_init=u0, u_final=uT, nablaJ=None,
                        J_scheduler=None, w_scheduler=None, enable_grad=False)
    save("tokamak_traj_calib", out=out, u0=u0, uT=uT, T=T, draws=st["i"], noise_seed=2000, dim=dim, weight_seed=200)
    # the reference DDPM + w_groundtruth bug -- record that it raises
    try:
        with injected_noise(noise):
            gd.sample(batch_size=B, guidance_u0=False, u_init=u0, u_final=uT, w_groundtruth=torch.zeros(B, 9, 128),
                      nablaJ=None, enable_grad=False)
        raised = "none"
    except Exception as e:  # noqa: BLE001
        raised = type(e).__name__
    save("tokamak_wgt_bug", raised=raised)
    # DDIM (eta = 1, 5 of 20 steps): guided, and calibration-style with w_groundtruth (works on this path)
    T2, S2 = 20, 5
    net = Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
    load_det(net, seed=200)
    gd2 = GaussianDiffusion(net, seq_length=128, nt=nt, timesteps=T2, sampling_timesteps=S2, ddim_sampling_eta=1.0,
                            use_conv2d=False, temporal=False, guidance_u0=True, is_condition_u0=True, is_condition_uT=True)
    noise2 = det_noise((B, 12, 128), 2500)
    with injected_noise(noise2) as st:
        out = gd2.sample(batch_size=B, clip_denoised=True, guidance_u0=True, u_init=u0, u_final=uT, nablaJ=nablaJ,
                         J_scheduler=lambda t: 1.0, w_scheduler=None, enable_grad=False)
    save("tokamak_ddim_guided", out=out, u0=u0, uT=uT, target=target2, T=T2, S=S2, eta=1.0, draws=st["i"], noise_seed=2500,
         dim=dim, weight_seed=200, w_obj=0.0, w_safe=1.0, scaler=0.01, thr=4.98, Q=0.0)
    wgt = det_tensor((B, 9, 128), 222, 0.2)
    with injected_noise(noise2) as st:
        out = gd2.sample(batch_size=B, clip_denoised=True, guidance_u0=False, u_init=u0, u_final=uT, w_groundtruth=wgt,
                         nablaJ=None, J_scheduler=None, w_scheduler=None, enable_grad=False)
    save("tokamak_ddim_calib", out=out, u0=u0, uT=uT, w_gt=wgt, T=T2, S=S2, eta=1.0, draws=st["i"], noise_seed=2500, dim=dim,
         weight_seed=200)


def gen_smoke():
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, os.path.join(REF, "2d"))
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D, RelativePositionBias
    from ddpm.diffusion_2d import GaussianDiffusion

    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7)
    print("Unet3D_with_Conv3D(64,(1,2,4),7): params", sum(p.numel() for p in net.parameters()),
          "keys", len(net.state_dict()))
    full_spec = spec_of(net)
    gd = GaussianDiffusion(net, image_size=64, frames=32, timesteps=1000, loss_type="l2")
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2"]
    save("schedule_sigmoid", **{n: getattr(gd, n) for n in names})
    save("smoke_fullspec", **spec_arrays(full_spec))
    del net, gd

    rpb = RelativePositionBias(heads=4, max_distance=32)
    w = det_tensor((32, 4), 300)
    rpb.relative_attention_bias.weight.data.copy_(w)
    with torch.no_grad():
        save("smoke_relpos", weight=w, bias32=rpb(32, "cpu"), bias8=rpb(8, "cpu"))

    dim = 8
    net = Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7)
    spec = load_det(net, seed=300)
    x = det_tensor((2, 8, 7, 16, 16), 301)
    t = torch.tensor([11, 700])
    stages = ["init_conv", "init_temporal_attn", "downs.0.0", "downs.0.2", "downs.0.3", "downs.0.4", "mid_block1",
              "mid_spatial_attn", "mid_temporal_attn", "ups.0.4", "ups.2.3", "final_conv.0"]
    got, hooks = grab_stages(net, stages)
    with torch.no_grad():
        eps = net(x, t)
    for h in hooks:
        h.remove()
    save("smoke_unet", dim=dim, x=x, t=t, eps=eps, **spec_arrays(spec), **{"stage:" + k: v for k, v in got.items()})

    # guidance KAT from the reference pipeline methods (no model needed)
    ref_pipe = False
    try:
        # inference_2d star-imports the vendored PhiFlow evaluation solver (needs imageio/TF; evaluation only)
        stub_modules("dataset", "dataset.apps", "dataset.apps.evaluate_solver")
        import importlib
        inf = importlib.import_module("inference_2d")
        ref_pipe = True
    except Exception as e:  # noqa: BLE001
        print("!! 2d inference_2d import failed, guidance KAT falls back to text restatement:", repr(e))
    RES = torch.tensor([2, 19, 20, 17, 20, 1, 1.0]).reshape(1, 1, 7, 1, 1)
    if ref_pipe:
        args = types.SimpleNamespace(device="cpu", w_safe=0.9, safe_bound=0.1, standard_fixed_ratio=100.0,
                                     finetune_lr=1e-4)
        pipe = inf.InferencePipeline.__new__(inf.InferencePipeline)
        pipe.args_general, pipe.RESCALER, pipe.Q = args, RES, 0.01
        xs = det_tensor((4, 8, 7, 16, 16), 310, 0.3)
        xs[:2, -1, 6] += 0.2
        x_ = xs.clone().requires_grad_()
        grad = pipe.design_fn(x_)
        J = pipe.guidance(xs)
        wt = pipe.get_weight(xs)
        out = dict(x=xs, grad=grad, J=J, weight=wt, Q=0.01, w_safe=0.9, safe_bound=0.1, ratio=100.0)
        for k, v in {"plain": torch.tensor([0.5, 1.5, 0.25, 3.0, 0.0, 1.0]),
                     "inf": torch.tensor([0.5, float("inf"), 0.25, 3.0, float("inf"), 1.0]),
                     "zero": torch.zeros(6)}.items():
            out["nw_in_" + k] = v.clone()
            out["nw_out_" + k] = pipe.normalize_weights(v.clone())
        sc = det_tensor((40,), 320).abs()
        out["q_scores"] = sc
        for a in (0.04, 0.01, 0.5):
            out[f"q_{a}"] = pipe.get_quantile(sc, None, None, a)
        save("smoke_guidance", **out)

    T = 8
    net = Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7)
    load_det(net, seed=300)
    gd = GaussianDiffusion(net, image_size=16, frames=8, timesteps=T, loss_type="l2", standard_fixed_ratio=100.0)
    gd.eval()
    B = 2
    init = det_tensor((B, 16, 16), 330, 0.2).abs()
    noise = det_noise((B, 8, 7, 16, 16), 3000)
    if ref_pipe:
        args = types.SimpleNamespace(device="cpu", w_safe=0.9, safe_bound=-5.0, standard_fixed_ratio=100.0)
        pipe = inf.InferencePipeline.__new__(inf.InferencePipeline)
        pipe.args_general, pipe.RESCALER, pipe.Q = args, RES, 0.01
        design_fn = pipe.design_fn
    else:
        from oracle.samplers import smoke_guidance
        design_fn = smoke_guidance(0.01, 0.9, -5.0)
    with injected_noise(noise) as st:
        out = gd.sample(batch_size=B, design_fn=design_fn, enable_grad=False, init=init)
    save("smoke_traj_guided", out=out, init=init, T=T, draws=st["i"], noise_seed=3000, dim=dim, weight_seed=300,
         Q=0.01, w_safe=0.9, safe_bound=-5.0, ratio=100.0)
    control = det_tensor((B, 8, 2, 16, 16), 331, 0.3)
    with injected_noise(noise) as st:
        out = gd.sample(batch_size=B, design_fn=None, init=init, control=control)
    save("smoke_traj_calib", out=out, init=init, control=control, T=T, draws=st["i"], noise_seed=3000, dim=dim,
         weight_seed=300)
    T2, S2 = 20, 5
    net = Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7)
    load_det(net, seed=300)
    gd2 = GaussianDiffusion(net, image_size=16, frames=8, timesteps=T2, sampling_timesteps=S2, ddim_sampling_eta=1.0,
                            loss_type="l2", standard_fixed_ratio=100.0)
    gd2.eval()
    noise2 = det_noise((B, 8, 7, 16, 16), 3500)
    with injected_noise(noise2) as st:
        out = gd2.sample(batch_size=B, design_fn=design_fn, enable_grad=False, init=init)
    save("smoke_ddim_guided", out=out, init=init, T=T2, S=S2, eta=1.0, draws=st["i"], noise_seed=3500, dim=dim,
         weight_seed=300, Q=0.01, w_safe=0.9, safe_bound=-5.0, ratio=100.0)
    with injected_noise(noise2) as st:
        out = gd2.sample(batch_size=B, design_fn=None, init=init, control=control)
    save("smoke_ddim_calib", out=out, init=init, control=control, T=T2, S=S2, eta=1.0, draws=st["i"], noise_seed=3500,
         dim=dim, weight_seed=300)


def gen_wide(which):
    """One production-width U-Net forward per tree through the REAL reference, so that the production kernels (Winograd conv
    tiles, fused attention blocks) meet reference output directly and not only through the oracle: Burgers dim 64 (C2
    width), tokamak dim 256 (C3 width), smoke dim 64 with 32 frames at 32x32 (C4 width, quarter of the C4 area).
    Inputs are det_tensor(seed) (regenerated by the test), only eps is stored."""
    if which == "burgers":
        sys.path.insert(0, os.path.join(REF, "1D"))
        from model.unet import Unet2D
        net = Unet2D(dim=64, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        spec = load_det(net, seed=110)
        x, t = det_tensor((2, 3, 16, 128), 111), torch.tensor([9, 731])
        with torch.no_grad():
            eps = net(x, t)
        save("burgers_unet_wide", dim=64, x_seed=111, t=t, eps=eps, weight_seed=110, **spec_arrays(spec))
    elif which == "tokamak":
        sys.path.insert(0, os.path.join(REF, "tokamak"))
        from model.unet import Unet1D
        net = Unet1D(dim=256, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        spec = load_det(net, seed=210)
        x, t = det_tensor((2, 12, 128), 211), torch.tensor([2, 640])
        with torch.no_grad():
            eps = net(x, t)
        save("tokamak_unet_wide", dim=256, x_seed=211, t=t, eps=eps, weight_seed=210, **spec_arrays(spec))
    else:
        sys.path.insert(0, os.path.join(HERE, "_shims"))
        sys.path.insert(0, os.path.join(REF, "2d"))
        from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
        net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7)
        spec = load_det(net, seed=310)
        x, t = det_tensor((1, 32, 7, 32, 32), 311), torch.tensor([413])
        with torch.no_grad():
            eps = net(x, t)
        save("smoke_unet_wide", dim=64, x_seed=311, t=t, eps=eps, weight_seed=310, **spec_arrays(spec))


def gen_turbo(which):
    """The widths the reference ships configurations for besides the BASELINE ones: 1D `Unet2D(dim=128)` --
    "turbo", the only shipped-checkpoint config (1D/configs/inference_config.py:125-134) -- and tokamak `Unet1D(dim=128)`
    (turbo, tokamak/configs/inference_config.py:118-141) / `Unet1D(dim=64)` (the default, :76).  One eps per width through
    the REAL reference; inputs are det_tensor(seed) (regenerated by the tests), only eps is stored."""
    if which == "burgers":
        sys.path.insert(0, os.path.join(REF, "1D"))
        from model.unet import Unet2D
        net = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        spec = load_det(net, seed=120)
        x, t = det_tensor((2, 3, 16, 128), 121), torch.tensor([17, 903])
        with torch.no_grad():
            eps = net(x, t)
        save("burgers_unet_turbo", dim=128, x_seed=121, t=t, eps=eps, weight_seed=120, **spec_arrays(spec))
    else:
        sys.path.insert(0, os.path.join(REF, "tokamak"))
        from model.unet import Unet1D
        for name, dim, ws in (("tokamak_unet_turbo", 128, 220), ("tokamak_unet_small", 64, 230)):
            net = Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
            spec = load_det(net, seed=ws)
            x, t = det_tensor((2, 12, 128), ws + 1), torch.tensor([5, 512])
            with torch.no_grad():
                eps = net(x, t)
            save(name, dim=dim, x_seed=ws + 1, t=t, eps=eps, weight_seed=ws, **spec_arrays(spec))


def gen_long(which):
    """One FULL-SCHEDULE (T = 1000) guided DDPM trajectory per tree through the REAL reference at dim 8: the real 1000-entry
    coefficient tables are walked end to end (posterior_log_variance clamp at t = 0, sqrt_recipm1 at t = 999), the conditioning
    writes and the guidance act 1000 times.  Noise draw i is det_noise(seed)(i); only the final state is stored."""
    T, B, dim = 1000, 2, 8
    stub_modules("h5py", "tensorboardX", "ema_pytorch", "IPython")
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["ema_pytorch"].EMA = object
    sys.modules["IPython"].embed = None
    if which == "burgers":
        sys.path.insert(0, os.path.join(REF, "1D"))
        from model.unet import Unet2D
        from model.diffusion import GaussianDiffusion
        from utils.guidance import get_finetune_guidance
        net = Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        load_det(net, seed=100)
        gd = GaussianDiffusion(net, seq_length=(16, 128), timesteps=T, temporal=True, use_conv2d=True, is_condition_u0=True,
                               is_condition_uT=True, condition_idx=10, train_on_padded_locations=False)
        u0, uT = det_tensor((B, 128), 130, 0.1, -0.1, 0.3), det_tensor((B, 128), 131, 0.1, -0.1, 0.3)
        cfg = types.SimpleNamespace(use_max_safety=True, u_bound=0.05, guidance_weights={"w_score": 500.0}, InfFT_Q=None)
        Q = 0.01
        with injected_noise(det_noise((B, 3, 16, 128), 4000)) as st:
            out = gd.sample(batch_size=B, clip_denoised=True, u_init=u0, u_final=uT, guidance_u0=True,
                            nablaJ=lambda x: get_finetune_guidance(cfg, x, Q), J_scheduler=lambda t: 1.0, w_scheduler=None,
                            enable_grad=False)
        save("burgers_traj_long", out=out, u0=u0, uT=uT, Q=Q, w_score=500.0, u_bound=0.05, T=T, draws=st["i"], noise_seed=4000,
             dim=dim, weight_seed=100)
    elif which == "tokamak":
        sys.path.insert(0, os.path.join(REF, "tokamak"))
        stub_modules("kstar_solver")
        sys.modules["kstar_solver"].KSTARSolver = object
        from model.unet import Unet1D
        from model.diffusion import GaussianDiffusion
        from utils.guidance import GradientGuidance
        nt = 122
        net = Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        load_det(net, seed=200)
        gd = GaussianDiffusion(net, seq_length=128, nt=nt, timesteps=T, use_conv2d=False, temporal=False, guidance_u0=True,
                               is_condition_u0=True, is_condition_uT=True)
        u0, uT = det_tensor((B, 3), 220, 0.1) + 0.6, det_tensor((B, 2, nt), 221, 0.1) + 0.6
        target = det_tensor((B, 3, nt), 212, 0.3) + 1.0
        g = GradientGuidance.__new__(GradientGuidance)
        # MSE term and hinge both on (the safety-only production setting has a one-hot gradient: a weaker check)
        g.w_obj, g.w_safe, g.guidance_scaler, g.Q, g.safety_threshold, g.nt = 0.7, 0.3, 0.5, 0.1, 3.6, nt
        g.state_target = target
        with injected_noise(det_noise((B, 12, 128), 4100)) as st:
            out = gd.sample(batch_size=B, clip_denoised=True, guidance_u0=True, u_init=u0, u_final=uT, nablaJ=g,
                            J_scheduler=lambda t: 1.0, w_scheduler=None, enable_grad=False)
        save("tokamak_traj_long", out=out, u0=u0, uT=uT, target=target, T=T, draws=st["i"], noise_seed=4100, dim=dim,
             weight_seed=200, w_obj=0.7, w_safe=0.3, scaler=0.5, thr=3.6, Q=0.1)
    else:
        sys.path.insert(0, os.path.join(HERE, "_shims"))
        sys.path.insert(0, os.path.join(REF, "2d"))
        from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
        from ddpm.diffusion_2d import GaussianDiffusion
        import importlib
        stub_modules("dataset", "dataset.apps", "dataset.apps.evaluate_solver")
        inf = importlib.import_module("inference_2d")
        net = Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7)
        load_det(net, seed=300)
        gd = GaussianDiffusion(net, image_size=16, frames=8, timesteps=T, loss_type="l2", standard_fixed_ratio=100.0)
        gd.eval()
        init = det_tensor((B, 16, 16), 330, 0.2).abs()
        RES = torch.tensor([2, 19, 20, 17, 20, 1, 1.0]).reshape(1, 1, 7, 1, 1)
        args = types.SimpleNamespace(device="cpu", w_safe=0.9, safe_bound=-5.0, standard_fixed_ratio=100.0)
        pipe = inf.InferencePipeline.__new__(inf.InferencePipeline)
        pipe.args_general, pipe.RESCALER, pipe.Q = args, RES, 0.01
        with injected_noise(det_noise((B, 8, 7, 16, 16), 4200)) as st:
            out = gd.sample(batch_size=B, design_fn=pipe.design_fn, enable_grad=False, init=init)
        save("smoke_traj_long", out=out, init=init, T=T, draws=st["i"], noise_seed=4200, dim=dim, weight_seed=300, Q=0.01,
             w_safe=0.9, safe_bound=-5.0, ratio=100.0)


def _grad_summary(module, seed):
    """per-parameter gradient digests: L2 norm and the dot product with det_tensor(shape, seed + index) for EVERY key, the
    full gradient for the small tensors (biases, gains, time MLP rows) and a handful of conv weights"""
    keys, norms, dots, full = [], [], [], {}
    for i, (k, p) in enumerate(module.named_parameters()):
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        keys.append(k)
        norms.append(g.double().norm().item())
        dots.append((g.double() * det_tensor(tuple(g.shape), seed + i).double()).sum().item())
        if g.numel() <= 4096 or k.endswith("block1.proj.weight") and ("downs.0.0" in k or "mid_block1" in k or "ups.0.0" in k):
            full["grad:" + k] = g.clone()
    return dict(grad_keys=np.array(keys), grad_norms=np.array(norms), grad_dots=np.array(dots), **full)


def gen_grad(which):
    """Fine-tuning loss and gradients from the REAL reference: loss_b = p_losses(x_start, t, noise, mean=False),
    total = mean(weight_b * loss_b) as in the reference's finetune step (1D/inference/inference_ft.py:183-187,
    2d/inference_2d.py:267-279); stored: loss_b, total, gradient digests of every parameter (see _grad_summary)."""
    dim, B = 8, 3
    stub_modules("h5py", "tensorboardX", "ema_pytorch", "IPython")
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["ema_pytorch"].EMA = object
    sys.modules["IPython"].embed = None
    wt = det_tensor((B,), 5003).abs() + 0.5
    if which == "burgers":
        sys.path.insert(0, os.path.join(REF, "1D"))
        from model.unet import Unet2D
        from model.diffusion import GaussianDiffusion
        net = Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        load_det(net, seed=100)
        gd = GaussianDiffusion(net, seq_length=(16, 128), timesteps=1000, temporal=True, use_conv2d=True, is_condition_u0=True,
                               is_condition_uT=True, condition_idx=10, train_on_padded_locations=False)
        x0, noise, t = det_tensor((B, 3, 16, 128), 5000, 0.3), det_tensor((B, 3, 16, 128), 5001), torch.tensor([17, 480, 933])
        seed = 5100
    elif which == "tokamak":
        sys.path.insert(0, os.path.join(REF, "tokamak"))
        from model.unet import Unet1D
        from model.diffusion import GaussianDiffusion
        net = Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        load_det(net, seed=200)
        gd = GaussianDiffusion(net, seq_length=128, nt=122, timesteps=1000, use_conv2d=False, temporal=False, guidance_u0=True,
                               is_condition_u0=True, is_condition_uT=True)
        x0, noise, t = det_tensor((B, 12, 128), 5010, 0.3), det_tensor((B, 12, 128), 5011), torch.tensor([3, 611, 998])
        seed = 5200
    else:
        sys.path.insert(0, os.path.join(HERE, "_shims"))
        sys.path.insert(0, os.path.join(REF, "2d"))
        from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
        from ddpm.diffusion_2d import GaussianDiffusion
        net = Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7)
        load_det(net, seed=300)
        gd = GaussianDiffusion(net, image_size=16, frames=8, timesteps=1000, loss_type="l2", standard_fixed_ratio=100.0)
        x0, noise, t = det_tensor((B, 8, 7, 16, 16), 5020, 0.3), det_tensor((B, 8, 7, 16, 16), 5021), torch.tensor([40, 500, 960])
        seed = 5300
    net.train()
    loss_b = gd.p_losses(x0.clone(), t, noise=noise.clone(), mean=False)
    total = (wt * loss_b).mean()
    total.backward()
    save(f"{which}_grad", loss_b=loss_b.detach(), total=total.detach(), weight=wt, t=t, x0_seed=seed - 100 + (0 if which == "burgers" else 0),
         dim=dim, dot_seed=seed, **_grad_summary(net, seed))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "all":
        for w in ("burgers", "tokamak", "smoke", "burgers_wide", "tokamak_wide", "smoke_wide", "burgers_turbo", "tokamak_turbo", "burgers_long", "tokamak_long",
                  "smoke_long", "burgers_grad", "tokamak_grad", "smoke_grad"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
    elif which.endswith("_wide"):
        gen_wide(which[:-5])
    elif which.endswith("_turbo"):
        gen_turbo(which[:-6])
    elif which.endswith("_long"):
        gen_long(which[:-5])
    elif which.endswith("_grad"):
        gen_grad(which[:-5])
    else:
        {"burgers": gen_burgers, "tokamak": gen_tokamak, "smoke": gen_smoke}[which]()
