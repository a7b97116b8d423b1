"""-m gpu: the fine-tuning path against fixtures produced by the REAL reference
(oracle/make_goldens.py gen_grad): loss_b = p_losses(x_start, t, noise, mean=False), total = mean(weight_b * loss_b),
total.backward() -- loss values and the gradient of EVERY parameter (L2 norm + a fixed random projection per key, full
tensors for the small ones and three conv weights).  The drop-in nets run libsdc_hip.so kernels in both directions
(safediffcon_amd/autograd.py)."""
import numpy as np
import pytest
import torch

import safediffcon_amd as sdc
from oracle.detweights import det_params, det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(tree, spec):
    if tree == "burgers":
        net = sdc.Unet2D(dim=8, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        net.load_state_dict(det_params(spec, 100))
        gd = sdc.GaussianDiffusionBurgers(net.to(DEV), seq_length=(16, 128), timesteps=1000, temporal=True, use_conv2d=True,
                                          is_condition_u0=True, is_condition_uT=True, condition_idx=10,
                                          train_on_padded_locations=False).to(DEV)
        x0, noise = det_tensor((3, 3, 16, 128), 5000, 0.3), det_tensor((3, 3, 16, 128), 5001)
    elif tree == "tokamak":
        net = sdc.Unet1D(dim=8, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        net.load_state_dict(det_params(spec, 200))
        gd = sdc.GaussianDiffusionTokamak(net.to(DEV), seq_length=128, nt=122, timesteps=1000, guidance_u0=True).to(DEV)
        x0, noise = det_tensor((3, 12, 128), 5010, 0.3), det_tensor((3, 12, 128), 5011)
    else:
        net = sdc.Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2, 4), channels=7)
        net.load_state_dict(det_params(spec, 300))
        gd = sdc.GaussianDiffusionSmoke(net.to(DEV), image_size=16, frames=8, timesteps=1000, loss_type="l2",
                                        standard_fixed_ratio=100.0).to(DEV)
        x0, noise = det_tensor((3, 8, 7, 16, 16), 5020, 0.3), det_tensor((3, 8, 7, 16, 16), 5021)
    return net, gd, x0, noise


@pytest.mark.parametrize("tree", ["burgers", "tokamak", "smoke"])
def test_finetune_loss_and_gradients_vs_reference(golden, tree):
    g = golden(tree + "_grad")
    net, gd, x0, noise = _build(tree, golden(tree + "_unet").spec())
    net.train()
    loss_b = gd.p_losses(x0.to(DEV), g["t"].to(DEV), noise=noise.to(DEV), mean=False)
    total = (g["weight"].to(DEV) * loss_b).mean()
    total.backward()
    el = ((loss_b.detach().cpu() - g["loss_b"]).abs() / g["loss_b"].abs()).max().item()
    assert el < 2e-5 and abs(total.item() - g.scalar("total")) < 2e-5 * abs(g.scalar("total")), (el, total.item(), g.scalar("total"))
    keys = [str(k) for k in g["grad_keys"]]
    params = dict(net.named_parameters())
    seed = int(g.scalar("dot_seed"))
    worst_n = worst_d = worst_f = 0.0
    noise_keys = 0
    gmax = float(np.max(g.z["grad_norms"]))
    for i, k in enumerate(keys):
        gr = params[k].grad
        n_ref, d_ref = float(g.z["grad_norms"][i]), float(g.z["grad_dots"][i])
        if not params[k].requires_grad:                   # rotary freqs: not learnable, in the reference neither
            assert n_ref == 0.0 and k.endswith("rotary_emb.freqs"), k
            continue
        assert gr is not None, f"no gradient for {k}"
        gr = gr.detach().double().cpu()
        # gradients that are zero in exact arithmetic (a conv bias in front of a GroupNorm with one channel per group) are
        # rounding noise on both sides: they only have to be as small here as they are in the reference
        if n_ref < 1e-4 * gmax:
            assert gr.norm().item() < 2e-4 * gmax, (k, gr.norm().item(), n_ref, gmax)
            noise_keys += 1
            continue
        scale = n_ref
        en = abs(gr.norm().item() - n_ref) / scale
        proj = det_tensor(tuple(gr.shape), seed + i).double()
        ed = abs((gr * proj).sum().item() - d_ref) / (scale * proj.norm().item())
        worst_n, worst_d = max(worst_n, en), max(worst_d, ed)
        assert en < 5e-5 and ed < 5e-5, (k, en, ed, n_ref)
        if "grad:" + k in g.keys():
            full = g["grad:" + k].double()
            ef = ((gr - full).abs().max() / full.abs().max()).item()
            worst_f = max(worst_f, ef)
            assert ef < 1e-4, (k, ef)
    # every parameter of the reference model received a gradient digest, and vice versa
    assert set(keys) == set(params) or set(keys) == {k for k in params if not k.endswith("rotary_emb.freqs")} | {k for k in keys if k.endswith("rotary_emb.freqs")}
    print(f"[measured] {tree} fine-tune step vs the reference: loss rel err {el:.2e}; gradients of {len(keys)} parameters: "
          f"worst norm err {worst_n:.2e}, worst projection err {worst_d:.2e}, worst element err (stored tensors) {worst_f:.2e}; "
          f"{noise_keys} gradients are zero in exact arithmetic (checked for smallness only)")


def test_finetune_step_updates_the_sampler():
    """loss.backward(); optimizer.step() -> the graph-replayed sampler sees the new weights WITHOUT a refresh() call: the
    reference's loops (1D/inference/inference_ft.py:183-187, tokamak/inference/pipeline.py:238-263, 2d/inference_2d.py:267-279)
    never make one; a plan re-packs itself when the parameters' version counters have moved"""
    torch.manual_seed(0)
    net = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1).to(DEV)
    gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=1000, temporal=True, use_conv2d=True,
                                      is_condition_u0=True, is_condition_uT=True).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x, t = det_tensor((2, 3, 16, 128), 1).to(DEV), torch.tensor([5, 700], device=DEV)
    e0 = net(x, t)
    state = det_tensor((4, 3, 16, 128), 2, 0.3).to(DEV)
    losses = []
    for _ in range(3):
        torch.manual_seed(1)
        loss = (torch.ones(4, device=DEV) * gd(state, mean=False)).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    e1 = net(x, t)
    assert losses[-1] < losses[0] and not torch.equal(e0, e1) and torch.isfinite(e1).all()
    # the differentiable forward and the sampler forward are the same function of the weights
    with torch.no_grad():
        e2 = net.forward_train(x, t)
    assert (e1 - e2).abs().max().item() < 1e-4 * e1.abs().max().item()
    # a fresh net loaded with the updated weights is the same function
    net2 = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1).to(DEV)
    net2.load_state_dict(net.state_dict())
    assert torch.equal(net2(x, t), e1)
    # ... and so is a sample() after the step: same injected noise, stale-plan net vs fresh net
    gd2 = sdc.GaussianDiffusionBurgers(net2, seq_length=(16, 128), timesteps=1000, sampling_timesteps=4, ddim_sampling_eta=1.0,
                                       temporal=True, use_conv2d=True, is_condition_u0=True, is_condition_uT=True).to(DEV)
    gd1 = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=1000, sampling_timesteps=4, ddim_sampling_eta=1.0,
                                       temporal=True, use_conv2d=True, is_condition_u0=True, is_condition_uT=True).to(DEV)
    u0, uT = det_tensor((2, 128), 3, 0.1).to(DEV), det_tensor((2, 128), 4, 0.1).to(DEV)
    noise = lambda i: det_tensor((2, 3, 16, 128), 100 + i)
    s_before = gd1.sample(batch_size=2, u_init=u0, u_final=uT, guidance_u0=False, enable_grad=False, noise=noise)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.01)
    net2.load_state_dict(net.state_dict())
    s1 = gd1.sample(batch_size=2, u_init=u0, u_final=uT, guidance_u0=False, enable_grad=False, noise=noise)
    s2 = gd2.sample(batch_size=2, u_init=u0, u_final=uT, guidance_u0=False, enable_grad=False, noise=noise)
    assert torch.equal(s1, s2) and not torch.equal(s1, s_before)


@pytest.mark.parametrize("tree", ["smoke", "burgers", "tokamak"])
def test_finetune_gradients_at_production_width(tree):
    """the production kernels (Winograd convs in both directions, the fused attention blocks, wgrad over long rows) in the
    fine-tuning step: smoke dim 64 at 32 frames of 32x32, burgers dim 64, tokamak dim 256 -- against the oracle's functional
    net under PyTorch-ROCm autograd on the same device (the oracle's forward is pinned by the *_unet_wide reference fixtures)"""
    from oracle import nets as onets
    if tree == "smoke":
        net = sdc.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7)
        shape, fwd, kw, B = (32, 7, 32, 32), onets.unet_smoke, dict(dim=64, dim_mults=(1, 2, 4)), 2
    elif tree == "burgers":
        net = sdc.Unet2D(dim=64, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        shape, fwd, kw, B = (3, 16, 128), onets.unet_burgers, dict(dim=64), 4
    else:
        net = sdc.Unet1D(dim=256, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        shape, fwd, kw, B = (12, 128), onets.unet_tokamak, dict(dim=256), 4
    spec = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(det_params(spec, 77))
    net.to(DEV).train()
    x, t = det_tensor((B, *shape), 78).to(DEV), torch.tensor([3, 400, 777, 999][:B], device=DEV)
    tgt, w = det_tensor((B, *shape), 79).to(DEV), (det_tensor((B,), 80).abs() + 0.5).to(DEV)
    loss = (w * ((net.forward_train(x, t) - tgt) ** 2).flatten(1).mean(1)).mean()
    loss.backward()
    P = {k: v.detach().clone().requires_grad_() for k, v in net.state_dict().items()}
    ref = (w * ((fwd(P, x, t, **kw) - tgt) ** 2).flatten(1).mean(1)).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst, checked = 0.0, 0
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        gr, gref = p.grad, P[k].grad
        assert gr is not None and gref is not None, k
        n_ref = gref.norm().item()
        if n_ref < 1e-4 * gmax:
            assert gr.norm().item() < 2e-4 * gmax, k
            continue
        err = ((gr - gref).norm() / n_ref).item()
        worst, checked = max(worst, err), checked + 1
        assert err < 2e-4, (k, err)
    print(f"[measured] {tree} production-width fine-tune step vs PyTorch-ROCm autograd of the oracle: loss {loss.item():.6f} vs {ref.item():.6f}; "
          f"{checked} parameter gradients, worst relative L2 error {worst:.2e}")


def test_differentiable_ddim_tail(golden):
    """sample(enable_grad=True) with DDIM: the last step is an autograd graph over the model parameters (1D/model/diffusion.py:
    524-551, 2d/ddpm/diffusion_2d.py:379-399) -- the numbers equal the gradient-free sampler's, a loss on the sample
    back-propagates into every parameter the last U-Net evaluation touches, and the gradient matches PyTorch-ROCm autograd of
    the oracle's functional net on the same last step."""
    from oracle import nets as onets
    g = golden("burgers_ddim_guided")
    T, S, eta = int(g.scalar("T")), int(g.scalar("S")), g.scalar("eta")
    spec = golden("burgers_unet").spec()
    net = sdc.Unet2D(dim=8, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    net.load_state_dict(det_params(spec, 100))
    net.to(DEV)
    gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta,
                                      temporal=True, use_conv2d=True, is_condition_u0=True, is_condition_uT=True,
                                      condition_idx=10, train_on_padded_locations=False).to(DEV)
    from oracle.detweights import det_noise
    noise = det_noise((2, 3, 16, 128), int(g.scalar("noise_seed")))
    guid = sdc.BurgersGuidance(g.scalar("Q"), g.scalar("w_score"), g.scalar("u_bound"))
    kw = dict(batch_size=2, clip_denoised=True, u_init=g["u0"], u_final=g["uT"], guidance_u0=True, nablaJ=guid,
              J_scheduler=lambda t: 1.0, noise=noise)
    plain = gd.sample(enable_grad=False, **kw)
    out = gd.sample(enable_grad=True, **kw)
    assert out.requires_grad and not plain.requires_grad
    err = (out.detach() - plain).abs().max().item()
    assert err < 2e-4 and (out.detach().cpu() - g["out"]).abs().max().item() < 1.5e-3
    # the reference's backward fine-tune loss (1D/inference/inference_ft.py:192-201)
    s = out[:, 2, :11, :].amax(dim=(-1, -2))
    loss = (torch.clamp(s + 0.01 - 0.05 ** 2, min=0) ** 2).mean() + 1e-3 * (out ** 2).mean()
    loss.backward()
    got = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    assert len(got) == len([1 for p in net.parameters() if p.requires_grad])
    # reference arithmetic of the same last step on the oracle's functional net (PyTorch-ROCm autograd)
    P = {k: v.detach().clone().requires_grad_() for k, v in net.state_dict().items()}
    hooks = {}
    orig = net.forward_train

    def spy(x, t):
        hooks["x"], hooks["t"] = x.detach().clone(), t.clone()
        return orig(x, t)
    net.forward_train = spy
    net.zero_grad()
    out2 = gd.sample(enable_grad=True, **kw)
    net.forward_train = orig
    x_last, t_last = hooks["x"], hooks["t"]
    tabs = gd
    a, b = tabs.sqrt_recip_alphas_cumprod[t_last[0]], tabs.sqrt_recipm1_alphas_cumprod[t_last[0]]
    eps = onets.unet_burgers(P, x_last, t_last, dim=8)
    x0 = (a * x_last - b * eps).clamp(-1, 1)
    gg = guid(x0.detach().clone().requires_grad_()).detach()
    ref = (a * x_last - b * (eps + gg)).clamp(-1, 1)
    assert (ref.detach() - out2.detach()).abs().max().item() < 2e-4
    s = ref[:, 2, :11, :].amax(dim=(-1, -2))
    lref = (torch.clamp(s + 0.01 - 0.05 ** 2, min=0) ** 2).mean() + 1e-3 * (ref ** 2).mean()
    lref.backward()
    gmax = max(v.grad.norm().item() for v in P.values())
    worst = 0.0
    for k, gr in got.items():
        n_ref = P[k].grad.norm().item()
        if n_ref < 1e-4 * gmax:
            continue
        worst = max(worst, ((gr - P[k].grad).norm() / n_ref).item())
    print(f"[measured] differentiable DDIM tail: sample vs gradient-free sampler {err:.2e}; loss {loss.item():.3e} vs {lref.item():.3e}; "
          f"worst relative gradient error vs PyTorch-ROCm autograd of the oracle {worst:.2e}")
    assert abs(loss.item() - lref.item()) < 1e-4 * abs(lref.item()) + 1e-9 and worst < 5e-4


def test_batched_weight_packing_equals_the_single_launch_form():
    """sdc_pack_batch_run packs many conv weights in one launch; bit for bit what sdc_pack_conv_weight writes for each"""
    from safediffcon_amd import grad_ops
    from safediffcon_amd.grad_ops import PackArena
    torch.manual_seed(3)
    shapes = [((64, 7, 1, 1, 3), 2), ((64, 7, 1, 1, 3), 5), ((24, 40, 1, 3, 3), 3), ((16, 24, 3, 3, 3), 4), ((130, 66, 1, 1, 1), 0),
              ((8, 12, 1, 7, 7), 0), ((256, 512, 1, 1, 3), 2), ((33, 17, 1, 3, 3), 4)]
    ws = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s, _ in shapes]
    arena = PackArena()
    arena.begin(torch.device(DEV))
    for w, (_, prec) in zip(ws, shapes):                       # first step: nothing cached, every request recorded
        assert PackArena.cacheable(w)
        for flip in (False, True):
            assert arena.get(w, prec, flip) is None
    arena.begin(torch.device(DEV))                             # second step: one launch
    assert arena.launch is not None and arena.launch[0] == 2 * len(ws)
    for w, (_, prec) in zip(ws, shapes):
        for flip in (False, True):
            got = arena.get(w, prec, flip)
            ref = grad_ops.pack_conv_weight(w, prec, flip=flip)
            assert got is not None and got.shape == ref.shape and torch.equal(got, ref), (tuple(w.shape), prec, flip)
    with torch.no_grad():                                      # the optimiser moves the weights in place: the next step re-packs
        for w in ws:
            w.mul_(1.5)
    arena.begin(torch.device(DEV))
    for w, (_, prec) in zip(ws, shapes):
        assert torch.equal(arena.get(w, prec, True), grad_ops.pack_conv_weight(w, prec, flip=True))
    # a weight computed inside the graph has no stable address: never cached
    assert not PackArena.cacheable(ws[0] * 2.0)
    # entries nobody asks for are dropped after two steps
    arena.begin(torch.device(DEV)); arena.begin(torch.device(DEV)); arena.begin(torch.device(DEV))
    assert arena.launch is None and not arena.meta


@pytest.mark.parametrize("tree", ["burgers", "tokamak", "smoke"])
def test_finetune_steps_through_the_pack_arena(golden, tree):
    """step 1 packs conv by conv and records; step 2 onwards reads every parameter conv weight from the arena's one launch:
    same loss, same gradients to the bit; after an in-place parameter update the arena follows (a fresh net agrees)"""
    spec = golden(tree + "_unet").spec()
    net, gd, x0, noise = _build(tree, spec)
    x0, noise = x0.to(DEV), noise.to(DEV)
    t = torch.tensor([5, 400, 900], device=DEV)

    def step(n, g):
        n.zero_grad(set_to_none=True)
        loss = g.p_losses(x0, t, noise=noise, mean=False).mean()
        loss.backward()
        return loss.item(), {k: p.grad.clone() for k, p in n.named_parameters() if p.grad is not None}

    l1, g1 = step(net, gd)
    arena = net._trainer().arena
    assert arena.launch is None and arena.pending
    l2, g2 = step(net, gd)
    assert arena.launch is not None and arena.used and not arena.pending
    assert l1 == l2 and all(torch.equal(g1[k], g2[k]) for k in g1)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.02)
    l3, g3 = step(net, gd)
    assert l3 != l2
    net2, gd2, _, _ = _build(tree, spec)
    net2.load_state_dict(net.state_dict())
    l4, g4 = step(net2, gd2)                                  # a fresh net: packs conv by conv
    assert net2._trainer().arena.launch is None
    assert l3 == l4 and all(torch.equal(g3[k], g4[k]) for k in g3)
