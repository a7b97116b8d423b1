"""-m gpu, round 5:
  * the widths the reference SHIPS besides the BASELINE ones -- Unet2D(dim=128) "turbo" (1D/configs/inference_config.py:125-134,
    the only shipped-checkpoint config), Unet1D(dim=128) turbo and Unet1D(dim=64) default (tokamak/configs/inference_config.py:
    118-141, :76): eps against fixtures from the REAL reference, a 3-step guided DDPM sampler and a 4-of-20 DDIM sampler against
    the oracle, C2-turbo at B = 256 through the size-independent properties;
  * the sampler -> score-check CHAIN on numbers: a sampled smoke batch through multi_evaluate against the oracle's rollout and
    the reference's metric formulas (2d/inference_2d.py:460-505);
  * weights written behind autograd's back (`p.data.lerp_`, `p.data = ...`: the reference's EMA updates) are seen by the next
    forward without a refresh() call.
Tolerances: fp32 kernels vs the fp32 reference differ by summation order only; gates ~2-5x the errors measured on MI355X
(printed by every test, `pytest -s`)."""
import numpy as np
import pytest
import torch

import safediffcon_amd as sdc
from oracle import nets as onets
from oracle import samplers as osam
from oracle import schedules as osched
from oracle.detweights import det_noise, det_params, det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _spec(net):
    return [(k, tuple(v.shape)) for k, v in net.state_dict().items()]


def _mse(a, b):
    return ((a - b) ** 2).mean().item()


def _report(tag, got, want):
    err = (got - want).abs().max().item()
    print(f"[measured] {tag}: max|err| {err:.3e}  eps-MSE {_mse(got, want):.3e}  (|ref|max {want.abs().max().item():.3f})")
    return err


def _net(tree, dim):
    if tree == "burgers":
        return sdc.Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    return sdc.Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)


WIDTHS = [("burgers", 128, "burgers_unet_turbo"), ("tokamak", 128, "tokamak_unet_turbo"), ("tokamak", 64, "tokamak_unet_small")]


# ------------------------------------------------------------------ eps at the shipped widths vs the REAL reference
@pytest.mark.parametrize("tree,dim,name", WIDTHS)
def test_shipped_width_eps_vs_reference_fixture(golden, tree, dim, name):
    g = golden(name)
    net = _net(tree, dim)
    net.load_state_dict(det_params(g.spec(), int(g.scalar("weight_seed"))))
    net.to(DEV)
    shape = (2, 3, 16, 128) if tree == "burgers" else (2, 12, 128)
    x = det_tensor(shape, int(g.scalar("x_seed")))
    eps = net(x.to(DEV), g["t"].to(DEV)).cpu()
    assert _report(f"{tree} dim {dim} vs reference", eps, g["eps"]) < 1e-4 and _mse(eps, g["eps"]) <= 1e-9
    used = sorted({fn.__name__ for fn, _ in net.entry(shape, 2)["plan"].calls})
    print(f"[kernels] {tree} dim {dim}: {used}")
    assert "sdc_conv_gn" in used or "sdc_conv" in used


# ------------------------------------------------------------------ guided DDPM + DDIM samplers at the shipped widths vs the oracle
@pytest.mark.parametrize("tree,dim,name", WIDTHS)
def test_shipped_width_samplers_vs_oracle(tree, dim, name):
    """1D/model/diffusion.py:368-449 / :451-555 and tokamak/model/diffusion.py:310-372 / :374-496 with the U-Net widths of the
    shipped configurations: 3 guided DDPM steps, then 4 of 20 DDIM steps (eta 1), injected noise, against the CPU oracle."""
    net = _net(tree, dim)
    P = det_params(_spec(net), 500 + dim)
    net.load_state_dict(P)
    net.to(DEV)
    B = 2
    if tree == "burgers":
        eps_fn = lambda a, b: onets.unet_burgers(P, a, b, dim=dim)
        u0, uT = det_tensor((B, 128), 61, 0.1, -0.1, 0.3), det_tensor((B, 128), 62, 0.1, -0.1, 0.3)
        kw = dict(u_init=u0, u_final=uT)
        mk = lambda **k: sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), temporal=True, use_conv2d=True, is_condition_u0=True,
                                                      is_condition_uT=True, condition_idx=10, train_on_padded_locations=False, **k).to(DEV)
        guid, oguid = sdc.BurgersGuidance(0.01, 500.0, 0.05), osam.burgers_guidance(0.01, 500.0, 0.05)
        osample, oddim, shape = osam.sample_burgers, osam.ddim_burgers, (B, 3, 16, 128)
    else:
        eps_fn = lambda a, b: onets.unet_tokamak(P, a, b, dim=dim)
        u0, uT = det_tensor((B, 3), 51, 0.1) + 0.6, det_tensor((B, 2, 122), 52, 0.1) + 0.6
        target = det_tensor((B, 3, 122), 53, 0.3) + 1.0
        kw = dict(u_init=u0, u_final=uT)
        mk = lambda **k: sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, **k).to(DEV)
        guid = sdc.TokamakGuidance(target, 122, w_obj=0.3, w_safe=1.0, guidance_scaler=0.5, Q=0.05, safety_threshold=4.98)
        oguid = osam.tokamak_guidance(target, 122, 0.05, 4.98, 0.3, 1.0, 0.5)
        osample, oddim, shape = osam.sample_tokamak, osam.ddim_tokamak, (B, 12, 128)
    # guided DDPM, 3 steps
    T = 3
    gd = mk(timesteps=T)
    noise = det_noise(shape, 9100 + dim)
    out = gd.sample(batch_size=B, nablaJ=guid, J_scheduler=lambda t: 1.0, enable_grad=False, noise=noise, **kw).cpu()
    ref = osample(eps_fn, osched.make_tables("cosine", T), B, noise, nablaJ=oguid, enable_grad=False, **kw)
    assert torch.isfinite(out).all()
    assert _report(f"{tree} dim {dim}: 3-step guided DDPM vs oracle", out, ref) < 1e-3 and _mse(out, ref) <= 1e-8
    # DDIM, 4 of 20 steps, eta 1
    T, S = 20, 4
    gd = mk(timesteps=T, sampling_timesteps=S, ddim_sampling_eta=1.0)
    noise = det_noise(shape, 9200 + dim)
    out = gd.sample(batch_size=B, nablaJ=guid, J_scheduler=lambda t: 1.0, enable_grad=False, noise=noise, **kw).cpu()
    ref = oddim(eps_fn, osched.make_tables("cosine", T), B, noise, S=S, eta=1.0, nablaJ=oguid, **kw)
    assert torch.isfinite(out).all()
    assert _report(f"{tree} dim {dim}: DDIM 4 of 20 (eta 1) vs oracle", out, ref) < 3e-3 and _mse(out, ref) <= 1e-8


# ------------------------------------------------------------------ C2 with the shipped ("turbo") net at the full batch
def test_c2_turbo_batch256_properties():
    """BASELINE configs[1] with Unet2D(dim=128): B = 256 -- determinism, batch independence at both ends of the batch, three
    samples against the full-width CPU oracle, two guided sampler steps (finite, clipped, a trajectory alone = in the batch)."""
    net = _net("burgers", 128)
    P = det_params(_spec(net), 628)
    net.load_state_dict(P)
    net.to(DEV)
    B = 256
    x = det_tensor((B, 3, 16, 128), 12)
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(13))
    eps = net(x.to(DEV), t.to(DEV))
    assert torch.isfinite(eps).all() and torch.equal(eps, net(x.to(DEV), t.to(DEV)))
    idx = [0, 137, 255]
    ref = onets.unet_burgers(P, x[idx], t[idx], dim=128)
    assert _report("C2-turbo B=256 forward, samples 0/137/255 vs oracle", eps[idx].cpu(), ref) < 2e-4 and _mse(eps[idx].cpu(), ref) <= 1e-9
    sub = net(x[idx].to(DEV), t[idx].to(DEV))
    assert _report("C2-turbo: in the batch of 256 vs in a batch of 3", eps[idx].cpu(), sub.cpu()) < 2e-5
    gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=2, temporal=True, use_conv2d=True, is_condition_u0=True,
                                      is_condition_uT=True, condition_idx=10, train_on_padded_locations=False).to(DEV)
    u0, uT = det_tensor((B, 128), 61, 0.1, -0.1, 0.3), det_tensor((B, 128), 62, 0.1, -0.1, 0.3)
    noise = det_noise((B, 3, 16, 128), 9300)
    out = gd.sample(batch_size=B, u_init=u0, u_final=uT, nablaJ=sdc.BurgersGuidance(0.01, 500.0, 0.05), enable_grad=False, noise=noise).cpu()
    # (the 1-D loop imposes its conditions BEFORE each p_sample and returns the last p_sample output as it is,
    # 1D/model/diffusion.py:432-449: the returned rows are not the conditions themselves)
    assert torch.isfinite(out).all() and out.abs().max() <= 1.0 + 1e-6
    one = gd.sample(batch_size=1, u_init=u0[200:201], u_final=uT[200:201], nablaJ=sdc.BurgersGuidance(0.01, 500.0, 0.05), enable_grad=False,
                    noise=lambda i: noise(i)[200:201]).cpu()
    assert _report("C2-turbo 2-step guided trajectory, sample 200: batch of 256 vs alone", out[200:201], one) < 1e-4


# ------------------------------------------------------------------ sampler -> score check, on numbers
def test_c4_sampled_batch_through_the_score_check_vs_oracle():
    """BASELINE configs[3] end to end on values (the chain was property-tested only): a short-schedule guided C4
    sample (B = 2) -> un-rescale -> control channel means like InferencePipeline.run_model (2d/inference_2d.py:197-237) ->
    multi_evaluate (:407-507); the rolled-out fields against oracle.smoke_solver.multi_evaluate_fields on the SAME sampled
    pred, and all eight metric arrays against the reference's formulas (:460-505) evaluated in numpy on the oracle's fields."""
    from oracle import smoke_solver as osolver
    from safediffcon_amd import smoke_solver as ss
    net = sdc.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7)
    net.load_state_dict(det_params(_spec(net), 31))
    net.to(DEV)
    B = 2
    gs = sdc.GaussianDiffusionSmoke(net, image_size=64, frames=32, timesteps=2, standard_fixed_ratio=100.0).to(DEV)
    state = torch.rand(B, 32, 7, 64, 64, generator=torch.Generator().manual_seed(3))
    state[:, 0, 0, 32:] = 0
    R = torch.tensor(sdc.diffusion.SMOKE_RESCALER, dtype=torch.float32).reshape(1, 1, 7, 1, 1)
    out = gs.sample(batch_size=B, design_fn=sdc.SmokeGuidance(0.01, 0.9, 0.1), init=(state[:, 0, 0] / R[0, 0, 0, 0, 0]).to(DEV),
                    noise=det_noise((B, 32, 7, 64, 64), 8000)) * R.to(DEV)
    pred = torch.zeros_like(out)
    pred[:, :, :-2] = out[:, :, :-2]
    pred[:, :, -2] = out[:, :, -2].mean((-2, -1), keepdim=True).expand(-1, -1, 64, 64)
    pred[:, :, -1] = out[:, :, -1].mean((-2, -1), keepdim=True).expand(-1, -1, 64, 64)
    pred_h, data_h = pred.cpu(), state
    sim = ss.init_sim_128()
    fields = ss.solver_out(sim, pred.clone(), state.to(DEV)).cpu().numpy()
    want = osolver.multi_evaluate_fields(pred_h.numpy(), data_h.numpy())                 # 2 x 255 steps x 500 CG iterations in numpy
    scale = lambda a: max(1e-30, float(np.max(np.abs(a))))
    for ch, tol in ((0, 1e-5), (1, 1e-8), (2, 1e-8), (3, 0.0), (4, 0.0), (5, 1e-10), (6, 1e-10)):
        err = float(np.max(np.abs(fields[:, :, ch] - want[:, :, ch]))) / scale(want[:, :, ch])
        print(f"[measured] sampled batch -> rollout, field channel {ch}: {err:.2e} of scale")
        assert err <= tol, (ch, err)
    Q, sb = 0.01, 0.1
    res = ss.multi_evaluate(pred.clone(), state.to(DEV), Q=Q, safe_bound=sb, sim=sim)
    p = pred_h.numpy().astype(np.float64).copy()
    p[:, 0, 0] = data_h[:, 0, 0].numpy()
    p[:, 0] = 0
    d = want.copy()
    d[:, 0] = 0
    diff = p - d
    exp = (-d[:, -1, 5, 0, 0], d[:, -1, 6, 0, 0], np.maximum(d[:, -1, 6, 0, 0] - sb, 0), np.maximum(p[:, -1, 6, 0, 0] + Q - sb, 0),
           np.maximum(d[:, :, 6, 0, 0] - sb, 0), np.maximum(p[:, :, 6, 0, 0] + Q - sb, 0),
           np.concatenate((diff[:, :, :3], diff[:, :, -2:]), axis=2).__pow__(2).mean((1, 2, 3, 4)),
           np.sqrt((diff[:, :, :3] ** 2).sum((1, 2, 3, 4))) / np.sqrt((d[:, :, :3] ** 2).sum((1, 2, 3, 4))))
    names = ("J_target", "safe_target", "J_safe", "J_safe_pred", "J_time", "J_pred_time", "mse", "normalized_l2")
    assert len(res) == 8
    for nm, r, e in zip(names, res, exp):
        assert isinstance(r, np.ndarray) and r.shape == e.shape, nm
        print(f"[measured] sampled batch -> metric {nm}: max|diff| {float(np.max(np.abs(r - e))):.2e} (|ref| {scale(e):.3e})")
        np.testing.assert_allclose(r, e, rtol=2e-5, atol=1e-9, err_msg=nm)


# ------------------------------------------------------------------ weights written through .data
def test_data_writes_and_reseats_are_seen_without_refresh():
    """ema_pytorch (1D/model/trainer.py, tokamak/model/trainer.py) updates the EMA copy with `p.data.lerp_` / `.copy_`, the 2-D
    tree's EMA re-seats `p.data = old * beta + (1 - beta) * new` (video_diffusion_pytorch_conv3d.py:121-124): neither moves an
    autograd version counter, and the allocator hands re-seated tensors the same two addresses in turn.  The plan's content
    stamp must see all of it; each forward is compared with a freshly built net holding the same weights."""
    torch.manual_seed(0)
    net = sdc.Unet1D(dim=8, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
    P0 = det_params(_spec(net), 77)
    net.load_state_dict(P0)
    net.to(DEV)
    x, t = det_tensor((2, 12, 128), 78).to(DEV), torch.tensor([3, 500], device=DEV)

    def fresh_eps():
        other = sdc.Unet1D(dim=8, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        other.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
        other.to(DEV)
        return other(x, t)
    e0 = net(x, t)
    assert torch.equal(e0, fresh_eps())
    w = net.P("init_conv.weight")
    v0 = w._version
    with torch.no_grad():
        w.data.lerp_(torch.zeros_like(w), 0.5)                      # in place through .data: no version bump
    assert w._version == v0
    e1 = net(x, t)
    assert not torch.equal(e1, e0) and torch.equal(e1, fresh_eps())
    # two consecutive re-seats of the same parameter (an even number: address and version can both come back)
    ptrs = [w.data_ptr()]
    for k in range(2):
        w.data = w.data * 0.9 + 0.1 * torch.ones_like(w.data) * (k + 1)
        ptrs.append(w.data_ptr())
        ek = net(x, t)
        assert torch.equal(ek, fresh_eps()), f"re-seat {k}"
    print(f"[measured] re-seated addresses: {[hex(p) for p in ptrs]}, versions stayed {w._version}")
    # ... and the sampler's LUT / graph follow too
    gd = sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=3).to(DEV)
    u0, uT = det_tensor((2, 3), 51, 0.1) + 0.6, det_tensor((2, 2, 122), 52, 0.1) + 0.6
    noise = det_noise((2, 12, 128), 9400)
    a = gd.sample(batch_size=2, u_init=u0, u_final=uT, nablaJ=None, enable_grad=False, noise=noise)
    with torch.no_grad():
        for p in net.parameters():
            p.data.mul_(0.97)
    b = gd.sample(batch_size=2, u_init=u0, u_final=uT, nablaJ=None, enable_grad=False, noise=noise)
    net.refresh()
    c = gd.sample(batch_size=2, u_init=u0, u_final=uT, nablaJ=None, enable_grad=False, noise=noise)
    assert not torch.equal(a, b) and torch.equal(b, c)


def test_chan_norm_refuses_misaligned_vector_rows():
    """the 16-byte form of sdc_chan_norm must not be swapped for the scalar one on a misaligned pointer (another
    summation order for C > 128): the call fails loudly instead"""
    from safediffcon_amd import _lib
    lib = _lib.get_lib()
    B, C_, S = 1, 256, 2048
    buf = torch.zeros(B * C_ * S + 4, device=DEV)
    y = torch.empty(B * C_ * S, device=DEV)
    g = torch.ones(C_, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.sdc_chan_norm(buf.data_ptr(), g.data_ptr(), None, y.data_ptr(), B, C_, S, 0, 1e-5, st) == 0
    rc = lib.sdc_chan_norm(buf.data_ptr() + 4, g.data_ptr(), None, y.data_ptr(), B, C_, S, 0, 1e-5, st)
    assert rc == -2 and "16-byte aligned" in _lib.last_error()
    torch.cuda.synchronize()


# ------------------------------------------------------------------ the fine-tuning step under one hipGraph
@pytest.mark.parametrize("tree", ["tokamak", "burgers", "smoke"])
def test_graphed_finetune_step_equals_eager_and_sees_optimizer_steps(tree):
    """sdc.GraphedLossStep: loss = mean(w * p_losses(state, t, noise)); loss.backward() captured once and replayed -- loss and
    every parameter gradient equal the eager step's bit for bit (the same kernels on the same buffers), and an eager
    optimizer.step() between two replays is seen by the next replay (the weights are packed inside the captured step):
    tokamak/inference/pipeline.py:238-263, 1D/inference/inference_ft.py:183-226."""
    torch.manual_seed(0)
    if tree == "tokamak":
        net = sdc.Unet1D(dim=32, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        shape = (12, 128)
        mk = lambda n: sdc.GaussianDiffusionTokamak(n, seq_length=128, nt=122, timesteps=50).to(DEV)
    elif tree == "burgers":
        net = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        shape = (3, 16, 128)
        mk = lambda n: sdc.GaussianDiffusionBurgers(n, seq_length=(16, 128), timesteps=50, temporal=True, use_conv2d=True,
                                                    is_condition_u0=True, is_condition_uT=True, condition_idx=10).to(DEV)
    else:           # 2d/inference_2d.py:267-279 (the transposed-conv upsampling and the temporal attention's bias gather are in this one)
        net = sdc.Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2, 4), channels=7)
        shape = (8, 7, 16, 16)
        mk = lambda n: sdc.GaussianDiffusionSmoke(n, image_size=16, frames=8, timesteps=50, loss_type="l2").to(DEV)
    net.load_state_dict(det_params(_spec(net), 41))
    net.to(DEV)
    gd = mk(net)
    B = 8
    state = det_tensor((B, *shape), 42, 0.3).to(DEV)
    w = (det_tensor((B,), 43, 0.2) + 1.0).to(DEV)
    t = torch.randint(0, 50, (B,), generator=torch.Generator().manual_seed(44)).to(DEV)
    noise = det_tensor((B, *shape), 45).to(DEV)
    params = [p for p in net.parameters() if p.requires_grad]

    def eager(st_, w_, t_, n_):
        for p in params:
            p.grad = None
        loss = (w_ * gd.p_losses(st_, t_, noise=n_, mean=False)).mean()
        loss.backward()
        return loss.detach().clone(), [p.grad.detach().clone() for p in params]
    l0, g0 = eager(state, w, t, noise)
    step = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise)
    l1 = step(state, w, t, noise).clone()
    assert torch.equal(l0, l1)
    worst = max((a - b).abs().max().item() for a, b in zip(g0, step.grads))
    print(f"[measured] {tree}: graphed vs eager fine-tuning step: loss {l1.item():.6f}, max |grad diff| {worst:.1e}")
    assert worst == 0.0
    # new inputs through the captured buffers
    state2, t2 = det_tensor((B, *shape), 46, 0.3).to(DEV), torch.randint(0, 50, (B,), generator=torch.Generator().manual_seed(47)).to(DEV)
    l2 = step(state2, None, t2, None).clone()
    g2 = [g.clone() for g in step.grads]
    le, ge = eager(state2, w, t2, noise)
    assert torch.equal(l2, le) and all(torch.equal(a, b) for a, b in zip(g2, ge))
    # an eager optimizer step between two replays is seen: the replay after it equals an eager step on the new weights
    for p in params:
        p.grad = None
    for p, g in zip(params, g2):
        p.grad = g.clone()
    torch.optim.SGD(params, lr=1e-4).step()                   # (a small step: the random-init net blows up under a large one)
    l3 = step(state2, None, t2, None).clone()
    g3 = [g.clone() for g in step.grads]
    le3, ge3 = eager(state2, w, t2, noise)
    assert not torch.equal(l3, l2) and torch.equal(l3, le3) and all(torch.equal(a, b) for a, b in zip(g3, ge3))
    # ... and the sampler that shares the net follows (content stamp), as the reference's alternation sample <-> fine-tune needs
    x, tt = det_tensor((2, *shape), 48).to(DEV), torch.tensor([3, 40], device=DEV)
    if tree == "smoke":
        fresh = sdc.Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2, 4), channels=7)
    else:
        fresh = type(net)(dim=net.dim, dim_mults=(1, 2, 4, 8), channels=shape[0], resnet_block_groups=1)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    fresh.to(DEV)
    assert torch.equal(net(x, tt), fresh(x, tt))


@pytest.mark.parametrize("tree", ["tokamak", "burgers"])
def test_forked_conv_backward_equals_serial(tree):
    """ConvFn.backward runs the weight gradient on a side stream beside the data gradient (autograd.OVERLAP_WGRAD): the same
    kernels on the same operands, so loss and every parameter gradient equal the serial order's bit for bit -- three times in a
    row (a missing join or a buffer recycled across the two streams would show as a difference between repeats)."""
    from safediffcon_amd import autograd as ag
    torch.manual_seed(0)
    if tree == "tokamak":
        net = sdc.Unet1D(dim=32, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        shape = (12, 128)
        gd = sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=50)
    else:
        net = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
        shape = (3, 16, 128)
        gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=50, temporal=True, use_conv2d=True,
                                          is_condition_u0=True, is_condition_uT=True, condition_idx=10)
    net.load_state_dict(det_params(_spec(net), 61))
    gd = gd.to(DEV)
    B = 16
    state = det_tensor((B, *shape), 62, 0.3).to(DEV)
    t = torch.randint(0, 50, (B,), generator=torch.Generator().manual_seed(63)).to(DEV)
    noise = det_tensor((B, *shape), 64).to(DEV)
    params = [p for p in net.parameters() if p.requires_grad]

    def run():
        for p in params:
            p.grad = None
        loss = gd.p_losses(state, t, noise=noise, mean=False).mean()
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), [p.grad.detach().clone() for p in params]
    assert ag.OVERLAP_WGRAD
    try:
        ag.OVERLAP_WGRAD = False
        l0, g0 = run()
    finally:
        ag.OVERLAP_WGRAD = True
    for _ in range(3):
        l1, g1 = run()
        assert torch.equal(l0, l1)
        assert all(torch.equal(a, b) for a, b in zip(g0, g1))


@pytest.mark.parametrize("rows,K,M", [(64, 1024, 4096), (64, 256, 128), (1, 8, 32), (100, 260, 36), (16, 32, 16), (130, 64, 1028)])
def test_linear_kernels_against_fp64(rows, K, M):
    """sdc_linear / sdc_linear_dgrad / sdc_linear_wgrad (the MLPs of the fine-tuning path) against fp64 torch on the same operands,
    ragged sizes included (rows not a multiple of 16 / 64, K and M not multiples of 16 / 64 / 256); strided rows; bit-reproducible"""
    from safediffcon_amd import _lib
    lib = _lib.get_lib()
    st = torch.cuda.current_stream().cuda_stream
    x = det_tensor((rows, K + 4), 71).to(DEV)[:, :K]            # row stride K + 4
    w = det_tensor((M, K), 72, 0.2).to(DEV)
    b = det_tensor((M,), 73).to(DEV)
    gy = det_tensor((rows, M), 74).to(DEV)
    y = torch.full((rows, M + 8), 7.0, device=DEV)
    assert lib.sdc_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, K, M, x.stride(0), y.stride(0), st) == 0
    ref = x.double() @ w.double().t() + b.double()
    tol = 2e-6 * (1 + ref.abs().max().item()) * max(1.0, K ** 0.5 / 8)
    assert (y[:, :M].double() - ref).abs().max().item() <= tol
    assert torch.all(y[:, M:] == 7.0)                                # nothing written past a row
    gx = torch.empty((rows, K), device=DEV)
    assert lib.sdc_linear_dgrad(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), rows, K, M, M, K, st) == 0
    ref = gy.double() @ w.double()
    assert (gx.double() - ref).abs().max().item() <= 2e-6 * (1 + ref.abs().max().item()) * max(1.0, M ** 0.5 / 8)
    gw, gb = torch.empty((M, K), device=DEV), torch.empty(M, device=DEV)
    xc = x.contiguous()
    assert lib.sdc_linear_wgrad(gy.data_ptr(), xc.data_ptr(), gw.data_ptr(), gb.data_ptr(), rows, K, M, M, K, st) == 0
    ref = gy.double().t() @ xc.double()
    assert (gw.double() - ref).abs().max().item() <= 2e-6 * (1 + ref.abs().max().item()) * max(1.0, rows ** 0.5 / 8)
    assert (gb.double() - gy.double().sum(0)).abs().max().item() <= 1e-5 * (1 + rows ** 0.5)
    gw2, gb2 = torch.empty_like(gw), torch.empty_like(gb)
    assert lib.sdc_linear_wgrad(gy.data_ptr(), xc.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), rows, K, M, M, K, st) == 0
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    # shapes outside the contract are refused, not mis-computed
    assert lib.sdc_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, K - 1, M, x.stride(0), y.stride(0), st) == -1 and "multiples of 4" in _lib.last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,c0,c1,cout,H,W", [(4, 512, 0, 512, 2, 16), (8, 256, 0, 256, 4, 32), (4, 256, 256, 256, 4, 32),
                                             (2, 64, 0, 128, 8, 64), (3, 128, 64, 64, 2, 16), (2, 96, 0, 64, 4, 16)])
def test_conv_splitk_equals_plain_conv(B, c0, c1, cout, H, W):
    """sdc_conv_splitk (the fine-tuning path's 3x3 convs on grids that leave most CUs idle: Cin split over 2-8 workgroups per
    tile, partial outputs summed in split order) against sdc_conv on the same operands: same values up to the summation order,
    against fp64 torch, bit-reproducible, strided output view; convs it does not split are exactly sdc_conv."""
    from safediffcon_amd import autograd as ag, grad_ops
    import torch.nn.functional as F
    x = det_tensor((B, c0, 1, H, W), 81).to(DEV)
    x1 = det_tensor((B, c1, 1, H, W), 82).to(DEV) if c1 else None
    w = det_tensor((cout, c0 + c1, 1, 3, 3), 83, 0.05).to(DEV)
    b = det_tensor((cout,), 84).to(DEV)
    wp = grad_ops.pack_conv_weight(w, 4)

    def run(split, out=None):
        ag.SPLIT_SMALL_GRIDS = split
        try:
            return ag.conv_raw(x, wp, b, cout, (1, 3, 3), x1=x1, pad=(0, 1, 1), out=out)
        finally:
            ag.SPLIT_SMALL_GRIDS = True
    y0, y1 = run(False), run(True)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = F.conv2d(xin[:, :, 0].double(), w[:, :, 0].double(), b.double(), padding=1)[:, :, None]
    scale = ref.abs().max().item()
    e0, e1 = (y0.double() - ref).abs().max().item() / scale, (y1.double() - ref).abs().max().item() / scale
    print(f"[measured] split-K conv {c0}+{c1}->{cout} @{H}x{W} B={B}: rel err plain {e0:.1e}, split {e1:.1e}")
    assert e0 < 5e-6 and e1 < 5e-6
    assert torch.equal(y1, run(True))
    big = torch.full((B, cout + 8, 1, H, W), 3.0, device=DEV)            # a channel-slice view as the output
    run(True, out=big[:, 4:4 + cout])
    assert torch.equal(big[:, 4:4 + cout], y1) and torch.all(big[:, :4] == 3.0) and torch.all(big[:, 4 + cout:] == 3.0)
    torch.cuda.synchronize()


def test_graphed_step_with_captured_adam_equals_eager_adam():
    """sdc.GraphedLossStep(optimizer=torch.optim.Adam(capturable=True)): the optimizer step is recorded behind the backward pass
    and replayed with it -- three replays against three eager steps of a twin net with its own Adam: the
    parameters agree to rounding (capturable Adam keeps its step count on the device; same update formula)."""
    torch.manual_seed(0)

    def make():
        net = sdc.Unet1D(dim=32, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
        net.load_state_dict(det_params(_spec(net), 91))
        return net, sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=50).to(DEV)
    (net_a, gd_a), (net_b, gd_b) = make(), make()
    B = 8
    state = det_tensor((B, 12, 128), 92, 0.3).to(DEV)
    w = (det_tensor((B,), 93, 0.2) + 1.0).to(DEV)
    t = torch.randint(0, 50, (B,), generator=torch.Generator().manual_seed(94)).to(DEV)
    noise = det_tensor((B, 12, 128), 95).to(DEV)
    pa = [p for p in net_a.parameters() if p.requires_grad]
    pb = [p for p in net_b.parameters() if p.requires_grad]
    opt_a = torch.optim.Adam(pa, lr=1e-4, capturable=True)
    opt_b = torch.optim.Adam(pb, lr=1e-4, capturable=True)
    step = sdc.GraphedLossStep(gd_a, state, weight=w, t=t, noise=noise, optimizer=opt_a)
    # the warm-up passes of the constructor ran backward only (no optimizer step): both nets still hold the same weights
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))
    losses = []
    for _ in range(3):
        losses.append(step(state, w, t, noise).item())
        for p in pb:
            p.grad = None
        lb = (w * gd_b.p_losses(state, t, noise=noise, mean=False)).mean()
        lb.backward()
        opt_b.step()
        assert abs(losses[-1] - lb.item()) <= 1e-6 * max(1.0, abs(lb.item()))
    worst = max(((a - b).abs().max() / (b.abs().max() + 1e-12)).item() for a, b in zip(pa, pb))
    print(f"[measured] captured Adam vs eager Adam after 3 steps: losses {losses}, worst relative parameter difference {worst:.1e}")
    assert worst < 1e-5
    assert losses[2] < losses[0]
