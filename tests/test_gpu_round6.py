"""-m gpu, round 6 (ADVICE r5 on the captured fine-tuning step):
  * GraphedLossStep pins the net's PackArena: forward-only calls between two replays cannot free the buffers the graph replays on;
  * an eager optimizer's zero_grad(set_to_none=True) between replays does not lose the captured gradients;
  * an optimizer that has already trained keeps its state through the capture's warm-up;
  * the differentiable last DDIM step does not split input channels by batch (a trajectory's bits do not depend on its batch).
Gates: bit equality where the same kernels run on the same buffers."""
import pytest
import torch

import safediffcon_amd as sdc
from oracle.detweights import det_params, det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _spec(net):
    return [(k, tuple(v.shape)) for k, v in net.state_dict().items()]


def _tokamak(dim=32, seed=61, timesteps=50):
    net = sdc.Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
    net.load_state_dict(det_params(_spec(net), seed))
    net.to(DEV)
    return net, sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=timesteps).to(DEV)


def _inputs(B, shape, seed):
    state = det_tensor((B, *shape), seed, 0.3).to(DEV)
    w = (det_tensor((B,), seed + 1, 0.2) + 1.0).to(DEV)
    t = torch.randint(0, 50, (B,), generator=torch.Generator().manual_seed(seed + 2)).to(DEV)
    noise = det_tensor((B, *shape), seed + 3).to(DEV)
    return state, w, t, noise


def test_graphed_step_survives_forward_only_calls_between_replays():
    """three forward-only forward_train calls on the same net used to let PackArena.begin() drop the flipped
    (data-gradient) layouts and reallocate buffer + table under the captured sdc_pack_batch_run.  Now the arena is frozen while a
    GraphedLossStep lives: same tensors before and after, and the replay still equals the eager step bit for bit."""
    net, gd = _tokamak()
    state, w, t, noise = _inputs(8, (12, 128), 62)
    params = [p for p in net.parameters() if p.requires_grad]
    step = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise)
    arena = net._trainer().arena
    assert arena.frozen == 1
    buf_ptr, tab_ptr, n_keys = arena.buf.data_ptr(), arena.table.data_ptr(), len(arena.index)
    l1 = step().clone()
    g1 = [g.clone() for g in step.grads]
    with torch.no_grad():
        for _ in range(4):                                   # validation-style calls: no backward, the flip layouts go idle
            gd.p_losses(state, t, noise=noise, mean=False)
    # junk allocations that would land on a freed arena buffer
    junk = [torch.full((arena.buf.numel(),), float("nan"), device=DEV) for _ in range(3)]
    assert (arena.buf.data_ptr(), arena.table.data_ptr(), len(arena.index)) == (buf_ptr, tab_ptr, n_keys)
    l2 = step().clone()
    assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(g1, step.grads))
    for p in params:
        p.grad = None
    le = (w * gd.p_losses(state, t, noise=noise, mean=False)).mean()
    le.backward()
    assert torch.equal(le.detach(), l2) and all(torch.equal(p.grad, g) for p, g in zip(params, g1))
    del junk
    step.close()
    assert arena.frozen == 0
    with pytest.raises(RuntimeError):
        step()


def test_graphed_step_reseats_grads_after_zero_grad():
    """optimizer.zero_grad() (set_to_none=True by default; the reference's loops call it every iteration) drops
    p.grad; the next replay seats the captured gradient tensors again so an eager optimizer.step() moves the weights"""
    net, gd = _tokamak(seed=63)
    state, w, t, noise = _inputs(4, (12, 128), 64)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-4)
    step = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise)
    step()
    opt.zero_grad()
    assert all(p.grad is None for p in params)
    before = [p.detach().clone() for p in params]
    step()
    assert all(p.grad is g for p, g in zip(params, step.grads))
    opt.step()
    moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, params))
    assert moved > len(params) // 2


def test_graphed_step_keeps_state_of_a_trained_optimizer():
    """an optimizer that has already stepped keeps its moments and step count through the capture's warm-up step"""
    net, gd = _tokamak(seed=65)
    state, w, t, noise = _inputs(4, (12, 128), 66)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-5, capturable=True)
    (w * gd.p_losses(state, t, noise=noise, mean=False)).mean().backward()
    opt.step()
    opt.step()
    snap = {p: {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in opt.state.items()}
    assert all(float(s["step"]) == 2.0 for s in snap.values())
    weights = [p.detach().clone() for p in params]
    step = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise, optimizer=opt)
    for p, st in opt.state.items():                          # (capture itself replays nothing)
        for k, v in snap[p].items():
            assert torch.equal(st[k], v), k
    assert all(torch.equal(a, p.detach()) for a, p in zip(weights, params))
    step()
    assert all(float(st["step"]) == 3.0 for st in opt.state.values())


def test_differentiable_last_step_is_batch_invariant():
    """sample(enable_grad=True) runs its last DDIM step through forward_train; inside it conv_raw must not take the
    batch-dependent split over input channels, so sample 0 of a batch of 2 equals sample 0 of a batch of 16 bit for bit"""
    from safediffcon_amd import autograd
    net = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    net.load_state_dict(det_params(_spec(net), 67))
    net.to(DEV)
    x = det_tensor((16, 3, 16, 128), 68).to(DEV)
    tt = torch.full((16,), 7, device=DEV, dtype=torch.long)
    with autograd.batch_invariant():
        big = net.forward_train(x, tt).detach()
        small = net.forward_train(x[:2].contiguous(), tt[:2]).detach()
    assert torch.equal(big[:2], small)
    # and the fine-tuning path still may split (documented: gradients are gated at 1e-9 MSE against the reference, not bitwise)
    assert autograd.SPLIT_SMALL_GRIDS


def test_t1000_guided_trajectory_at_shipped_width_burgers_turbo():
    """one FULL-schedule (T = 1000) guided DDPM trajectory at the only width the reference ships a checkpoint
    for -- Unet2D(dim=128) "turbo", 1D/configs/inference_config.py:125-134 -- against the oracle's loop + functional net run by
    PyTorch-ROCm eager on the same device (1D/model/diffusion.py:368-449).  Gate: element-wise ~2x the error measured on MI355X
    (printed), and far inside the north star's eps-MSE <= 1e-5."""
    from oracle import nets as onets
    from oracle import samplers as osam
    from oracle import schedules as osched
    from oracle.detweights import det_noise
    net = sdc.Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)
    P = det_params(_spec(net), 71)
    net.load_state_dict(P)
    net.to(DEV)
    T, B = 1000, 2
    gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=T, temporal=True, use_conv2d=True,
                                      is_condition_u0=True, is_condition_uT=True, condition_idx=10,
                                      train_on_padded_locations=False).to(DEV)
    u0, uT = det_tensor((B, 128), 72, 0.1), det_tensor((B, 128), 73, 0.1)
    noise = det_noise((B, 3, 16, 128), 93000)
    Q, w, ub = 0.01, 500.0, 0.3                       # u_bound 0.3: the hinge is active along the way
    out = gd.sample(batch_size=B, clip_denoised=True, u_init=u0, u_final=uT, guidance_u0=True,
                    nablaJ=sdc.BurgersGuidance(Q, w, ub, use_max_safety=True), J_scheduler=None, enable_grad=False, noise=noise).cpu()
    Pg = {k: v.to(DEV) for k, v in P.items()}
    ref = osam.sample_burgers(lambda a, b: onets.unet_burgers(Pg, a, b.to(a.device), dim=128), osched.make_tables("cosine", T), B,
                              lambda i: noise(i).to(DEV), u_init=u0.to(DEV), u_final=uT.to(DEV),
                              nablaJ=osam.burgers_guidance(Q, w, ub, True), enable_grad=False,
                              train_on_padded_locations=False).cpu()
    err = (out - ref).abs().max().item()
    mse = ((out - ref) ** 2).mean().item()
    print(f"[measured] C2-turbo width (dim 128), T = 1000 guided DDPM (B = 2) vs the eager-GPU oracle: max|err| {err:.3e}  MSE {mse:.3e}")
    assert torch.isfinite(out).all() and err < 8e-6 and mse <= 2e-13          # measured on MI355X: 3.8e-6, 8.2e-14


# ------------------------------------------------------------------ the small-batch sampler plan
@pytest.mark.parametrize("B,c0,c1,cout,L", [(16, 2048, 0, 2048, 16), (16, 1024, 1024, 1024, 32), (16, 256, 0, 256, 128), (8, 512, 0, 1024, 64),
                                            (16, 96, 0, 64, 128), (300, 256, 0, 256, 128)])
def test_conv1d_splitk_equals_plain_conv(B, c0, c1, cout, L):
    """sdc_conv_splitk on the 1-D F(2,3) form (tokamak Block's Conv1d k3 at a per-rank batch of 16: 64 workgroups per layer):
    Cin split over the workgroups sdc_conv_splitk_bytes sizes, against sdc_conv on the same operands and against fp64 torch;
    bit-reproducible; a conv it does not split (odd stage count, a grid that fills the chip) is exactly sdc_conv."""
    import ctypes as C
    import torch.nn.functional as F
    from safediffcon_amd import autograd as ag, grad_ops, _lib
    x = det_tensor((B, c0, 1, 1, L), 81).to(DEV)
    x1 = det_tensor((B, c1, 1, 1, L), 82).to(DEV) if c1 else None
    w = det_tensor((cout, c0 + c1, 1, 1, 3), 83, 0.05).to(DEV)
    b = det_tensor((cout,), 84).to(DEV)
    wp = grad_ops.pack_conv_weight(w, 4)

    def run(split):
        ag.SPLIT_SMALL_GRIDS = split
        try:
            return ag.conv_raw(x, wp, b, cout, (1, 1, 3), x1=x1, pad=(0, 0, 1))
        finally:
            ag.SPLIT_SMALL_GRIDS = True
    y0, y1 = run(False), run(True)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = F.conv1d(xin[:, :, 0, 0].double(), w[:, :, 0, 0].double(), b.double(), padding=1)[:, :, None, None]
    scale = ref.abs().max().item()
    e0, e1 = (y0.double() - ref).abs().max().item() / scale, (y1.double() - ref).abs().max().item() / scale
    splits = not torch.equal(y0, y1)
    print(f"[measured] 1-D split-K conv {c0}+{c1}->{cout} L={L} B={B}: rel err plain {e0:.1e}, split {e1:.1e} ({'split' if splits else 'not split'})")
    assert e0 < 5e-6 and e1 < 5e-6
    assert torch.equal(y1, run(True))
    expect_split = (c0 + c1) % 64 == 0 and ((B * L + 127) // 128) * (cout // 64) <= 128
    assert splits == expect_split
    torch.cuda.synchronize()


@pytest.mark.parametrize("tree,dim,B", [("burgers", 64, 32), ("tokamak", 256, 16)])
def test_small_batch_plan_eps_vs_oracle_and_graph(tree, dim, B):
    """net.split_small_grids at the per-rank batch of an 8-way shard (1D/model/unet.py:382-426,
    tokamak/model/unet.py:359-408): eps against the eager-GPU oracle net (gate: eps-MSE <= 1e-9, the split changes the
    summation order), the plan does call sdc_conv_splitk, the graph-replayed forward equals the launched call list bit for bit,
    and with the switch off the same input gives the batch-invariant result."""
    from oracle import nets as onets
    net = (sdc.Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1) if tree == "burgers"
           else sdc.Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1))
    P = det_params(_spec(net), 75)
    net.load_state_dict(P)
    net.to(DEV)
    shape = (B, 3, 16, 128) if tree == "burgers" else (B, 12, 128)
    x = det_tensor(shape, 76).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(77)).to(DEV)
    Pg = {k: v.to(DEV) for k, v in P.items()}
    fwd = onets.unet_burgers if tree == "burgers" else onets.unet_tokamak
    with torch.no_grad():
        ref = fwd(Pg, x, t, dim=dim)
    plain = net(x, t).clone()
    net.split_small_grids = True
    got = net(x, t).clone()
    used = [fn.__name__ for fn, _ in net.entry(shape, B)["plan"].calls]
    nsplit = used.count("sdc_conv_splitk")
    mse, mse0 = ((got - ref) ** 2).mean().item(), ((plain - ref) ** 2).mean().item()
    print(f"[measured] {tree} dim {dim} B = {B}: small-batch plan eps-MSE {mse:.2e} (plain {mse0:.2e}), {nsplit} convs split of "
          f"{sum(u.startswith('sdc_conv') for u in used)}")
    assert nsplit >= 8 and mse <= 1e-9 and mse0 <= 1e-9
    assert not torch.equal(got, plain)
    net.forward_graph = False
    eager = net(x, t).clone()
    net.forward_graph = True
    assert torch.equal(got, eager)
    # the first two samples riding in a smaller batch: bit-equal with the switch off, not required with it on
    net.split_small_grids = False
    assert torch.equal(net(x[:2].contiguous(), t[:2]), plain[:2])


@pytest.mark.parametrize("B,cin,cout,L", [(16, 2048, 1024, 16), (16, 512, 256, 64), (300, 512, 256, 64)])
def test_conv1d_upsample_splitk_equals_plain_conv(B, cin, cout, L):
    """tokamak Upsample (nearest x2 + Conv1d k3, tokamak/model/unet.py:24-28) with the upsampling folded into the conv's gather:
    sdc_conv_splitk splits its input channels at small batch like the plain 1-D convs; against sdc_conv and fp64 torch."""
    import torch.nn.functional as F
    from safediffcon_amd import autograd as ag, grad_ops
    x = det_tensor((B, cin, 1, 1, L), 85).to(DEV)
    w = det_tensor((cout, cin, 1, 1, 3), 86, 0.05).to(DEV)
    b = det_tensor((cout,), 87).to(DEV)
    wp = grad_ops.pack_conv_weight(w, 4)

    def run(split):
        ag.SPLIT_SMALL_GRIDS = split
        try:
            return ag.conv_raw(x, wp, b, cout, (1, 1, 3), pad=(0, 0, 1), up=(1, 1, 2))
        finally:
            ag.SPLIT_SMALL_GRIDS = True
    y0, y1 = run(False), run(True)
    ref = F.conv1d(F.interpolate(x[:, :, 0, 0].double(), scale_factor=2, mode="nearest"), w[:, :, 0, 0].double(), b.double(), padding=1)[:, :, None, None]
    scale = ref.abs().max().item()
    e0, e1 = (y0.double() - ref).abs().max().item() / scale, (y1.double() - ref).abs().max().item() / scale
    splits = not torch.equal(y0, y1)
    print(f"[measured] 1-D upsample split-K conv {cin}->{cout} L={L}->{2 * L} B={B}: rel err plain {e0:.1e}, split {e1:.1e} ({'split' if splits else 'not split'})")
    assert e0 < 5e-6 and e1 < 5e-6 and torch.equal(y1, run(True))
    assert splits == (B < 100)


# ------------------------------------------------------------------ the last GroupNorm apply inside the output conv
@pytest.mark.parametrize("B,Cc,G,cout,sp,frame_major,res", [(2, 64, 8, 7, (4, 16, 16), True, True), (3, 64, 1, 3, (1, 16, 128), False, True),
                                                            (2, 256, 1, 12, (1, 1, 128), False, True), (1, 24, 3, 16, (2, 8, 12), False, False)])
def test_gn_pointwise_out_matches_fp64(B, Cc, G, cout, sp, frame_major, res):
    """sdc_gn_pointwise_out: y = conv1x1(SiLU(GroupNorm(h)) + residual) in one pass (conv3d.py:468-471, 1D/model/unet.py:376-378)
    against torch in fp64; output through strides (frame-major eps of the smoke net), Cout 3 / 7 / 12 / 16."""
    import torch.nn.functional as F
    from safediffcon_amd import _lib
    lib = _lib.get_lib()
    h = det_tensor((B, Cc, *sp), 201).to(DEV)
    r = det_tensor((B, Cc, *sp), 202).to(DEV) if res else None
    gam, bet = (det_tensor((Cc,), 203, 0.3) + 1.0).to(DEV), det_tensor((Cc,), 204, 0.2).to(DEV)
    w, b = det_tensor((cout, Cc), 205, 0.2).to(DEV), det_tensor((cout,), 206, 0.1).to(DEV)
    S = sp[0] * sp[1] * sp[2]
    st = torch.empty((int(lib.sdc_gn_stats_bytes(B, G)) + 3) // 4, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.sdc_gn_stats(h.data_ptr(), st.data_ptr(), B, Cc, G, S, 1e-5, stream) == 0
    if frame_major:      # (B, F, C, H, W) storage viewed as (B, C, F, H, W)
        store = torch.full((B, sp[0], cout, sp[1], sp[2]), 7.0, device=DEV)
        out = store.permute(0, 2, 1, 3, 4)
    else:
        out = torch.full((B, cout, *sp), 7.0, device=DEV)
    ys = out.stride()
    rc = lib.sdc_gn_pointwise_out(h.data_ptr(), st.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0 if r is None else r.data_ptr(), w.data_ptr(),
                                  b.data_ptr(), out.data_ptr(), B, Cc, G, cout, S, sp[1] * sp[2], ys[0], ys[1], ys[2], stream)
    assert rc == 0, _lib.last_error()
    v = F.silu(F.group_norm(h.double(), G, gam.double(), bet.double(), 1e-5))
    if r is not None:
        v = v + r.double()
    ref = torch.einsum("oc,bcdhw->bodhw", w.double(), v) + b.double().view(1, -1, 1, 1, 1)
    e = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[measured] gn_pointwise_out C={Cc} G={G} -> {cout} {sp}: rel err vs fp64 {e:.2e}")
    assert e < 3e-6
    # contract violations are refused, not mis-computed
    assert lib.sdc_gn_pointwise_out(h.data_ptr(), st.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0, w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                    B, Cc, G, 17, S, sp[1] * sp[2], ys[0], ys[1], ys[2], stream) == -1 and "Cout <= 16" in _lib.last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("tree", ["smoke", "burgers", "tokamak"])
def test_fused_final_conv_equals_the_two_pass_form(tree):
    """net.fuse_final_conv (default on): eps with the last GroupNorm apply inside final_conv against the same net with the separate
    sdc_gn_apply + sdc_conv passes -- same values up to the order of the 1x1 conv's sums -- and one launch fewer of sdc_gn_apply."""
    if tree == "smoke":
        net, shape = sdc.Unet3D_with_Conv3D(dim=16, dim_mults=(1, 2, 4), channels=7), (2, 4, 7, 16, 16)
    elif tree == "burgers":
        net, shape = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1), (2, 3, 16, 128)
    else:
        net, shape = sdc.Unet1D(dim=32, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1), (2, 12, 128)
    net.load_state_dict(det_params(_spec(net), 78))
    net.to(DEV)
    x, t = det_tensor(shape, 79).to(DEV), torch.tensor([3, 700], device=DEV)
    fused = net(x, t).clone()
    names = [fn.__name__ for fn, _ in net.entry(shape, 2)["plan"].calls]
    # (the tokamak net's 128 positions per sample are below the kernel's threshold: its plan keeps the two-pass form by itself)
    assert names.count("sdc_gn_pointwise_out") == (0 if tree == "tokamak" else 1)
    net.fuse_final_conv = False
    plain = net(x, t).clone()
    names0 = [fn.__name__ for fn, _ in net.entry(shape, 2)["plan"].calls]
    assert "sdc_gn_pointwise_out" not in names0
    err = (fused - plain).abs().max().item()
    print(f"[measured] {tree}: fused final conv vs two-pass form: max|diff| {err:.2e} (|eps|max {plain.abs().max().item():.2f})")
    assert err <= 2e-6 * max(1.0, plain.abs().max().item())


def test_t1000_guided_trajectory_at_shipped_width_tokamak_turbo():
    """the tokamak counterpart: T = 1000 guided DDPM at `Unet1D(dim=128)` "turbo" (tokamak/configs/inference_config.py:118-141)
    against the oracle's loop + functional net under PyTorch-ROCm eager (tokamak/model/diffusion.py:310-372)."""
    from oracle import nets as onets
    from oracle import samplers as osam
    from oracle import schedules as osched
    from oracle.detweights import det_noise
    net = sdc.Unet1D(dim=128, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)
    P = det_params(_spec(net), 81)
    net.load_state_dict(P)
    net.to(DEV)
    T, B = 1000, 2
    gd = sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=T).to(DEV)
    u0 = det_tensor((B, 3), 82, 0.1) + 0.6
    uT = det_tensor((B, 2, 122), 83, 0.1) + 0.6
    target = det_tensor((B, 3, 122), 84, 0.3) + 1.0
    noise = det_noise((B, 12, 128), 94000)
    args = dict(w_obj=0.3, w_safe=1.0, guidance_scaler=0.5, Q=0.05, safety_threshold=4.98)
    out = gd.sample(batch_size=B, u_init=u0, u_final=uT, nablaJ=sdc.TokamakGuidance(target, 122, **args), enable_grad=False,
                    noise=noise).cpu()
    Pg = {k: v.to(DEV) for k, v in P.items()}
    ref = osam.sample_tokamak(lambda a, b: onets.unet_tokamak(Pg, a, b.to(a.device), dim=128), osched.make_tables("cosine", T), B,
                              lambda i: noise(i).to(DEV), u_init=u0.to(DEV), u_final=uT.to(DEV),
                              nablaJ=osam.tokamak_guidance(target.to(DEV), 122, 0.05, 4.98, 0.3, 1.0, 0.5), enable_grad=False).cpu()
    err = (out - ref).abs().max().item()
    mse = ((out - ref) ** 2).mean().item()
    print(f"[measured] C3-turbo width (dim 128), T = 1000 guided DDPM (B = 2) vs the eager-GPU oracle: max|err| {err:.3e}  MSE {mse:.3e}")
    assert torch.isfinite(out).all() and err < 7e-6 and mse <= 2e-13          # measured on MI355X: 3.0e-6, 6.9e-14


@pytest.mark.parametrize("case", [
    dict(B=16, cin=2048, cout=384, sp=(1, 1, 16), k=(1, 1, 1), tag="1x1 to_qkv, tokamak mid level at B = 16"),
    dict(B=16, cin=1024, cin1=1024, cout=1024, sp=(1, 1, 32), k=(1, 1, 1), tag="1x1 res_conv over a concat"),
    dict(B=16, cin=512, cout=512, sp=(1, 1, 64), k=(1, 1, 4), stride=(1, 1, 2), pad=(0, 0, 1), tag="Conv1d k4 s2 (tokamak Downsample)"),
    dict(B=32, cin=512, cout=256, sp=(1, 2, 16), k=(1, 2, 2), pad=(0, 1, 1), parity=True, tag="sub-pixel 2x2 conv into a parity view (Burgers Upsample2d)"),
    dict(B=300, cin=256, cout=256, sp=(1, 1, 128), k=(1, 1, 1), split=False, tag="a grid that fills the chip: not split"),
])
def test_conv_direct_splitk_equals_plain_conv(case):
    """sdc_conv_splitk on the direct-form kernels' smallest tile (conv_pw_kernel<64,64>, conv_kernel<64,64,FAST>): 1x1 convs, the strided
    Conv1d k4 and the sub-pixel 2x2 convs at the per-rank batches of an 8-way shard -- against sdc_conv on the same operands and fp64."""
    import torch.nn.functional as F
    from safediffcon_amd import autograd as ag, grad_ops
    B, cin, cin1, cout, sp, k = case["B"], case["cin"], case.get("cin1", 0), case["cout"], case["sp"], case["k"]
    stride, pad = case.get("stride", (1, 1, 1)), case.get("pad", (0, 0, 0))
    x = det_tensor((B, cin, *sp), 91).to(DEV)
    x1 = det_tensor((B, cin1, *sp), 92).to(DEV) if cin1 else None
    w = det_tensor((cout, cin + cin1, *k), 93, 0.05).to(DEV)
    b = det_tensor((cout,), 94).to(DEV)
    wp = grad_ops.pack_conv_weight(w, 4)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = F.conv3d(xin.double(), w.double(), b.double(), stride=stride, padding=pad)
    if case.get("parity"):
        ref = ref[:, :, :, :sp[1], :sp[2]]                          # the output rows / columns a parity sub-grid holds

    def run(split):
        ag.SPLIT_SMALL_GRIDS = split
        try:
            if case.get("parity"):
                big = torch.zeros(B, cout, sp[0], 2 * sp[1], 2 * sp[2], device=DEV)
                ag.conv_raw(x, wp, b, cout, k, x1=x1, stride=stride, pad=pad, out=big[:, :, :, 0::2, 0::2])
                return big[:, :, :, 0::2, 0::2].clone(), big
            return ag.conv_raw(x, wp, b, cout, k, x1=x1, stride=stride, pad=pad), None
        finally:
            ag.SPLIT_SMALL_GRIDS = True
    (y0, _), (y1, big1) = run(False), run(True)
    scale = ref.abs().max().item()
    e0, e1 = (y0.double() - ref).abs().max().item() / scale, (y1.double() - ref).abs().max().item() / scale
    splits = not torch.equal(y0, y1)
    print(f"[measured] direct split-K {case['tag']}: rel err plain {e0:.1e}, split {e1:.1e} ({'split' if splits else 'not split'})")
    assert e0 < 5e-6 and e1 < 5e-6 and torch.equal(y1, run(True)[0])
    assert splits == case.get("split", True)
    if big1 is not None:                                            # nothing written beside the parity sub-grid
        assert torch.all(big1[:, :, :, 1::2, :] == 0) and torch.all(big1[:, :, :, :, 1::2] == 0)


def test_fused_final_conv_falls_back_outside_the_kernels_contract():
    """a net with more than 16 output channels: Plan.gn_pointwise_out declines, final_conv finishes the last ResnetBlock with the
    block's own apply pass (from the statistics already taken) and runs the plain conv -- same result as with the switch off"""
    net = sdc.Unet2D(dim=16, dim_mults=(1, 2, 4, 8), channels=20, resnet_block_groups=1)
    net.load_state_dict(det_params(_spec(net), 88))
    net.to(DEV)
    shape = (2, 20, 16, 128)
    x, t = det_tensor(shape, 89).to(DEV), torch.tensor([5, 600], device=DEV)
    a = net(x, t).clone()
    names = [fn.__name__ for fn, _ in net.entry(shape, 2)["plan"].calls]
    assert "sdc_gn_pointwise_out" not in names and torch.isfinite(a).all()
    net.fuse_final_conv = False
    b = net(x, t).clone()
    err = (a - b).abs().max().item()
    print(f"[measured] final conv fallback vs plain path: max|diff| {err:.2e}")
    assert err <= 1e-6 * max(1.0, b.abs().max().item())
