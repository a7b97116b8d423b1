"""-m gpu: the backward (VJP) kernels of the fine-tuning path against torch autograd in fp64 (include/sdc.h "backward"):
weight / bias gradient of every conv form the three U-Nets hold, GroupNorm + scale/shift + SiLU backward, channel
LayerNorm / RMSNorm backward, SiLU / GELU backward, the nearest-upsample VJP.  Tolerances are relative to the gradient's
scale (fp32 kernels, fp64 reference: summation order only)."""
import pytest
import torch
import torch.nn.functional as F

from oracle.detweights import det_tensor
from safediffcon_amd import grad_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


CASES = [
    # nd, B, cin, cout, spatial, k, stride, pad
    dict(nd=3, B=2, cin=64, cout=64, sp=(4, 16, 32), k=3, pad=1),                      # smoke ResnetBlock conv (3 taps, 9 (kd, kh))
    dict(nd=3, B=1, cin=7, cout=64, sp=(8, 16, 16), k=7, pad=3),                       # smoke stem, ragged Cin
    dict(nd=3, B=2, cin=96, cout=40, sp=(2, 16, 16), k=3, pad=1),                      # ragged M and N tiles
    dict(nd=2, B=3, cin=128, cout=64, sp=(16, 128), k=3, pad=1),                       # Burgers 3x3
    dict(nd=2, B=2, cin=3, cout=64, sp=(16, 128), k=7, pad=3),                         # Burgers stem
    dict(nd=1, B=5, cin=256, cout=512, sp=(32,), k=3, pad=1),                          # tokamak k3
    dict(nd=1, B=4, cin=256, cout=512, sp=(64,), k=4, stride=2, pad=1),                # tokamak Downsample (k4 s2): oW = 32
    dict(nd=1, B=3, cin=12, cout=256, sp=(128,), k=7, pad=3),                          # tokamak stem
    dict(nd=2, B=2, cin=64, cout=384, sp=(16, 128), k=1),                              # 1x1 / to_qkv
    dict(nd=3, B=2, cin=64, cout=64, sp=(2, 32, 32), k=(1, 4, 4), stride=(1, 2, 2), pad=(0, 1, 1)),   # smoke Downsample
    # merged-kh form (3x3 'same', rows of 16 / 32 / 64): splits that start and end inside a (b, od) block, ragged tiles,
    # single-row and two-row images, 1 x 3 x 3 taps
    dict(nd=3, B=3, cin=72, cout=100, sp=(3, 10, 64), k=3, pad=1),
    dict(nd=2, B=5, cin=64, cout=64, sp=(6, 16), k=3, pad=1),
    dict(nd=2, B=7, cin=32, cout=48, sp=(1, 32), k=3, pad=1),
    dict(nd=3, B=1, cin=64, cout=64, sp=(5, 2, 64), k=(1, 3, 3), pad=(0, 1, 1)),
    dict(nd=3, B=2, cin=128, cout=128, sp=(4, 64, 64), k=3, pad=1),
    # rows of 128 / 192 through the merged-kh kernel as chunks of 64 columns (each chunk one more partial copy), bias included
    dict(nd=2, B=2, cin=40, cout=72, sp=(5, 192), k=3, pad=1),
    dict(nd=3, B=1, cin=64, cout=64, sp=(2, 6, 128), k=3, pad=1),
    # ... and its tap-folded 7 x 7 form (stems): seven row taps per workgroup, eight-slot X ring
    dict(nd=3, B=2, cin=7, cout=64, sp=(3, 20, 64), k=(1, 7, 7), pad=(0, 3, 3)),
    dict(nd=2, B=2, cin=3, cout=64, sp=(9, 64), k=7, pad=3),
    dict(nd=2, B=3, cin=9, cout=40, sp=(4, 32), k=7, pad=3),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_wgrad_vs_autograd(case):
    nd, B, cin, cout, sp = case["nd"], case["B"], case["cin"], case["cout"], case["sp"]
    k = case["k"] if isinstance(case["k"], tuple) else (case["k"],) * nd
    st = case.get("stride", 1)
    st = st if isinstance(st, tuple) else (st,) * nd
    pd = case.get("pad", 0)
    pd = pd if isinstance(pd, tuple) else (pd,) * nd
    x = det_tensor((B, cin, *sp), 91)
    w = det_tensor((cout, cin, *k), 92, 0.1).double().requires_grad_()
    b = det_tensor((cout,), 93, 0.1).double().requires_grad_()
    y = (F.conv1d, F.conv2d, F.conv3d)[nd - 1](x.double(), w, b, stride=st, padding=pd)
    gy = det_tensor(tuple(y.shape), 94)
    y.backward(gy.double())
    k3, s3, p3 = (1,) * (3 - nd) + k, (1,) * (3 - nd) + st, (0,) * (3 - nd) + pd
    dw, db = grad_ops.conv_wgrad(gy.to(DEV), x.to(DEV), k3, s3, p3)
    ew, eb = _rel(dw.reshape(w.shape), w.grad), _rel(db, b.grad)
    print(f"[measured] wgrad {case}: rel err dw {ew:.2e} db {eb:.2e}")
    assert ew < 2e-5 and eb < 2e-5


def test_conv_transpose_and_upsample_wgrad():
    """ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1) (smoke Upsample): G = x, X = dL/dy;  nn.Upsample(2) + Conv2d 3x3 (Burgers
    Upsample2d) and Upsample + Conv1d k3 (tokamak): X read through the folded nearest upsampling; strided G views."""
    x = det_tensor((2, 64, 2, 16, 16), 95)
    w = det_tensor((64, 32, 1, 4, 4), 96, 0.1).double().requires_grad_()
    y = F.conv_transpose3d(x.double(), w, None, stride=(1, 2, 2), padding=(0, 1, 1))
    gy = det_tensor(tuple(y.shape), 97)
    y.backward(gy.double())
    dw, _ = grad_ops.conv_wgrad(x.to(DEV), gy.to(DEV), (1, 4, 4), (1, 2, 2), (0, 1, 1), bias=False)
    assert _rel(dw, w.grad) < 2e-5

    x = det_tensor((2, 64, 8, 32), 98)
    w = det_tensor((32, 64, 3, 3), 99, 0.1).double().requires_grad_()
    y = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w, None, padding=1)
    gy = det_tensor(tuple(y.shape), 100)
    y.backward(gy.double())
    dw, _ = grad_ops.conv_wgrad(gy.to(DEV), x.to(DEV), (1, 3, 3), (1, 1, 1), (0, 1, 1), up=(1, 2, 2), bias=False)
    assert _rel(dw.reshape(w.shape), w.grad) < 2e-5

    x = det_tensor((3, 128, 16), 101)
    w = det_tensor((64, 128, 3), 102, 0.1).double().requires_grad_()
    y = F.conv1d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w, None, padding=1)
    gy = det_tensor(tuple(y.shape), 103)
    y.backward(gy.double())
    dw, _ = grad_ops.conv_wgrad(gy.to(DEV), x.to(DEV), (1, 1, 3), (1, 1, 1), (0, 0, 1), up=(1, 1, 2), bias=False)
    assert _rel(dw.reshape(w.shape), w.grad) < 2e-5

    # a strided gradient view (frame-major smoke state: eps gradient arrives as (B,F,C,H,W).permute)
    gfm = det_tensor((2, 8, 7, 16, 16), 104)
    xin = det_tensor((2, 64, 8, 16, 16), 105)
    w = det_tensor((7, 64, 1, 1, 1), 106, 0.1).double().requires_grad_()
    y = F.conv3d(xin.double(), w)
    y.backward(gfm.permute(0, 2, 1, 3, 4).double())
    dw, db = grad_ops.conv_wgrad(gfm.to(DEV).permute(0, 2, 1, 3, 4), xin.to(DEV), (1, 1, 1))
    assert _rel(dw, w.grad) < 2e-5 and _rel(db, gfm.double().sum((0, 1, 3, 4))) < 2e-5


@pytest.mark.parametrize("shape,G,cond,res", [((3, 64, 4, 16, 16), 8, True, True), ((2, 128, 1, 8, 64), 1, True, False),
                                              ((4, 256, 1, 1, 32), 1, False, True), ((2, 64, 8, 32, 32), 8, False, False),
                                              # ragged: row counts that are not multiples of 4, rows that are not multiples of 64 / 256
                                              ((3, 70, 1, 1, 19), 7, True, True), ((5, 24, 1, 3, 37), 8, True, False),
                                              ((1, 16, 2, 24, 25), 4, False, False), ((67, 6, 1, 1, 16), 1, True, False),
                                              # long rows, few of them: the row cut into pieces over several workgroups
                                              ((2, 16, 16, 32, 32), 4, True, True), ((1, 8, 9, 52, 40), 8, False, False)])
def test_gn_silu_backward(shape, G, cond, res):
    B, Cc = shape[0], shape[1]
    h = det_tensor(shape, 110)
    gamma, beta = 1 + 0.1 * det_tensor((Cc,), 111), 0.1 * det_tensor((Cc,), 112)
    ss = 0.2 * det_tensor((B, 2 * Cc), 113) if cond else None
    r = det_tensor(shape, 114) if res else None
    gy = det_tensor(shape, 115)
    hd = h.double().requires_grad_()
    gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
    sd = ss.double().requires_grad_() if cond else None
    u = F.group_norm(hd, G, gd, bd, eps=1e-5)
    if cond:
        bshape = (B, Cc) + (1,) * (len(shape) - 2)
        u = u * (sd[:, :Cc].reshape(bshape) + 1) + sd[:, Cc:].reshape(bshape)
    y = F.silu(u)
    y.backward(gy.double())
    hg = h.to(DEV)
    st = grad_ops.gn_stats(hg, G)
    y_hip = grad_ops.gn_apply(hg, st, gamma.to(DEV), beta.to(DEV), G, None if ss is None else ss.to(DEV), None if r is None else r.to(DEV))
    want = y.detach() + (r.double() if res else 0)
    assert _rel(y_hip, want) < 1e-5
    gh, dg, db, dss = grad_ops.gn_silu_bwd(hg, gy.to(DEV), st, gamma.to(DEV), beta.to(DEV), G, None if ss is None else ss.to(DEV))
    errs = dict(gh=_rel(gh, hd.grad), dgamma=_rel(dg, gd.grad), dbeta=_rel(db, bd.grad))
    if cond:
        errs["dss"] = _rel(dss, sd.grad)
    print(f"[measured] gn_silu_bwd {shape} G={G}: {errs}")
    assert max(errs.values()) < 5e-5


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(2, 64, 4, 16, 16), (3, 256, 1, 1, 40)])
def test_chan_norm_backward(mode, shape):
    x = det_tensor(shape, 120)
    g = 1 + 0.1 * det_tensor((shape[1],), 121)
    gy = det_tensor(shape, 122)
    xd, gd = x.double().requires_grad_(), g.double().requires_grad_()
    gb = gd.reshape(1, -1, *([1] * (len(shape) - 2)))
    if mode == 0:
        var = xd.var(dim=1, unbiased=False, keepdim=True)
        y = (xd - xd.mean(dim=1, keepdim=True)) * (var + 1e-5).rsqrt() * gb
    else:
        y = F.normalize(xd, dim=1) * gb * (shape[1] ** 0.5)
    y.backward(gy.double())
    gx, dgain = grad_ops.chan_norm_bwd(x.to(DEV), gy.to(DEV), g.to(DEV), mode)
    e1, e2 = _rel(gx, xd.grad), _rel(dgain, gd.grad)
    print(f"[measured] chan_norm_bwd mode {mode} {shape}: gx {e1:.2e} dgain {e2:.2e}")
    assert e1 < 2e-5 and e2 < 2e-5


def test_act_backward_and_sumpool():
    x, gy = det_tensor((5, 333), 130, 2.0), det_tensor((5, 333), 131)
    for kind, fn in ((0, F.silu), (1, F.gelu)):
        xd = x.double().requires_grad_()
        fn(xd).backward(gy.double())
        assert _rel(grad_ops.act_bwd(x.to(DEV), gy.to(DEV), kind), xd.grad) < 2e-6
    g = det_tensor((2, 3, 8, 32), 132)
    xd = det_tensor((2, 3, 4, 16), 133).double().requires_grad_()
    F.interpolate(xd, scale_factor=2, mode="nearest").backward(g.double())
    assert _rel(grad_ops.sumpool2(g.to(DEV), 2, 2), xd.grad) < 1e-6
    g1 = det_tensor((2, 3, 32), 134)
    xd = det_tensor((2, 3, 16), 135).double().requires_grad_()
    F.interpolate(xd, scale_factor=2, mode="nearest").backward(g1.double())
    assert _rel(grad_ops.sumpool2(g1.to(DEV).unsqueeze(-2), 1, 2).squeeze(-2), xd.grad) < 1e-6


# ------------------------------------------------------------------ attention cores
def _rot_table(F_, dev="cpu"):
    fr = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    ang = torch.arange(F_, dtype=torch.float32)[:, None] * fr[None, :]
    return torch.stack((ang.cos(), ang.sin()), dim=-1).reshape(-1).to(dev), ang.repeat_interleave(2, dim=-1).double()


def _ref_attention(q, k, v, ang=None, bias=None):
    """q, k, v (..., heads, n, 32) fp64"""
    q = q * 32 ** -0.5
    if ang is not None:
        def rot(x):
            x2 = x.reshape(*x.shape[:-1], 16, 2)
            half = torch.stack((-x2[..., 1], x2[..., 0]), dim=-1).reshape(x.shape)
            return x * ang.cos() + half * ang.sin()
        q, k = rot(q), rot(k)
    sim = torch.einsum("...id,...jd->...ij", q, k)
    if bias is not None:
        sim = sim + bias
    return torch.einsum("...ij,...jd->...id", sim.softmax(-1), v)


@pytest.mark.parametrize("Fr,H,W,B", [(32, 4, 8, 2), (8, 4, 4, 3), (32, 3, 8, 1), (32, 3, 5, 2), (32, 16, 16, 3)])
def test_temporal_attention_core_backward(Fr, H, W, B):
    """'b c f h w -> b (h w) f c' attention with rotary + relative-position bias (conv3d.py:277-353): sdc_attn / sdc_attn_bwd
    on the channel-major qkv against fp64 autograd"""
    hw = H * W
    qkv = det_tensor((B, 384, Fr, H, W), 140)
    bias = 0.5 * det_tensor((4, Fr, Fr), 141)
    gout = det_tensor((B, 128, Fr, H, W), 142)
    rot, ang = _rot_table(Fr)
    qd, bd = qkv.double().requires_grad_(), bias.double().requires_grad_()
    q, k, v = (t.reshape(B, 4, 32, Fr, hw).permute(0, 4, 1, 3, 2) for t in qd.chunk(3, dim=1))       # (B, hw, heads, F, 32)
    o = _ref_attention(q, k, v, ang, bd)                                                              # (B, hw, heads, F, 32)
    out_ref = o.permute(0, 2, 4, 3, 1).reshape(B, 128, Fr, H, W)
    out_ref.backward(gout.double())
    qs, os_ = (384 * Fr * hw, Fr * hw, 1, hw), (128 * Fr * hw, Fr * hw, 1, hw)
    qg = qkv.to(DEV)
    out = torch.empty((B, 128, Fr, H, W), device=DEV)
    grad_ops.attn_core(qg, out, 4, B, hw, Fr, qs, os_, rot.to(DEV), bias.to(DEV))
    assert _rel(out, out_ref.detach()) < 1e-5
    dqkv, dbias = grad_ops.attn_core_bwd(qg, gout.to(DEV), 4, B, hw, Fr, qs, os_, rot.to(DEV), bias.to(DEV))
    e1, e2 = _rel(dqkv, qd.grad), _rel(dbias, bd.grad)
    print(f"[measured] temporal attention core backward F={Fr} {H}x{W} B={B}: dqkv {e1:.2e} dbias {e2:.2e}")
    assert e1 < 2e-5 and e2 < 2e-5


@pytest.mark.parametrize("B,inner,n", [(3, 1, 32), (2, 4, 256), (5, 1, 16), (1, 2, 100)])
def test_full_attention_core_backward(B, inner, n):
    """token-contiguous softmax attention (mid_attn of the 1-D nets: inner = 1; smoke mid spatial attention: inner = frames)"""
    qkv = det_tensor((B, 384, inner, n), 150)
    gout = det_tensor((B, 128, inner, n), 151)
    qd = qkv.double().requires_grad_()
    q, k, v = (t.reshape(B, 4, 32, inner, n).permute(0, 3, 1, 4, 2) for t in qd.chunk(3, dim=1))      # (B, inner, heads, n, 32)
    out_ref = _ref_attention(q, k, v).permute(0, 2, 4, 1, 3).reshape(B, 128, inner, n)
    out_ref.backward(gout.double())
    qs, os_ = (384 * inner * n, inner * n, n, 1), (128 * inner * n, inner * n, n, 1)
    qg = qkv.to(DEV)
    out = torch.empty((B, 128, inner, n), device=DEV)
    grad_ops.attn_core(qg, out, 4, B, inner, n, qs, os_)
    assert _rel(out, out_ref.detach()) < 1e-5
    dqkv, _ = grad_ops.attn_core_bwd(qg, gout.to(DEV), 4, B, inner, n, qs, os_)
    err = _rel(dqkv, qd.grad)
    print(f"[measured] softmax attention core backward B={B} inner={inner} n={n}: dqkv {err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("B,inner,n", [(2, 1, 2048), (3, 4, 256), (2, 1, 77), (4, 1, 16)])
def test_linear_attention_core_backward(B, inner, n):
    """LinearAttention core (1D/model/unet.py:203-216, conv3d.py:246-256): softmax over d of q, over n of k, context, output"""
    qkv = det_tensor((B, 384, inner, n), 160)
    gout = det_tensor((B, 128, inner, n), 161)
    qd = qkv.double().requires_grad_()
    q, k, v = (t.reshape(B, 4, 32, inner, n).permute(0, 3, 1, 2, 4) for t in qd.chunk(3, dim=1))      # (B, inner, heads, 32, n)
    qq = q.softmax(dim=-2) * 32 ** -0.5
    kk = k.softmax(dim=-1)
    ctx = torch.einsum("...dn,...en->...de", kk, v)
    out_ref = torch.einsum("...de,...dn->...en", ctx, qq).permute(0, 2, 3, 1, 4).reshape(B, 128, inner, n)
    out_ref.backward(gout.double())
    qs, os_ = (384 * inner * n, inner * n, n), (128 * inner * n, inner * n, n)
    qg = qkv.to(DEV)
    out = torch.empty((B, 128, inner, n), device=DEV)
    grad_ops.linattn_core(qg, out, 4, B, inner, n, qs, os_)
    assert _rel(out, out_ref.detach()) < 1e-5
    dqkv = grad_ops.linattn_core_bwd(qg, gout.to(DEV), 4, B, inner, n, qs, os_)
    err = _rel(dqkv, qd.grad)
    print(f"[measured] linear attention core backward B={B} inner={inner} n={n}: dqkv {err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("shape,prec", [((64, 48, 3, 3, 3), 4), ((40, 64, 1, 3, 3), 4), ((96, 32, 3), 4), ((64, 7, 7, 7, 7), 4),
                                        ((128, 64, 1, 1), 4), ((32, 16, 3, 3, 3), 3), ((32, 16, 3, 3), 2), ((8, 8, 3, 3, 3), 0),
                                        ((7, 5, 3, 3, 3), 4), ((70, 33, 3, 3), 4), ((2048, 24, 3), 4)])
def test_pack_conv_weight_kernel_equals_host_packing(shape, prec):
    """sdc_pack_conv_weight (one launch) against engine.pack_conv_weight (torch, fp64 einsums): Wp and every Winograd section,
    forward and data-gradient (transposed, flipped) forms"""
    from safediffcon_amd.engine import as5, pack_conv_weight
    w = det_tensor(shape, 180, 0.3).to(DEV)
    got, want = grad_ops.pack_conv_weight(w, prec), pack_conv_weight(w, "conv", prec).reshape(-1)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-7 * want.abs().max().item()
    w5 = as5(w)
    wt = w5.transpose(0, 1).flip(2, 3, 4).contiguous()
    got, want = grad_ops.pack_conv_weight(w, prec, flip=True), pack_conv_weight(wt, "conv", prec).reshape(-1)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-7 * want.abs().max().item()


def test_conv_node_gradients_seeded_sweep():
    """autograd.ConvFn (forward sdc_conv, data gradient sdc_conv with flipped taps, weight / bias gradient sdc_conv_wgrad) over a
    seeded sweep of the conv forms the three nets hold -- 1-D / 2-D / 3-D, 1 / 3 / 7 taps, stride-1 'same' convs with ragged
    channel counts and row lengths, two concatenated inputs, the strided / transposed / unshuffle / upsample forms -- against
    torch autograd in fp64"""
    import random
    from safediffcon_amd.autograd import ConvFn
    from safediffcon_amd.engine import as5
    rng = random.Random(1234)
    worst = {}
    for case in range(36):
        nd = rng.choice([1, 2, 3])
        k = rng.choice([1, 3, 3, 3, 7]) if nd < 3 else rng.choice([1, 3, 3])
        cin, cout = rng.choice([3, 7, 8, 12, 16, 24, 40, 64, 96]), rng.choice([7, 8, 16, 32, 40, 64, 72, 128])
        cin1 = rng.choice([0, 0, 8, 16]) if k == 3 else 0
        B = rng.choice([1, 2, 3])
        sp = tuple(rng.choice([4, 6, 8, 16, 20, 32, 64] if i == nd - 1 else [1, 2, 4, 6, 8]) for i in range(nd))
        x = det_tensor((B, cin, *sp), 2000 + case, 0.5)
        x1 = det_tensor((B, cin1, *sp), 2100 + case, 0.5) if cin1 else None
        w = det_tensor((cout, cin + cin1, *([k] * nd)), 2200 + case, 0.2)
        b = det_tensor((cout,), 2300 + case, 0.1)
        xd = x.double().requires_grad_()
        x1d = x1.double().requires_grad_() if cin1 else None
        wd, bd = w.double().requires_grad_(), b.double().requires_grad_()
        xin = xd if x1d is None else torch.cat((xd, x1d), 1)
        ref = (F.conv1d, F.conv2d, F.conv3d)[nd - 1](xin, wd, bd, padding=k // 2)
        gy = det_tensor(tuple(ref.shape), 2400 + case)
        ref.backward(gy.double())
        xg = as5(x.to(DEV)).requires_grad_()
        x1g = as5(x1.to(DEV)).requires_grad_() if cin1 else None
        wg, bg = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        k3 = (1,) * (3 - nd) + (k,) * nd
        y = ConvFn.apply(xg, x1g, wg, bg, ("conv", (1, 1, 1), tuple(kk // 2 for kk in k3), (1, 1, 1), 4))
        y.backward(as5(gy.to(DEV)))
        errs = dict(y=_rel(y.detach().reshape(ref.shape), ref.detach()), gx=_rel(xg.grad.reshape(x.shape), xd.grad),
                    gw=_rel(wg.grad, wd.grad), gb=_rel(bg.grad, bd.grad))
        if cin1:
            errs["gx1"] = _rel(x1g.grad.reshape(x1.shape), x1d.grad)
        for kk, v in errs.items():
            worst[kk] = max(worst.get(kk, 0.0), v)
            assert v < 3e-5, (case, nd, k, cin, cin1, cout, B, sp, kk, v)
    print(f"[measured] conv node sweep (36 shapes): worst relative errors {worst}")
    # the strided / transposed / resampling forms
    forms = [
        ("conv", (2, 64, 2, 16, 16), (32, 64, 1, 4, 4), dict(stride=(1, 2, 2), pad=(0, 1, 1)),
         lambda x_, w_, b_: F.conv3d(x_, w_, b_, stride=(1, 2, 2), padding=(0, 1, 1))),
        ("convT422", (2, 48, 2, 8, 8), (48, 24, 1, 4, 4), dict(),
         lambda x_, w_, b_: F.conv_transpose3d(x_, w_, b_, stride=(1, 2, 2), padding=(0, 1, 1))),
        ("unshuffle", (2, 16, 1, 8, 32), (40, 64, 1, 1), dict(),
         lambda x_, w_, b_: F.conv2d(x_[:, :, 0].reshape(2, 16, 4, 2, 16, 2).permute(0, 1, 3, 5, 2, 4).reshape(2, 64, 4, 16), w_, b_).unsqueeze(2)),
        ("conv", (3, 24, 1, 1, 32), (16, 24, 4), dict(stride=(1, 1, 2), pad=(0, 0, 1)),
         lambda x_, w_, b_: F.conv1d(x_[:, :, 0, 0], w_, b_, stride=2, padding=1).unsqueeze(2).unsqueeze(2)),
        ("conv", (2, 32, 1, 4, 16), (24, 32, 3, 3), dict(up=(1, 2, 2), pad=(0, 1, 1)),
         lambda x_, w_, b_: F.conv2d(F.interpolate(x_[:, :, 0], scale_factor=2, mode="nearest"), w_, b_, padding=1).unsqueeze(2)),
        ("conv", (2, 40, 1, 1, 16), (24, 40, 3), dict(up=(1, 1, 2), pad=(0, 0, 1)),
         lambda x_, w_, b_: F.conv1d(F.interpolate(x_[:, :, 0, 0], scale_factor=2, mode="nearest"), w_, b_, padding=1).unsqueeze(2).unsqueeze(2)),
    ]
    for i, (kind, xs, ws, kw, fn) in enumerate(forms):
        x, w = det_tensor(xs, 2500 + i, 0.5), det_tensor(ws, 2600 + i, 0.2)
        nb = ws[1] if kind == "convT422" else ws[0]
        b = det_tensor((nb,), 2700 + i, 0.1)
        xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
        ref = fn(xd, wd, bd)
        gy = det_tensor(tuple(ref.shape), 2800 + i)
        ref.backward(gy.double())
        xg, wg, bg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        w5 = wg if kind == "convT422" else wg
        cfg = (kind, kw.get("stride", (1, 1, 1)), kw.get("pad", (0, 0, 0)), kw.get("up", (1, 1, 1)), 4)
        y = ConvFn.apply(xg, None, w5, bg, cfg)
        assert _rel(y.detach(), ref.detach()) < 3e-5, (kind, i)
        y.backward(gy.to(DEV))
        assert _rel(xg.grad, xd.grad) < 3e-5 and _rel(wg.grad, wd.grad) < 3e-5 and _rel(bg.grad, bd.grad) < 3e-5, (kind, i)
