"""-m gpu: every libsdc_hip.so stage against the CPU oracle / plain torch fp32 ops on the same seeded
inputs (called through the C ABI via the engine), including the edge cases the path has: ragged tile
edges, Cin not a multiple of the K chunk, two-input concat, stride/upsample/transposed gathers,
strided (frame-major) tensors, hinge on/off, empty-ish extents."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle import nets as onets
from oracle import samplers as osam
from oracle.detweights import det_tensor

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
# fp32 MFMA is an exact fp32 FMA chain; the only differences vs the CPU reference are summation order
TOL = dict(rtol=2e-4, atol=2e-5)


@pytest.fixture(scope="module")
def plan_cls():
    from safediffcon_amd.engine import Plan
    return Plan


def _stream():
    return torch.cuda.current_stream(torch.device(DEV)).cuda_stream


def _run(plan):
    plan.run(_stream())
    torch.cuda.synchronize()


def _conv_case(plan_cls, nd, B, cin, cout, sp, k, stride=1, pad=0, up=1, cin1=0, residual=False, seed=0):
    from safediffcon_amd.engine import as5
    x = det_tensor((B, cin, *sp), seed + 1)
    x1 = det_tensor((B, cin1, *sp), seed + 2) if cin1 else None
    w = det_tensor((cout, cin + cin1, *([k] * nd)), seed + 3, 0.2)
    b = det_tensor((cout,), seed + 4, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    if up > 1:
        xin = F.interpolate(xin, scale_factor=up, mode="nearest")
    ref = (F.conv1d, F.conv2d, F.conv3d)[nd - 1](xin, w, b, stride=stride, padding=pad)
    res = det_tensor(tuple(ref.shape), seed + 5) if residual else None
    if residual:
        ref = ref + res
    plan = plan_cls(DEV)
    k3 = (1,) * (3 - nd) + (k,) * nd
    s3 = (1,) * (3 - nd) + (stride,) * nd
    p3 = (0,) * (3 - nd) + (pad,) * nd
    u3 = (1,) * (3 - nd) + (up,) * nd
    wp = plan.conv_weight(w.to(DEV))
    out = plan.conv(as5(x.to(DEV)), wp, b.to(DEV), cout, k3, x1=None if x1 is None else as5(x1.to(DEV)), stride=s3,
                    pad=p3, up=u3, residual=None if res is None else as5(res.to(DEV)))
    _run(plan)
    # summation-order noise scales with the output magnitude (K up to 3456 products per output)
    torch.testing.assert_close(out.cpu().reshape(ref.shape), ref, rtol=2e-4, atol=1e-5 * ref.abs().max().item())


@pytest.mark.parametrize("case", [
    dict(nd=1, B=3, cin=12, cout=8, sp=(128,), k=7, pad=3),                 # tokamak init conv, Cin % 16 != 0
    dict(nd=1, B=2, cin=32, cout=64, sp=(64,), k=4, stride=2, pad=1),       # Downsample k4 s2
    dict(nd=1, B=2, cin=32, cout=16, sp=(32,), k=3, pad=1, up=2),           # Upsample + conv
    dict(nd=2, B=2, cin=3, cout=8, sp=(16, 128), k=7, pad=3),               # burgers init conv
    dict(nd=2, B=2, cin=64, cout=64, sp=(16, 128), k=3, pad=1),             # fast path, 64x128 tile
    dict(nd=2, B=5, cin=48, cout=130, sp=(8, 20), k=3, pad=1, cin1=16, residual=True),   # ragged M and N, concat
    dict(nd=2, B=2, cin=16, cout=3, sp=(16, 128), k=1),                     # final 1x1, Cout=3
    dict(nd=2, B=2, cin=32, cout=16, sp=(4, 32), k=3, pad=1, up=2),         # Upsample2d
    dict(nd=3, B=1, cin=7, cout=8, sp=(8, 16, 16), k=7, pad=3),             # smoke init conv
    dict(nd=3, B=2, cin=16, cout=32, sp=(4, 8, 8), k=3, pad=1, cin1=16),    # Conv3d 3^3 with skip concat
    dict(nd=3, B=36, cin=128, cout=160, sp=(4, 16, 16), k=3, pad=1),        # 128x128 tile path (Ntot >= 32768), row-halo
    dict(nd=2, B=3, cin=3, cout=64, sp=(16, 128), k=7, pad=3),              # stem conv path (generalized k rows, KW=7)
    dict(nd=3, B=2, cin=7, cout=64, sp=(8, 32, 32), k=7, pad=3),            # smoke stem: 343 (kd,kh,ci) rows, ragged last stage
    dict(nd=1, B=5, cin=12, cout=64, sp=(128,), k=7, pad=3),                # tokamak stem
    dict(nd=2, B=130, cin=64, cout=64, sp=(16, 128), k=3, pad=1),           # 64x256 row-halo tile, ragged last tile rows
    dict(nd=3, B=2, cin=32, cout=64, sp=(6, 16, 32), k=3, pad=1, cin1=32, residual=True),   # row-halo with concat + residual
])
def test_conv_matches_torch(plan_cls, case):
    _conv_case(plan_cls, **case)


def test_conv_special_gathers(plan_cls):
    from safediffcon_amd.engine import as5
    # Downsample2d: pixel-unshuffle + 1x1  (1D/model/unet.py:39-43)
    x = det_tensor((2, 16, 8, 32), 10)
    w = det_tensor((24, 64, 1, 1), 11, 0.2)
    b = det_tensor((24,), 12, 0.1)
    xs = x.reshape(2, 16, 4, 2, 16, 2).permute(0, 1, 3, 5, 2, 4).reshape(2, 64, 4, 16)
    ref = F.conv2d(xs, w, b)
    plan = plan_cls(DEV)
    out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV), "unshuffle"), b.to(DEV), 24, (1, 2, 2), stride=(1, 2, 2))
    # ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1)  (conv3d.py:159-160)
    xt = det_tensor((2, 16, 3, 8, 8), 13)
    wt = det_tensor((16, 16, 1, 4, 4), 14, 0.2)
    bt = det_tensor((16,), 15, 0.1)
    reft = F.conv_transpose3d(xt, wt, bt, stride=(1, 2, 2), padding=(0, 1, 1))
    outt = plan.conv(xt.to(DEV), plan.conv_weight(wt.to(DEV), "convT"), bt.to(DEV), 16, (1, 4, 4), up=(1, 2, 2),
                     up_mode=1, pad=(0, 2, 2))
    # Conv3d (1,4,4) stride (1,2,2) pad (0,1,1)
    wd = det_tensor((16, 16, 1, 4, 4), 16, 0.2)
    refd = F.conv3d(xt, wd, bt, stride=(1, 2, 2), padding=(0, 1, 1))
    outd = plan.conv(xt.to(DEV), plan.conv_weight(wd.to(DEV)), bt.to(DEV), 16, (1, 4, 4), stride=(1, 2, 2), pad=(0, 1, 1))
    # strided input/output views: frame-major (B,F,C,H,W) storage read and written through strides
    xf = det_tensor((2, 4, 7, 8, 8), 17)
    wf = det_tensor((7, 7, 3, 3, 3), 18, 0.2)
    reff = F.conv3d(xf.permute(0, 2, 1, 3, 4), wf, None, padding=1).permute(0, 2, 1, 3, 4)
    outf_store = torch.zeros(2, 4, 7, 8, 8, device=DEV)
    plan.conv(xf.to(DEV).permute(0, 2, 1, 3, 4), plan.conv_weight(wf.to(DEV)), None, 7, (3, 3, 3), pad=(1, 1, 1),
              out=outf_store.permute(0, 2, 1, 3, 4))
    _run(plan)
    torch.testing.assert_close(out.cpu().reshape(ref.shape), ref, **TOL)
    torch.testing.assert_close(outt.cpu(), reft, **TOL)
    torch.testing.assert_close(outd.cpu(), refd, **TOL)
    torch.testing.assert_close(outf_store.cpu(), reff, **TOL)


def test_conv_rejects_bad_shapes(plan_cls):
    from safediffcon_amd._lib import SdcConvDesc, get_lib, last_error
    d = SdcConvDesc()
    d.B, d.Cin0, d.Cout = 1, 4, 4
    d.iD = d.iH = d.iW = 4
    d.oD = d.oH = d.oW = 5          # inconsistent
    d.kD = d.kH = d.kW = 3
    d.sD = d.sH = d.sW = 1
    d.uD = d.uH = d.uW = 1
    t = torch.zeros(256, device=DEV)
    rc = get_lib().sdc_conv(C.byref(d), t.data_ptr(), 0, t.data_ptr(), 0, 0, t.data_ptr(), _stream())
    assert rc == -1 and "inconsistent" in last_error()


@pytest.mark.parametrize("B,Cc,G,sp", [(3, 8, 1, (16, 128)), (2, 64, 8, (4, 8, 8)), (2, 16, 8, (6, 5, 3)), (2, 32, 1, (16,)),
                                       (1, 8, 8, (32, 64, 64))])
def test_groupnorm_silu(plan_cls, B, Cc, G, sp):
    from safediffcon_amd.engine import as5
    x = det_tensor((B, Cc, *sp), 20) * 3 + 1.5
    gamma, beta = det_tensor((Cc,), 21) * 0.2 + 1, det_tensor((Cc,), 22) * 0.2
    ss = det_tensor((B, 2 * Cc + 5), 23, 0.3)
    res = det_tensor((B, Cc, *sp), 24)
    y = F.group_norm(x, G, gamma, beta, eps=1e-5)
    sc = ss[:, 5:5 + Cc].reshape(B, Cc, *([1] * len(sp)))
    sh = ss[:, 5 + Cc:5 + 2 * Cc].reshape(B, Cc, *([1] * len(sp)))
    ref = F.silu(y * (sc + 1) + sh) + res
    plan = plan_cls(DEV)
    xd = as5(x.to(DEV)).contiguous()
    out = plan.gn_silu(xd, gamma.to(DEV), beta.to(DEV), G, ss=ss.to(DEV), ss_b_stride=ss.shape[1], ss_off=5,
                       residual=as5(res.to(DEV)).contiguous())
    _run(plan)
    torch.testing.assert_close(out.cpu().reshape(ref.shape), ref, **TOL)
    # unconditioned, no residual, device-side t indexing
    plan = plan_cls(DEV)
    xd = as5(x.to(DEV)).contiguous()
    t_dev = torch.tensor([2], dtype=torch.int32, device=DEV)
    lut = torch.zeros(4, 2 * Cc, device=DEV)
    lut[2] = ss[0, 5:5 + 2 * Cc].to(DEV)
    out = plan.gn_silu(xd, gamma.to(DEV), beta.to(DEV), G, ss=lut, t_dev=t_dev, ss_t_stride=2 * Cc)
    _run(plan)
    sc0 = ss[0, 5:5 + Cc].reshape(1, Cc, *([1] * len(sp)))
    sh0 = ss[0, 5 + Cc:5 + 2 * Cc].reshape(1, Cc, *([1] * len(sp)))
    torch.testing.assert_close(out.cpu().reshape(ref.shape), F.silu(y * (sc0 + 1) + sh0), **TOL)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("B,Cc,S", [(2, 8, 2048), (3, 64, 100), (2, 2048, 16), (1, 5, 3), (3, 256, 128), (2, 1024, 20), (2, 512, 32),
                                    (2, 4096, 16), (2, 448, 70)])
def test_channel_norms(plan_cls, mode, B, Cc, S):
    x = det_tensor((B, Cc, S), 30) * 2 + 0.7
    g = det_tensor((1, Cc, 1), 31) * 0.3 + 1
    res = det_tensor((B, Cc, S), 32)
    ref = (onets.chan_layernorm(x, g) if mode == 0 else onets.chan_rmsnorm(x, g)) + res
    plan = plan_cls(DEV)
    out = plan.chan_norm(x.to(DEV).reshape(B, Cc, 1, 1, S), g.to(DEV).reshape(-1), mode, residual=res.to(DEV))
    _run(plan)
    torch.testing.assert_close(out.cpu().reshape(ref.shape), ref, **TOL)


@pytest.mark.parametrize("kind", [0, 1])
def test_activations(plan_cls, kind):
    x = det_tensor((1000,), 40) * 3
    plan = plan_cls(DEV)
    xd = x.to(DEV)
    plan.act(xd, kind)
    _run(plan)
    torch.testing.assert_close(xd.cpu(), F.silu(x) if kind == 0 else F.gelu(x), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,inner,n", [(2, 1, 2048), (2, 3, 256), (1, 2, 100), (3, 1, 16)])
def test_linear_attention_core(plan_cls, B, inner, n):
    heads = 4
    qkv = det_tensor((B, 3 * 128, inner, n), 50) * 1.5
    q, k, v = (t.permute(0, 2, 1, 3).reshape(B * inner, heads, 32, n) for t in qkv.chunk(3, dim=1))
    qs = q.softmax(dim=-2) * 32 ** -0.5
    ks = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", ks, v)
    ref = torch.einsum("bhde,bhdn->bhen", ctx, qs).reshape(B, inner, 128, n).permute(0, 2, 1, 3)
    plan = plan_cls(DEV)
    out = torch.zeros(B, 128, inner, n, device=DEV)
    plan.linattn(qkv.to(DEV), heads, B, inner, n, (384 * inner * n, inner * n, n), out, (128 * inner * n, inner * n, n))
    _run(plan)
    torch.testing.assert_close(out.cpu(), ref, **TOL)


@pytest.mark.parametrize("contig,B,inner,ntok,rot_bias", [
    (True, 3, 1, 32, False), (True, 2, 1, 16, False), (True, 2, 4, 256, False), (True, 5, 3, 20, False),
    (False, 2, 64, 32, True), (False, 1, 12, 8, True), (False, 2, 7, 32, True),
    (False, 3, 24, 32, True),          # 8 pixels per workgroup (inner % 16 != 0)
    (False, 2, 4096, 32, True)])       # 16 pixels per workgroup, two pipelined tiles per workgroup
def test_softmax_attention_core(plan_cls, contig, B, inner, ntok, rot_bias):
    heads = 4
    if contig:     # (B, 384, inner, ntok): tokens contiguous
        qkv = det_tensor((B, 384, inner, ntok), 60) * 1.2
        seqs = qkv.permute(0, 2, 3, 1)                      # (B, inner, tok, c)
    else:          # (B, 384, ntok, inner): tokens strided (temporal attention)
        qkv = det_tensor((B, 384, ntok, inner), 61) * 1.2
        seqs = qkv.permute(0, 3, 2, 1)
    q, k, v = (t.reshape(B, inner, ntok, heads, 32).transpose(-2, -3) for t in seqs.chunk(3, dim=-1))
    q = q * 32 ** -0.5
    rot = bias = None
    if rot_bias:
        freqs = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
        q, k = onets.rotary(q, freqs), onets.rotary(k, freqs)
        bias = det_tensor((heads, ntok, ntok), 62, 0.5)
        ang = torch.arange(ntok, dtype=torch.float32)[:, None] * freqs[None, :]
        rot = torch.stack((ang.cos(), ang.sin()), dim=-1).reshape(-1).to(DEV)
    sim = torch.einsum("...hid,...hjd->...hij", q, k)
    if bias is not None:
        sim = sim + bias
    o = torch.einsum("...hij,...hjd->...hid", sim.softmax(dim=-1), v)      # (B, inner, h, tok, d)
    o = o.transpose(-2, -3).reshape(B, inner, ntok, 128)
    plan = plan_cls(DEV)
    if contig:
        ref = o.permute(0, 3, 1, 2)
        out = torch.zeros(B, 128, inner, ntok, device=DEV)
        plan.attn(qkv.to(DEV), out, heads, B, inner, ntok, (384 * inner * ntok, inner * ntok, ntok, 1),
                  (128 * inner * ntok, inner * ntok, ntok, 1))
    else:
        ref = o.permute(0, 3, 2, 1)
        out = torch.zeros(B, 128, ntok, inner, device=DEV)
        plan.attn(qkv.to(DEV), out, heads, B, inner, ntok, (384 * inner * ntok, inner * ntok, 1, inner),
                  (128 * inner * ntok, inner * ntok, 1, inner), rot, None if bias is None else bias.to(DEV))
    _run(plan)
    torch.testing.assert_close(out.cpu(), ref, **TOL)


def test_philox_normal_statistics():
    from safediffcon_amd._lib import get_lib, check
    n = 1 << 22
    x = torch.empty(n, device=DEV)
    draw = torch.zeros(1, dtype=torch.int32, device=DEV)
    check(get_lib().sdc_randn(x.data_ptr(), n, 1234, draw.data_ptr(), _stream()))
    a = x.clone()
    draw.fill_(1)
    check(get_lib().sdc_randn(x.data_ptr(), n, 1234, draw.data_ptr(), _stream()))
    torch.cuda.synchronize()
    assert abs(a.mean().item()) < 3e-3 and abs(a.std().item() - 1) < 3e-3
    assert abs((a ** 4).mean().item() - 3) < 0.05          # kurtosis of N(0,1)
    assert abs((a * x).mean().item()) < 3e-3               # draws are independent
    assert torch.isfinite(a).all()


def test_conformal_scores_and_weights():
    from safediffcon_amd import conformal
    pred, state = det_tensor((6, 3, 16, 128), 70, 0.1), det_tensor((6, 3, 16, 128), 71, 0.1)
    state[:3, 2] += 0.07
    for ums in (True, False):
        s, w = conformal.scores_and_weights("burgers", pred.to(DEV), state.to(DEV), [500.0, 0.8 ** 2, 0.01, 10.0], use_max=not ums)
        torch.testing.assert_close(s.cpu(), osam.burgers_score(pred, state, ums), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(w.cpu(), osam.burgers_weight(state, 0.01, 500.0, 0.8, ums), rtol=1e-3, atol=1e-30)
    pred, state = det_tensor((4, 12, 128), 72, 0.3) + 0.5, det_tensor((4, 12, 128), 73, 0.3) + 0.5
    target = det_tensor((4, 3, 122), 74, 0.3) + 1.0
    s, w = conformal.scores_and_weights("tokamak", pred.to(DEV), state.to(DEV), [0.7, 0.3, 0.5, 3.6, 0.1], target=target.to(DEV))
    torch.testing.assert_close(s.cpu(), osam.tokamak_score(pred, state, 122), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(w.cpu(), osam.tokamak_weight(state, target, 122, 0.1, 3.6, 0.7, 0.3, 0.5), rtol=1e-4, atol=0)
    pred, state = det_tensor((4, 8, 7, 16, 16), 75, 0.3), det_tensor((4, 8, 7, 16, 16), 76, 0.3)
    state[:2, -1, 6] += 0.2
    s, w = conformal.scores_and_weights("smoke", pred.to(DEV), state.to(DEV), [0.9, 0.1, 0.01, 100.0])
    torch.testing.assert_close(s.cpu(), osam.smoke_score(pred, state), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(w.cpu(), osam.smoke_weight(state, 0.01, 0.9, 0.1, 100.0), rtol=1e-3, atol=0)


@pytest.mark.parametrize("case", [
    dict(nd=2, B=4, cin=64, cout=64, sp=(16, 128), k=3, pad=1),                                 # whole rows per tile
    dict(nd=2, B=40, cin=128, cout=192, sp=(8, 64), k=3, pad=1, cin1=64, residual=True),        # 128-row tile, concat, ragged Cout tile
    dict(nd=2, B=6, cin=32, cout=128, sp=(2, 16), k=3, pad=1),                                  # 16-wide rows: 8 segments per tile
    dict(nd=2, B=300, cin=64, cout=64, sp=(16, 128), k=3, pad=1, residual=True),                # 64x512 tile
    dict(nd=2, B=40, cin=32, cout=48, sp=(16, 128), k=3, pad=1),                                # 64x256 tile, ragged Cout
    dict(nd=2, B=64, cin=32, cout=128, sp=(16, 64), k=3, pad=1, residual=True),                 # 128x256 tile
    dict(nd=2, B=260, cin=16, cout=160, sp=(16, 16), k=3, pad=1),                               # 128x256 tile, 16-wide rows, ragged
    dict(nd=2, B=1, cin=16, cout=36, sp=(5, 256), k=3, pad=1),                                  # rows longer than a tile, ragged N
    dict(nd=3, B=2, cin=32, cout=96, sp=(4, 16, 16), k=3, pad=1),                               # 3x3x3
    dict(nd=1, B=3, cin=48, cout=64, sp=(128,), k=3, pad=1),                                    # Conv1d k3
    dict(nd=2, B=2, cin=64, cout=384, sp=(16, 128), k=1),                                       # not covered -> direct kernel
    dict(nd=2, B=2, cin=24, cout=64, sp=(8, 30), k=3, pad=1),                                   # Cin % 16 != 0 -> direct kernel
])
def test_conv_winograd_mode(plan_cls, case):
    """precision=2: fp32 Winograd F(2,3) along W.  Same fp32 arithmetic in a different summation order: error stays
    within a few fp32 ulps of the output scale (direct kernel: <= 2e-6, Winograd: <= 6e-6)."""
    from safediffcon_amd.engine import as5
    nd, B, cin, cout, sp, k = case["nd"], case["B"], case["cin"], case["cout"], case["sp"], case["k"]
    pad, cin1 = case.get("pad", 0), case.get("cin1", 0)
    x, x1 = det_tensor((B, cin, *sp), 91), (det_tensor((B, cin1, *sp), 92) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, *([k] * nd)), 93, 0.2), det_tensor((cout,), 94, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = (F.conv1d, F.conv2d, F.conv3d)[nd - 1](xin.double(), w.double(), b.double(), padding=pad)
    res = det_tensor(tuple(ref.shape), 95) if case.get("residual") else None
    if res is not None:
        ref = ref + res.double()
    outs = {}
    for prec in (0, 2):
        plan = plan_cls(DEV, precision=prec)
        k3, p3 = (1,) * (3 - nd) + (k,) * nd, (0,) * (3 - nd) + (pad,) * nd
        out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, k3,
                        x1=None if x1 is None else as5(x1.to(DEV)), pad=p3,
                        residual=None if res is None else as5(res.to(DEV)))
        _run(plan)
        outs[prec] = out.cpu().reshape(ref.shape).double()
    scale = ref.abs().max().item()
    e0 = (outs[0] - ref).abs().max().item() / scale
    e2 = (outs[2] - ref).abs().max().item() / scale
    print(f"rel err direct {e0:.2e} winograd {e2:.2e}")
    assert e0 < 2e-6 and e2 < 6e-6, (e0, e2)
    covered = k == 3 and cin % 16 == 0 and cin1 % 16 == 0 and cout > 32
    assert torch.equal(outs[0], outs[2]) != covered       # Winograd where eligible, the direct kernel elsewhere


@pytest.mark.parametrize("case", [
    dict(nd=2, B=5, cin=64, cout=64, sp=(16, 128), G=1),                      # Burgers top level: one row pair per workgroup
    dict(nd=2, B=3, cin=32, cout=128, sp=(8, 64), cin1=32, G=8),              # two inputs (skip concat), two m-tiles
    dict(nd=2, B=7, cin=16, cout=96, sp=(4, 32), residual=True, G=0),         # ragged Cout (96 = 64 + 32), 4 row pairs per tile, partial last tile
    dict(nd=2, B=3, cin=24, cout=64, sp=(6, 16), G=0),                        # 16-wide rows: 8 row pairs per tile, 3 pairs per image, Cin % 8 == 0 only
    dict(nd=3, B=2, cin=64, cout=64, sp=(5, 8, 64), residual=True, G=8),      # 3x3x3: depth padding at both ends
    dict(nd=3, B=1, cin=40, cout=256, sp=(3, 16, 32), cin1=24, G=8),          # 3x3x3, concat, 4 m-tiles
    dict(nd=3, B=2, cin=32, cout=64, sp=(2, 2, 16), G=0),                     # one row pair per image
])
def test_conv_winograd_2d_mode(plan_cls, case):
    """precision=3: fp32 Winograd F(2x2,3x3) over (H, W) (3x3 and 3x3x3 stride-1 convs over whole rows), against torch in
    fp64 and beside the direct kernel (precision 0): still fp32 end to end, the error stays within a few fp32 ulps of the
    output scale (gate 1e-5 of the output scale).  With G > 0 the GroupNorm statistics come out of its epilogue."""
    from safediffcon_amd.engine import as5
    nd, B, cin, cout, sp = case["nd"], case["B"], case["cin"], case["cout"], case["sp"]
    cin1, G = case.get("cin1", 0), case["G"]
    x, x1 = det_tensor((B, cin, *sp), 191), (det_tensor((B, cin1, *sp), 192) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, *([3] * nd)), 193, 0.2), det_tensor((cout,), 194, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = (F.conv2d, F.conv3d)[nd - 2](xin.double(), w.double(), b.double(), padding=1)
    res = det_tensor(tuple(ref.shape), 195) if case.get("residual") else None
    if res is not None:
        ref = ref + res.double()
    outs, names = {}, {}
    for prec in (0, 3):
        plan = plan_cls(DEV, precision=prec)
        k3, p3 = (1,) * (3 - nd) + (3,) * nd, (0,) * (3 - nd) + (1,) * nd
        out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, k3,
                        x1=None if x1 is None else as5(x1.to(DEV)), pad=p3,
                        residual=None if res is None else as5(res.to(DEV)), gn_groups=G if prec == 3 else 0)
        d = plan.calls[0][1][0]._obj
        buf = C.create_string_buffer(128)
        assert plan.lib.sdc_conv_describe(C.byref(d), buf, 128, None) == 0
        names[prec] = buf.value.decode()
        if prec == 3 and G:
            assert plan.calls[0][0] is plan.lib.sdc_conv_gn            # statistics fused into this kernel's epilogue
            gam, bet = det_tensor((cout,), 196, 0.3) + 1.0, det_tensor((cout,), 197, 0.2)
            y = plan.pool.get(tuple(out.shape))
            plan.gn_silu(out, gam.to(DEV), bet.to(DEV), G, out=y)
        _run(plan)
        outs[prec] = out.cpu().reshape(ref.shape).double()
        if prec == 3 and G:
            refn = F.silu(F.group_norm(ref, G, gam.double(), bet.double(), 1e-5))
            torch.testing.assert_close(y.cpu().reshape(ref.shape).double(), refn, rtol=1e-4, atol=2e-5)
    # (round 6: kD = 1, Cout % 64 == 0, rows of 128 / 64 / 32 run the two-workgroups-per-CU form conv_wg2s_kernel)
    assert names[3].startswith("conv_wg2") and not names[0].startswith("conv_wg")
    scale = ref.abs().max().item()
    e0 = (outs[0] - ref).abs().max().item() / scale
    e3 = (outs[3] - ref).abs().max().item() / scale
    print(f"[measured] rel err vs fp64: direct {e0:.2e}  winograd F(2x2,3x3) {e3:.2e}")
    assert e0 < 4e-6 and e3 < 1e-5, (e0, e3)


@pytest.mark.parametrize("case", [
    dict(B=2, cin=64, cout=64, sp=(4, 4, 64), G=8),                           # 64-wide rows: 2 row pairs per workgroup, both depth edges in every pair
    dict(B=1, cin=16, cout=128, sp=(6, 8, 32), cin1=24, G=8),                 # concat, 2 m-tiles, interior plane pair
    dict(B=3, cin=32, cout=64, sp=(2, 16, 16), G=0),                          # 16-wide rows (two rows interleaved per DPP row)
    dict(B=1, cin=8, cout=192, sp=(8, 8, 32), G=3),                           # one stage per depth component, 3 m-tiles, a group = one m-tile
    dict(B=1, cin=512, cout=128, sp=(2, 16, 16), G=1),                        # long K (64 stages per pass), one group over both m-tiles
    dict(B=2, cin=16, cout=64, sp=(2, 4, 64), cin1=8, G=8, x1_pad=3),         # single plane pair; second input with its own (padded) strides
])
def test_conv_winograd_3d_mode(plan_cls, case):
    """precision=4: fp32 Winograd F(2x2x2,3x3x3) over (D, H, W) against torch in fp64 and beside the direct kernel; with
    G > 0 the GroupNorm statistics come out of its last fold.  Gate 1e-5 of the output scale, like the other fp32 modes."""
    from safediffcon_amd.engine import as5
    B, cin, cout, sp = case["B"], case["cin"], case["cout"], case["sp"]
    cin1, G = case.get("cin1", 0), case["G"]
    x, x1 = det_tensor((B, cin, *sp), 291), (det_tensor((B, cin1, *sp), 292) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, 3, 3, 3), 293, 0.2), det_tensor((cout,), 294, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = F.conv3d(xin.double(), w.double(), b.double(), padding=1)
    res = det_tensor(tuple(ref.shape), 295) if case.get("residual") else None
    if res is not None:
        ref = ref + res.double()
    outs, names = {}, {}
    x1d = None
    if x1 is not None:
        x1d = x1.to(DEV)
        if case.get("x1_pad"):      # a channel slice of a wider buffer: batch stride != Cin1 * channel stride
            wide = torch.zeros(B, cin1 + case["x1_pad"], *sp, device=DEV)
            wide[:, :cin1] = x1d
            x1d = wide[:, :cin1]
    for prec in (0, 4):
        plan = plan_cls(DEV, precision=prec)
        out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (3, 3, 3),
                        x1=None if x1d is None else as5(x1d), pad=(1, 1, 1),
                        residual=None if res is None else as5(res.to(DEV)), gn_groups=G if prec == 4 else 0)
        d = plan.calls[0][1][0]._obj
        buf = C.create_string_buffer(128)
        share = C.c_double(0)
        assert plan.lib.sdc_conv_describe(C.byref(d), buf, 128, C.byref(share)) == 0
        names[prec] = buf.value.decode()
        if prec == 4 and G:
            assert plan.calls[0][0] is plan.lib.sdc_conv_gn
            gam, bet = det_tensor((cout,), 296, 0.3) + 1.0, det_tensor((cout,), 297, 0.2)
            y = plan.pool.get(tuple(out.shape))
            plan.gn_silu(out, gam.to(DEV), bet.to(DEV), G, out=y)
        _run(plan)
        _run(plan)                                                    # a replay starts from y holding the previous result
        outs[prec] = out.cpu().reshape(ref.shape).double()
        if prec == 4 and G:
            refn = F.silu(F.group_norm(ref, G, gam.double(), bet.double(), 1e-5))
            torch.testing.assert_close(y.cpu().reshape(ref.shape).double(), refn, rtol=1e-4, atol=2e-5)
    assert names[4].startswith("conv_wg3") and abs(share.value - 8 / 27) < 1e-12 and not names[0].startswith("conv_wg")
    scale = ref.abs().max().item()
    e0 = (outs[0] - ref).abs().max().item() / scale
    e4 = (outs[4] - ref).abs().max().item() / scale
    print(f"[measured] rel err vs fp64: direct {e0:.2e}  winograd F(2x2x2,3x3x3) {e4:.2e}")
    assert e0 < 4e-6 and e4 < 1e-5, (e0, e4)


def test_conv_winograd_3d_falls_back_where_not_covered(plan_cls):
    """precision=4 descriptors the F(2x2x2,3x3x3) kernel does not take (odd depth, Cout % 64 != 0, 3x3 taps, a fused
    residual) run the F(2x2,3x3) kernel on the same weight buffer."""
    from safediffcon_amd.engine import as5
    for B, cin, cout, sp, k, withres in [(1, 16, 64, (3, 8, 32), (3, 3, 3), False), (1, 16, 96, (4, 8, 32), (3, 3, 3), False),
                                         (2, 16, 64, (1, 8, 32), (1, 3, 3), False), (1, 16, 64, (4, 8, 32), (3, 3, 3), True)]:
        x = det_tensor((B, cin, *sp), 301)
        w, b = det_tensor((cout, cin, *k), 302, 0.2), det_tensor((cout,), 303, 0.1)
        ref = F.conv3d(x.double(), w.double(), b.double(), padding=(k[0] // 2, 1, 1))
        res = det_tensor(tuple(ref.shape), 304) if withres else None
        if withres:
            ref = ref + res.double()
        plan = plan_cls(DEV, precision=4)
        out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, k, pad=(k[0] // 2, 1, 1),
                        residual=None if res is None else as5(res.to(DEV)))
        buf = C.create_string_buffer(128)
        plan.lib.sdc_conv_describe(C.byref(plan.calls[0][1][0]._obj), buf, 128, None)
        assert buf.value.decode().startswith("conv_wg2"), buf.value      # (conv_wg2_kernel, or conv_wg2s_kernel for the (1, 3, 3) taps)
        _run(plan)
        e = (out.cpu().reshape(ref.shape).double() - ref).abs().max().item() / ref.abs().max().item()
        assert e < 1e-5, e


def test_conv_default_mode_over_a_sweep_of_shapes(plan_cls):
    """precision=4 over a seeded sweep of 3x3x3 / 1x3x3 shapes (depths odd and even, rows of 8..64, channel counts on and off
    the tile sizes, with and without a second input / residual / GroupNorm): whatever kernel the dispatch picks -- F(2x2x2),
    F(2x2), F(2,3) or direct -- the result is the fp64 conv within 1e-5 of the output scale, and the fused statistics, where
    offered, normalise like torch."""
    from safediffcon_amd.engine import as5
    g = torch.Generator().manual_seed(1234)
    picks = {}
    for case in range(40):
        r = lambda *opts: opts[int(torch.randint(len(opts), (1,), generator=g))]     # noqa: E731
        kd = r(3, 3, 1)
        B, cin, cout = r(1, 2, 3), r(8, 16, 24, 40, 64), r(32, 64, 96, 128)
        cin1 = r(0, 0, 8, 16)
        D, H, W = (r(1, 2, 3, 4, 6) if kd == 3 else r(1, 2)), r(2, 4, 6, 8, 16), r(8, 16, 32, 64)
        withres, G = r(False, False, True), r(0, 8)
        if case >= 28:      # the last dozen lean towards what the F(2x2x2,3x3x3) kernel takes
            kd, cout, D, withres = 3, r(64, 128), r(2, 4, 6), False
            W = r(16, 32, 64)
            H = r(16, 16, 32) if W == 16 else (r(8, 16) if W == 32 else r(4, 8, 12))
        x = torch.randn(B, cin, D, H, W, generator=g)
        x1 = torch.randn(B, cin1, D, H, W, generator=g) if cin1 else None
        w = torch.randn(cout, cin + cin1, kd, 3, 3, generator=g) * 0.2
        b = torch.randn(cout, generator=g) * 0.1
        ref = F.conv3d((x if x1 is None else torch.cat((x, x1), 1)).double(), w.double(), b.double(), padding=(kd // 2, 1, 1))
        res = torch.randn(ref.shape, generator=g) if withres else None
        if withres:
            ref = ref + res.double()
        plan = plan_cls(DEV, precision=4)
        out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (kd, 3, 3),
                        x1=None if x1 is None else as5(x1.to(DEV)), pad=(kd // 2, 1, 1),
                        residual=None if res is None else as5(res.to(DEV)), gn_groups=G)
        buf = C.create_string_buffer(128)
        assert plan.lib.sdc_conv_describe(C.byref(plan.calls[0][1][0]._obj), buf, 128, None) == 0
        name = buf.value.decode().split("<")[0]
        picks[name] = picks.get(name, 0) + 1
        if G:
            gam, bet = torch.randn(cout, generator=g) * 0.3 + 1.0, torch.randn(cout, generator=g) * 0.2
            y = plan.pool.get(tuple(out.shape))
            plan.gn_silu(out, gam.to(DEV), bet.to(DEV), G, out=y)
        _run(plan)
        e = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert e < 1e-5, (case, name, tuple(x.shape), cin1, cout, kd, e)
        if G:
            refn = F.silu(F.group_norm(ref, G, gam.double(), bet.double(), 1e-5))
            torch.testing.assert_close(y.cpu().double(), refn, rtol=1e-4, atol=3e-5)
    print(f"[measured] kernels picked over the sweep: {picks}")
    assert len(picks) >= 3 and (picks.get("conv_wg3_kernel", 0) + picks.get("conv_wg3s_kernel", 0)) >= 8, picks


def test_conv_winograd_2d_strided_output_and_residual(plan_cls):
    """F(2x2,3x3) kernel writing through a strided view (every other column of a wider buffer) with a residual read
    through another strided view: the scalar (non 8-byte) store / load path of its epilogue."""
    from safediffcon_amd.engine import as5
    B, cin, cout, sp = 3, 32, 96, (8, 32)
    x = det_tensor((B, cin, *sp), 211)
    w, b = det_tensor((cout, cin, 3, 3), 212, 0.2), det_tensor((cout,), 213, 0.1)
    res = det_tensor((B, cout, *sp), 214)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    plan = plan_cls(DEV, precision=3)
    ybuf = torch.full((B, cout, 1, sp[0], 2 * sp[1]), 7.0, device=DEV)
    rbuf = torch.zeros((B, cout, 1, sp[0], 2 * sp[1] + 2), device=DEV)
    rbuf[..., 1:2 * sp[1] + 1:2] = as5(res.to(DEV))
    out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (1, 3, 3), pad=(0, 1, 1),
                    residual=rbuf[..., 1:2 * sp[1] + 1:2], out=ybuf[..., ::2])
    buf = C.create_string_buffer(128)
    plan.lib.sdc_conv_describe(C.byref(plan.calls[0][1][0]._obj), buf, 128, None)
    assert buf.value.decode().startswith("conv_wg2_kernel")
    _run(plan)
    err = (out.cpu().reshape(ref.shape).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err
    assert torch.all(ybuf[..., 1::2] == 7.0)                       # the columns in between are untouched


def test_conv_winograd_2d_falls_back_where_not_covered(plan_cls):
    """precision 3 on shapes the F(2x2,3x3) kernel does not take: odd row counts, rows wider than 128 or not a power of
    two, Conv1d -- run the F(2,3)-along-W or the direct kernel on the same weight buffer and stay correct."""
    from safediffcon_amd.engine import as5
    for nd, B, cin, cout, sp in ((2, 2, 32, 64, (5, 32)), (2, 1, 16, 64, (4, 48)), (1, 3, 48, 64, (128,)), (2, 2, 32, 64, (4, 256))):
        x = det_tensor((B, cin, *sp), 201)
        w, b = det_tensor((cout, cin, *([3] * nd)), 202, 0.2), det_tensor((cout,), 203, 0.1)
        ref = (F.conv1d, F.conv2d)[nd - 1](x.double(), w.double(), b.double(), padding=1)
        plan = plan_cls(DEV, precision=3)
        out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (1,) * (3 - nd) + (3,) * nd,
                        pad=(0,) * (3 - nd) + (1,) * nd)
        buf = C.create_string_buffer(128)
        plan.lib.sdc_conv_describe(C.byref(plan.calls[0][1][0]._obj), buf, 128, None)
        assert not buf.value.decode().startswith("conv_wg2")
        _run(plan)
        err = (out.cpu().reshape(ref.shape).double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 6e-6, (sp, buf.value, err)


@pytest.mark.parametrize("case", [
    dict(B=6, cin=128, cout=64, sp=(8, 64), up=(2, 2)),        # Upsample2d: 16x128 out, Cout 64
    dict(B=40, cin=256, cout=128, sp=(4, 32), up=(2, 2)),      # 128-row tile
    dict(B=3, cin=64, cout=96, sp=(1, 64), up=(1, 2)),         # tokamak Upsample (1-D): kH = 1
    dict(B=2, cin=32, cout=40, sp=(5, 8), up=(2, 2)),          # 16-wide upsampled rows, odd row count, ragged Cout
])
def test_conv_winograd_nearest_upsample(plan_cls, case):
    """nn.Upsample(scale 2, nearest) + conv k3 p1 folded into one launch, direct (precision 0) and Winograd (2)."""
    from safediffcon_amd.engine import as5
    B, cin, cout, sp, up = case["B"], case["cin"], case["cout"], case["sp"], case["up"]
    x = det_tensor((B, cin, *sp), 101)
    kh = 3 if up[0] == 2 else 1
    w, b = det_tensor((cout, cin, kh, 3), 102, 0.2), det_tensor((cout,), 103, 0.1)
    xu = x.double().repeat_interleave(up[0], 2).repeat_interleave(up[1], 3)
    ref = F.conv2d(xu, w.double(), b.double(), padding=(kh // 2, 1))
    outs = {}
    for prec in (0, 2):
        plan = plan_cls(DEV, precision=prec)
        out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (1, kh, 3), pad=(0, kh // 2, 1),
                        up=(1, *up))
        _run(plan)
        outs[prec] = out.cpu().reshape(ref.shape).double()
    scale = ref.abs().max().item()
    e0 = (outs[0] - ref).abs().max().item() / scale
    e2 = (outs[2] - ref).abs().max().item() / scale
    assert e0 < 6e-6 and e2 < 6e-6, (e0, e2)        # K up to 2304: a few fp32 ulps of the output scale
    assert not torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("case", [
    dict(B=3, F=1, C=64, n=2048, pre=0, post=0),       # Burgers top level: LayerNorm in and out, 32 token tiles, splits
    dict(B=5, F=1, C=128, n=128, pre=0, post=0),       # Burgers level 2
    dict(B=2, F=1, C=64, n=128, pre=1, post=1),        # tokamak-style RMSNorm (dim 64 configuration)
    dict(B=2, F=3, C=64, n=256, pre=0, post=-1),       # smoke SpatialLinearAttention: per-frame sequences, no post norm
    dict(B=300, F=1, C=64, n=64, pre=0, post=0),       # one tile per sequence, many sequences
])
def test_linear_attention_block_fused(plan_cls, case):
    """sdc_linattn_block == x + post(Wo LA(pre(x)) + bo) in fp64 torch (1D/model/unet.py:182-222, conv3d.py:232-258)."""
    B, Fr, Cc, n, pre, post = case["B"], case["F"], case["C"], case["n"], case["pre"], case["post"]
    x = det_tensor((B, Cc, Fr, n), 111)
    g1, g2 = det_tensor((Cc,), 112, 0.3) + 1.0, det_tensor((Cc,), 113, 0.3) + 1.0
    wqkv, wo, bo = det_tensor((384, Cc), 114, 0.4), det_tensor((Cc, 128), 115, 0.3), det_tensor((Cc,), 116, 0.1)

    def norm(t, g, mode):          # over the channel axis (dim 1)
        if mode == 0:
            return (t - t.mean(1, keepdim=True)) * (t.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * g.view(1, -1, 1, 1)
        return F.normalize(t, dim=1) * g.view(1, -1, 1, 1) * Cc ** 0.5

    xd = x.double()
    xn = norm(xd, g1.double(), pre)
    qkv = torch.einsum("oc,bcfn->bofn", wqkv.double(), xn)
    q, k, v = (t.reshape(B, 4, 32, Fr, n) for t in qkv.chunk(3, 1))
    q = q.softmax(2) * 32 ** -0.5
    k = k.softmax(-1)
    ctx = torch.einsum("bhdfn,bhefn->bhfde", k, v)
    out = torch.einsum("bhfde,bhdfn->bhefn", ctx, q).reshape(B, 128, Fr, n)
    y = torch.einsum("oc,bcfn->bofn", wo.double(), out) + bo.double().view(1, -1, 1, 1)
    if post >= 0:
        y = norm(y, g2.double(), post)
    ref = y + xd

    plan = plan_cls(DEV)
    xg = x.to(DEV)
    got = plan.linattn_block(xg, g1.to(DEV), plan.conv_weight(wqkv.view(384, Cc, 1).to(DEV)),
                             plan.conv_weight(wo.view(Cc, 128, 1).to(DEV)), bo.to(DEV), g2.to(DEV) if post >= 0 else None,
                             B, Fr, n, (Cc * Fr * n, Fr * n, n), pre, post)
    _run(plan)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("B,H,W", [(2, 4, 8), (1, 16, 16), (3, 2, 4)])
def test_temporal_attention_block_fused(plan_cls, B, H, W):
    """sdc_tattn_block == x + Wo softmax(rot(s q) rot(k)^T + relpos) v over the 32 frames of every pixel, q/k/v from the
    channel-LayerNormed input (conv3d.py:165-184, 277-353), in fp64 torch."""
    from oracle import nets as onets
    Cc, Fr = 64, 32
    x = det_tensor((B, Cc, Fr, H, W), 121)
    g = det_tensor((Cc,), 122, 0.3) + 1.0
    wqkv, wo = det_tensor((384, Cc), 123, 0.3), det_tensor((Cc, 128), 124, 0.3)
    freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))
    relw = det_tensor((32, 4), 125, 0.5)                       # (num_buckets, heads) embedding
    bias = onets.rel_pos_bias(relw, Fr).double()               # (heads, query, key)
    xd = x.double()
    xn = (xd - xd.mean(1, keepdim=True)) * (xd.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() * g.double().view(1, -1, 1, 1, 1)
    tok = xn.permute(0, 3, 4, 2, 1).reshape(B * H * W, Fr, Cc)                      # b (h w) f c
    q, k, v = (tok @ wqkv.double().t()).chunk(3, -1)
    sp = lambda t: t.reshape(-1, Fr, 4, 32).permute(0, 2, 1, 3)                     # n heads f d
    q, k, v = sp(q) * 32 ** -0.5, sp(k), sp(v)
    q, k = onets.rotary(q, freqs.double()), onets.rotary(k, freqs.double())
    att = (q @ k.transpose(-1, -2) + bias[None]).softmax(-1)
    out = (att @ v).permute(0, 2, 1, 3).reshape(-1, Fr, 128) @ wo.double().t()
    ref = xd + out.reshape(B, H, W, Fr, Cc).permute(0, 4, 3, 1, 2)

    plan = plan_cls(DEV)
    ang = torch.arange(Fr, dtype=torch.float32)[:, None] * freqs[None, :]
    rot = torch.stack((ang.cos(), ang.sin()), dim=-1).reshape(-1).to(DEV)
    got = plan.tattn_block(x.to(DEV), g.to(DEV), plan.conv_weight(wqkv.view(384, Cc, 1).to(DEV)),
                           plan.conv_weight(wo.view(Cc, 128, 1).to(DEV)), rot, bias.float().reshape(-1).to(DEV))
    _run(plan)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=2e-4, atol=1e-4)       # |y| up to ~10 here


@pytest.mark.parametrize("case", [
    dict(B=300, cin=64, cout=64, sp=(16, 128), G=1),           # 64x512 tiles, one group spanning the row tile
    dict(B=40, cin=32, cout=64, sp=(16, 128), G=8),            # 64x256 tiles, 8 groups of 8 channels inside the row tile
    dict(B=64, cin=32, cout=128, sp=(16, 64), G=1),            # 128x256 tiles
    dict(B=260, cin=16, cout=256, sp=(16, 16), G=8),           # 128x256 tiles, groups of 32 channels, 16-wide rows
    dict(B=40, cin=32, cout=256, sp=(8, 64), G=1),             # 128x128 tiles, the group spans two row tiles
    dict(B=6, cin=32, cout=128, sp=(8, 32), G=8),              # 64x128 tiles (4 waves)
    dict(B=40, cin=32, cout=192, sp=(8, 64), G=1, fused=False),   # 192 rows do not line up with 128-row tiles -> plain pass
    dict(B=3, cin=32, cout=64, sp=(2, 16), G=1, fused=False),     # sample smaller than a tile -> plain statistics pass
])
def test_conv_groupnorm_statistics_in_the_epilogue(plan_cls, case):
    """conv -> GroupNorm+SiLU with the statistics summed in the conv epilogue (sdc_conv_gn + sdc_gn_finalize) against the
    two-pass form (sdc_conv + sdc_gn_stats) and against torch in fp64."""
    from safediffcon_amd.engine import as5
    B, cin, cout, sp, G = case["B"], case["cin"], case["cout"], case["sp"], case["G"]
    x = det_tensor((B, cin, *sp), 131)
    w, b = det_tensor((cout, cin, 3, 3), 132, 0.2), det_tensor((cout,), 133, 0.3)
    gam, bet = det_tensor((cout,), 134, 0.3) + 1.0, det_tensor((cout,), 135, 0.2)
    ref = F.silu(F.group_norm(F.conv2d(x.double(), w.double(), b.double(), padding=1), G, gam.double(), bet.double(), 1e-5))
    outs = []
    for fuse in (True, False):
        plan = plan_cls(DEV, precision=2)
        plan.fuse_gn_stats = fuse
        h = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (1, 3, 3), pad=(0, 1, 1), gn_groups=G)
        plan.gn_silu(h, gam.to(DEV), bet.to(DEV), G)
        used = any(fn is plan.lib.sdc_conv_gn for fn, _ in plan.calls)
        assert used == (fuse and case.get("fused", True))
        _run(plan)
        outs.append(h.cpu().reshape(ref.shape).double())
    torch.testing.assert_close(outs[0], ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("B,Cc,Co,sp", [(2, 16, 32, (3, 8, 8)), (1, 64, 48, (2, 5, 7)), (3, 32, 32, (1, 16, 16))])
def test_conv_transpose_subpixel(plan_cls, B, Cc, Co, sp):
    """ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1) as four 2x2 sub-pixel convs == F.conv_transpose3d == the zero-stuffed form."""
    x = det_tensor((B, Cc, *sp), 141)
    w, b = det_tensor((Cc, Co, 1, 4, 4), 142, 0.2), det_tensor((Co,), 143, 0.1)
    ref = F.conv_transpose3d(x.double(), w.double(), b.double(), stride=(1, 2, 2), padding=(0, 1, 1))
    plan = plan_cls(DEV)
    out = plan.conv_transpose_422(x.to(DEV), w.to(DEV), b.to(DEV), Co)
    old = plan.conv(x.to(DEV), plan.conv_weight(w.to(DEV), "convT"), b.to(DEV), Co, (1, 4, 4), up=(1, 2, 2), up_mode=1, pad=(0, 2, 2))
    _run(plan)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(old.cpu().double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,Cc,Co,sp", [(2, 16, 32, (8, 16)), (1, 64, 48, (5, 7)), (3, 128, 64, (4, 32))])
def test_upsample_conv_subpixel(plan_cls, B, Cc, Co, sp):
    """nearest x2 + 3x3 conv as four merged-tap 2x2 convs == F.interpolate + F.conv2d."""
    from safediffcon_amd.engine import as5
    x = det_tensor((B, Cc, *sp), 151)
    w, b = det_tensor((Co, Cc, 3, 3), 152, 0.2), det_tensor((Co,), 153, 0.1)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    plan = plan_cls(DEV)
    out = plan.upsample2_conv3(as5(x.to(DEV)), w.to(DEV), b.to(DEV), Co)
    _run(plan)
    torch.testing.assert_close(out.cpu().reshape(ref.shape).double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", [
    dict(B=300, cin=64, cout=64, sp=(16, 128), k=1),                            # widest tile
    dict(B=300, cin=128, cout=64, sp=(16, 128), k=1, cin1=64, residual=True),   # concat + residual
    dict(B=64, cin=64, cout=384, sp=(16, 64), k=1),                             # 128x256 tiles
    dict(B=40, cin=32, cout=192, sp=(8, 64), k=1),                              # 128x128 tiles, ragged Cout tile
    dict(B=48, cin=32, cout=48, sp=(16, 128), k=1),                             # 64x256 tiles, ragged Cout
    dict(B=260, cin=16, cout=160, sp=(16, 16), k=1),                            # 16-wide rows, 128x256 tiles
    dict(B=300, cin=64, cout=64, sp=(16, 128), k=3, prec=0),                    # 3 taps, direct (precision 0)
    dict(B=64, cin=32, cout=128, sp=(16, 64), k=3, prec=0, residual=True),
    dict(B=2, cin=32, cout=96, sp=(8, 16, 16), k=3, prec=0, nd=3),              # 3x3x3 direct, small grid
    dict(B=6, cin=32, cout=96, sp=(16, 32, 32), k=3, prec=0, nd=3),             # 3x3x3 direct, large grid
])
def test_conv_direct_big_grids(plan_cls, case):
    """direct-form convs (1x1, and 3-tap in precision 0) on grids large enough for the biggest tiles, against fp64 torch"""
    from safediffcon_amd.engine import as5
    nd = case.get("nd", 2)
    B, cin, cout, sp, k = case["B"], case["cin"], case["cout"], case["sp"], case["k"]
    cin1, prec = case.get("cin1", 0), case.get("prec", 2)
    x, x1 = det_tensor((B, cin, *sp), 161), (det_tensor((B, cin1, *sp), 162) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, *([k] * nd)), 163, 0.2), det_tensor((cout,), 164, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = (F.conv2d, F.conv3d)[nd - 2](xin.double(), w.double(), b.double(), padding=k // 2)
    res = det_tensor(tuple(ref.shape), 165) if case.get("residual") else None
    if res is not None:
        ref = ref + res.double()
    plan = plan_cls(DEV, precision=prec)
    k3, p3 = (1,) * (3 - nd) + (k,) * nd, (0,) * (3 - nd) + (k // 2,) * nd
    out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, k3,
                    x1=None if x1 is None else as5(x1.to(DEV)), pad=p3, residual=None if res is None else as5(res.to(DEV)))
    _run(plan)
    got = out.cpu().reshape(ref.shape).double()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() / scale < 3e-6


@pytest.mark.parametrize("B,Cc,G,sp,cond,res", [(128, 256, 1, (64,), True, False), (96, 256, 1, (4, 32), True, True),
                                                 (64, 2048, 1, (16,), False, True), (80, 64, 8, (2, 8, 8), False, False),
                                                 (70, 96, 1, (3, 7), True, True),
                                                 (2, 256, 1, (64,), True, True)])       # the choice does not depend on the batch
def test_gn_fused_small_groups_equal_the_three_launch_path(B, Cc, G, sp, cond, res):
    """sdc_gn_fused (statistics + apply in one launch for small groups: the deep levels of Unet2D / Unet1D) against
    sdc_gn_stats + sdc_gn_apply and against torch in fp64"""
    from safediffcon_amd.engine import Plan, as5
    x = det_tensor((B, Cc, *sp), 170)
    gamma, beta = (1 + 0.1 * det_tensor((Cc,), 171)).to(DEV), (0.1 * det_tensor((Cc,), 172)).to(DEV)
    ss = (0.2 * det_tensor((B, 2 * Cc), 173)).to(DEV) if cond else None
    r = det_tensor((B, Cc, *sp), 174).to(DEV) if res else None
    outs, used = [], []
    for small in (True, False):
        plan = Plan(DEV)
        plan.fuse_gn_small = small
        xx = as5(x.to(DEV).clone())
        y = plan.gn_silu(xx, gamma, beta, G, ss=ss, ss_b_stride=2 * Cc if cond else 0, residual=None if r is None else as5(r))
        plan.run(torch.cuda.current_stream().cuda_stream)
        outs.append(y.clone())
        used.append({fn.__name__ for fn, _ in plan.calls})
    assert used[0] == {"sdc_gn_fused"} and "sdc_gn_stats" in used[1]
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-6          # fp64 statistics on both sides, summed in another order
    u = F.group_norm(x.double(), G, gamma.double().cpu(), beta.double().cpu(), eps=1e-5)
    if cond:
        bs = (B, Cc) + (1,) * len(sp)
        u = u * (ss.double().cpu()[:, :Cc].reshape(bs) + 1) + ss.double().cpu()[:, Cc:].reshape(bs)
    want = F.silu(u) + (r.double().cpu() if res else 0)
    assert (outs[0].cpu().double().reshape(want.shape) - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("case", [
    dict(B=8, cin=256, cout=256, W=128, G=1),                              # tokamak level 0: GroupNorm(1) sums in the epilogue
    dict(B=8, cin=512, cin1=512, cout=512, W=32, residual=True),           # up path: skip concat, residual
    dict(B=16, cin=1024, cout=2048, W=16, G=1),                            # deepest level: 16-wide rows, 8 samples per tile (no fused sums)
    dict(B=3, cin=48, cout=160, W=64, G=1),                                # ragged Cout (128 + 32), Cin = 3 stages
    dict(B=5, cin=64, cout=128, W=20),                                     # W = 20: rows shorter than a tile and not a divisor -> not covered
    dict(B=2, cin=32, cout=128, W=256, residual=True),                     # rows longer than a tile
])
def test_conv1d_f43_mode(plan_cls, case):
    """Conv1d k3 (the tokamak Unet1D's hot op, tokamak/model/unet.py Block) on the F(4,3) Winograd kernel: six products per output
    quad = half of the direct form's MFMA work.  fp32 end to end; the larger transform constants cost accuracy against fp64:
    gate 1e-5 of the output scale, measured 2-6e-6 (F(2,3): ~1e-6).  GroupNorm statistics from its epilogue normalise
    like torch."""
    from safediffcon_amd.engine import as5
    B, cin, cin1, cout, W, G = case["B"], case["cin"], case.get("cin1", 0), case["cout"], case["W"], case.get("G", 0)
    x, x1 = det_tensor((B, cin, W), 401), (det_tensor((B, cin1, W), 402) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, 3), 403, 0.2 / (cin + cin1) ** 0.5 * 8), det_tensor((cout,), 404, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = F.conv1d(xin.double(), w.double(), b.double(), padding=1)
    res = det_tensor(tuple(ref.shape), 405) if case.get("residual") else None
    if res is not None:
        ref = ref + res.double()
    plan = plan_cls(DEV, precision=5)          # opt-in mode: as 4, plus F(4,3) on the 1-D convs
    out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (1, 1, 3),
                    x1=None if x1 is None else as5(x1.to(DEV)), pad=(0, 0, 1), residual=None if res is None else as5(res.to(DEV)),
                    gn_groups=G)
    buf, share = C.create_string_buffer(128), C.c_double(0)
    assert plan.lib.sdc_conv_describe(C.byref(plan.calls[0][1][0]._obj), buf, 128, C.byref(share)) == 0
    covered = W % 4 == 0 and (W % 128 == 0 or 128 % W == 0)
    assert buf.value.decode().endswith("F43>") == covered, buf.value      # conv_f43_kernel<128,128,4,16,F43>
    assert (share.value == 0.5) == covered
    if G:
        gam, bet = 1 + 0.3 * det_tensor((cout,), 406), 0.2 * det_tensor((cout,), 407)
        y = plan.pool.get(tuple(out.shape))
        plan.gn_silu(out, gam.to(DEV), bet.to(DEV), G, out=y)
        assert any(fn.__name__ == "sdc_conv_gn" for fn, _ in plan.calls) == (covered and W % 128 == 0)   # a tile must lie inside one sample
    _run(plan)
    e = (out.cpu().reshape(ref.shape).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[measured] conv1d {buf.value.decode()} {case}: rel err vs fp64 {e:.2e}")
    assert e < 1e-5, e
    if G:
        want = F.silu(F.group_norm(ref, G, gam.double(), bet.double(), eps=1e-5))
        assert (y.cpu().reshape(want.shape).double() - want).abs().max().item() < 6e-5     # (the conv error over the group's std)


@pytest.mark.parametrize("Cc,F_,hw,res", [(64, 3, (16, 16), True), (128, 2, (8, 16), True), (64, 1, (8, 8), False)])
def test_linattn_block_with_groupnorm_on_load(plan_cls, Cc, F_, hw, res):
    """sdc_linattn_block_gn (the producing ResnetBlock's GroupNorm(8) + SiLU + residual applied while pass 1 loads its tiles,
    conv3d.py:189-230 -> :232-258) == sdc_gn_stats + sdc_gn_apply followed by sdc_linattn_block, bit for bit: same expressions,
    same statistics kernel.  (The statistics buffer carries sdc_gn_stats' fp64 partials behind the (mean, rstd) pairs.)"""
    from safediffcon_amd.engine import Plan
    B, G = 2, 8
    H, W = hw
    x = det_tensor((B, Cc, F_, H, W), 501).to(DEV)
    r = det_tensor((B, Cc, F_, H, W), 502).to(DEV) if res else None
    gam, bet = (1 + 0.2 * det_tensor((Cc,), 503)).to(DEV), (0.1 * det_tensor((Cc,), 504)).to(DEV)
    g_pre = (1 + 0.1 * det_tensor((Cc,), 505)).to(DEV)
    outs = []
    for fused in (True, False):
        plan = Plan(DEV)
        wqkv = plan.conv_weight(det_tensor((384, Cc, 1, 1), 506, 0.1).to(DEV))
        wo = plan.conv_weight(det_tensor((Cc, 128, 1, 1), 507, 0.1).to(DEV))
        bo = (0.1 * det_tensor((Cc,), 508)).to(DEV)
        xx = x.clone()
        n = H * W
        strides = (Cc * F_ * n, F_ * n, n)
        if fused:
            st = plan.gn_stats_deferred(xx, G)
            y = plan.linattn_block(xx, g_pre, wqkv, wo, bo, None, B, F_, n, strides, 0, -1, gn=(st, gam, bet, G, r))
        else:
            h = plan.gn_silu(xx, gam, bet, G, residual=r)
            y = plan.linattn_block(h, g_pre, wqkv, wo, bo, None, B, F_, n, strides, 0, -1)
        _run(plan)
        outs.append(y.clone())
        assert torch.isfinite(y).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("case", [
    dict(B=8, cin=256, cout=256, W=128, G=1),                              # tokamak level 0 (single-tap form, 16 stages)
    dict(B=128, cin=256, cout=256, W=128, G=1),                            # the same at the C3 batch: the 128 x 128 tile instance
    dict(B=128, cin=48, cout=128, W=128),                                  # ... and its general form (3 stages)
    dict(B=8, cin=512, cin1=512, cout=512, W=32, residual=True),           # up path: skip concat (two inputs), residual
    dict(B=16, cin=1024, cout=2048, W=16, G=1),                            # deepest level: 16-wide rows
    dict(B=3, cin=48, cout=160, W=64, G=1),                                # Cin = 3 stages (odd: the general form), ragged Cout (128 + 32)
    dict(B=2, cin=32, cout=96, W=256, residual=True),                      # 2 stages, rows longer than a tile, ragged Cout
    dict(B=4, cin=16, cin1=32, cout=64, W=128),                            # 3 stages over two inputs
])
def test_conv1d_f23_default_mode(plan_cls, case):
    """Conv1d k3 on the default (precision 4) path: F(2,3) along W, `conv_wg_kernel` -- its single-tap specialisation (even stage
    counts: gather offsets and padding mask formed once, compile-time LDS buffers, no zeroing of weight rows beyond Cout) and the
    general form (odd stage counts) against fp64; GroupNorm statistics from the epilogue normalise like torch."""
    from safediffcon_amd.engine import as5
    B, cin, cin1, cout, W, G = case["B"], case["cin"], case.get("cin1", 0), case["cout"], case["W"], case.get("G", 0)
    x, x1 = det_tensor((B, cin, W), 411), (det_tensor((B, cin1, W), 412) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, 3), 413, 0.2 / (cin + cin1) ** 0.5 * 8), det_tensor((cout,), 414, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = F.conv1d(xin.double(), w.double(), b.double(), padding=1)
    res = det_tensor(tuple(ref.shape), 415) if case.get("residual") else None
    if res is not None:
        ref = ref + res.double()
    plan = plan_cls(DEV, precision=4)
    out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (1, 1, 3),
                    x1=None if x1 is None else as5(x1.to(DEV)), pad=(0, 0, 1), residual=None if res is None else as5(res.to(DEV)),
                    gn_groups=G)
    buf, share = C.create_string_buffer(128), C.c_double(0)
    assert plan.lib.sdc_conv_describe(C.byref(plan.calls[0][1][0]._obj), buf, 128, C.byref(share)) == 0
    assert buf.value.decode().startswith("conv_wg_kernel") and abs(share.value - 2 / 3) < 1e-12, buf.value
    if G:
        gam, bet = 1 + 0.3 * det_tensor((cout,), 416), 0.2 * det_tensor((cout,), 417)
        y = plan.pool.get(tuple(out.shape))
        plan.gn_silu(out, gam.to(DEV), bet.to(DEV), G, out=y)
    _run(plan)
    e = (out.cpu().reshape(ref.shape).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[measured] conv1d {buf.value.decode()} {case}: rel err vs fp64 {e:.2e}")
    assert e < 3e-6, e
    if G:
        want = F.silu(F.group_norm(ref, G, gam.double(), bet.double(), eps=1e-5))
        assert (y.cpu().reshape(want.shape).double() - want).abs().max().item() < 2e-5


# ------------------------------------------------------------------ round 6: the round-5 dispatches, pinned by name
def _describe(plan, i=0):
    buf, share = C.create_string_buffer(128), C.c_double(0)
    assert plan.lib.sdc_conv_describe(C.byref(plan.calls[i][1][0]._obj), buf, 128, C.byref(share)) == 0
    return buf.value.decode(), share.value


@pytest.mark.parametrize("case", [
    dict(B=1, cin=16, cout=64, sp=(2, 4, 64), G=0, want="conv_wg3s_kernel<64>"),                 # rows of 64, one input, no statistics
    dict(B=2, cin=64, cout=64, sp=(4, 6, 64), cin1=64, G=8, want="conv_wg3s_kernel<64>"),        # 64 + 64 -> 64 (ups), GN epilogue
    dict(B=1, cin=24, cout=128, sp=(6, 8, 64), G=1, want="conv_wg3s_kernel<64>"),                # Cout 128: two channel tiles per position tile
    dict(B=2, cin=32, cout=64, sp=(2, 8, 32), G=8, want="conv_wg3s_kernel<32>"),                 # rows of 32: two row pairs per workgroup
    dict(B=1, cin=16, cout=128, sp=(4, 4, 32), cin1=16, G=4, want="conv_wg3s_kernel<32>"),       # rows of 32, two inputs, Cout 128
    dict(B=1, cin=32, cout=64, sp=(2, 16, 16), G=0, want="conv_wg3_kernel<16>"),                 # rows of 16 stay with the one-workgroup form
    dict(B=1, cin=16, cout=64, sp=(3, 4, 64), G=0, want="conv_wg2_kernel"),                      # odd depth: F(2x2,3x3) over (H, W)
    dict(B=1, cin=16, cout=64, sp=(5, 8, 32), G=8, want="conv_wg2_kernel"),                      # odd depth with statistics
    dict(B=1, cin=16, cout=64, sp=(2, 6, 32), G=0, want="conv_wg2_kernel"),                      # 3 row pairs: not a whole number of 32-tile workgroups
])
def test_conv_wg3s_dispatch_is_pinned_and_matches_fp64(plan_cls, case):
    """the two-workgroups-per-CU F(2x2x2,3x3x3) kernel is THE kernel compared with fp64 here: sdc_conv_describe names it for rows
    of 64 and 32 (one and two inputs, with and without the GroupNorm epilogue, Cout 64 / 128), rows of 16 name conv_wg3_kernel,
    odd depths and ragged row-pair counts fall to conv_wg2_kernel -- each against torch in fp64 (gate 1e-5 of the output scale)."""
    from safediffcon_amd.engine import as5
    B, cin, cout, sp, cin1, G = case["B"], case["cin"], case["cout"], case["sp"], case.get("cin1", 0), case["G"]
    x, x1 = det_tensor((B, cin, *sp), 601), (det_tensor((B, cin1, *sp), 602) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, 3, 3, 3), 603, 0.2), det_tensor((cout,), 604, 0.1)
    ref = F.conv3d((x if x1 is None else torch.cat((x, x1), 1)).double(), w.double(), b.double(), padding=1)
    plan = plan_cls(DEV, precision=4)
    out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, (3, 3, 3),
                    x1=None if x1 is None else as5(x1.to(DEV)), pad=(1, 1, 1), gn_groups=G)
    name, share = _describe(plan)
    assert name.startswith(case["want"]), (name, case["want"])
    assert abs(share - (8 / 27 if "wg3" in case["want"] else 4 / 9)) < 1e-12
    y = None
    if G:
        assert plan.calls[0][0] is plan.lib.sdc_conv_gn          # the statistics come out of the conv's epilogue
        gam, bet = det_tensor((cout,), 605, 0.3) + 1.0, det_tensor((cout,), 606, 0.2)
        y = plan.pool.get(tuple(out.shape))
        plan.gn_silu(out, gam.to(DEV), bet.to(DEV), G, out=y)
    _run(plan)
    _run(plan)                                                    # (y is scratch while the kernel runs: a replay must not see the old result)
    e = (out.cpu().reshape(ref.shape).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[measured] {name}: rel err vs fp64 {e:.2e}")
    assert e < 1e-5, (name, e)
    if G:
        refn = F.silu(F.group_norm(ref, G, gam.double(), bet.double(), 1e-5))
        torch.testing.assert_close(y.cpu().reshape(ref.shape).double(), refn, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("case", [
    dict(B=8, cin=32, cout=128, sp=(32, 32, 32), want="conv_pw2_kernel<2,2>"),                          # exactly 1024 tiles of 128 x 256
    dict(B=3, cin=48, cout=128, sp=(8, 121, 92), residual=True, want="conv_pw2_kernel<2,2>"),           # ragged last position tile, residual
    dict(B=2, cin=16, cout=384, sp=(8, 128, 128), cin1=16, want="conv_pw2_kernel<2,2>"),                # two inputs, three channel blocks
    dict(B=4, cin=32, cout=64, sp=(8, 128, 128), want="conv_pw2_kernel<1,4>"),                          # Cout 64: 64 x 512 tiles
    dict(B=5, cin=128, cout=64, sp=(4, 164, 164), residual=True, want="conv_pw2_kernel<1,4>"),          # ragged, residual, K = 128 (8 chunks)
    dict(B=8, cin=32, cout=96, sp=(32, 32, 32), want="conv_pw_kernel"),                                 # Cout % 128 != 0: the older kernel
    dict(B=2, cin=32, cout=128, sp=(16, 32, 32), want="conv_pw_kernel"),                                # < 1024 tiles: the older kernel
])
def test_conv_pw2_dispatch_is_pinned_and_matches_fp64(plan_cls, case):
    """the interleaved-tile 1x1 kernel by name (`<2,2>` and `<1,4>`; residual; ragged position counts; two inputs) against fp64
    torch, and -- DESIGN 3.7's claim -- BIT-identical to conv_pw_kernel: the same call into an output buffer whose batch stride is
    not a multiple of 4 floats takes the older kernel (its k-ordered fp32 FMA chains are the same)."""
    from safediffcon_amd.engine import as5
    B, cin, cout, sp, cin1 = case["B"], case["cin"], case["cout"], case["sp"], case.get("cin1", 0)
    x, x1 = det_tensor((B, cin, *sp), 611), (det_tensor((B, cin1, *sp), 612) if cin1 else None)
    w, b = det_tensor((cout, cin + cin1, 1, 1, 1), 613, 0.2), det_tensor((cout,), 614, 0.1)
    res = det_tensor((B, cout, *sp), 615) if case.get("residual") else None
    xd, x1d, wd, bd = x.to(DEV), (None if x1 is None else x1.to(DEV)), w.to(DEV), b.to(DEV)
    resd = None if res is None else res.to(DEV)
    ref = F.conv3d((xd if x1d is None else torch.cat((xd, x1d), 1)).double(), wd.double(), bd.double())
    if resd is not None:
        ref = ref + resd.double()
    outs = {}
    for tag in ("aligned", "shifted"):
        plan = plan_cls(DEV, precision=4)
        out = None
        if tag == "shifted":
            n = cout * sp[0] * sp[1] * sp[2]
            wide = torch.zeros(B, n + 1, device=DEV)              # batch stride n + 1: rows of y no longer 16-byte aligned
            out = wide[:, :n].view(B, cout, *sp)
        out = plan.conv(as5(xd), plan.conv_weight(wd), bd, cout, (1, 1, 1), x1=None if x1d is None else as5(x1d),
                        residual=None if resd is None else as5(resd), out=out)
        name, _ = _describe(plan)
        if tag == "aligned":
            assert name.startswith(case["want"]), (name, case["want"])
        else:
            assert name.startswith("conv_pw_kernel"), name
        _run(plan)
        outs[tag] = out.reshape(ref.shape).clone()
    e = (outs["aligned"].double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[measured] {case['want']}: rel err vs fp64 {e:.2e}")
    assert e < 3e-6, e
    assert torch.equal(outs["aligned"], outs["shifted"])


@pytest.mark.parametrize("case", [
    dict(B=2, cin=8, cout=64, sp=(2, 128), G=0, want="conv_wg2s_kernel<128>"),                  # one row pair, two half-row workgroups, two stages
    dict(B=3, cin=24, cout=128, sp=(6, 128), cin1=8, G=8, want="conv_wg2s_kernel<128>"),         # two inputs, two channel tiles, GN epilogue
    dict(B=2, cin=64, cout=64, sp=(16, 128), G=1, want="conv_wg2s_kernel<128>"),                 # the Burgers level-0 conv
    dict(B=2, cin=64, cout=64, sp=(8, 64), G=1, want="conv_wg2s_kernel<64>"),                    # rows of 64: one row pair per workgroup
    dict(B=2, cin=16, cout=64, sp=(4, 32), G=0, want="conv_wg2s_kernel<32>"),                    # rows of 32: two row pairs per workgroup
    dict(B=1, cin=32, cout=192, sp=(8, 32), cin1=32, G=3, want="conv_wg2s_kernel<32>"),          # rows of 32, two inputs, three channel tiles
    dict(B=2, cin=16, cout=64, sp=(2, 4, 64), G=8, k3=(1, 3, 3), want="conv_wg2s_kernel<64>"),   # a Conv3d with (1, 3, 3) taps: planes are the D axis
    dict(B=1, cin=16, cout=64, sp=(6, 32), G=0, want="conv_wg2_kernel"),                         # 3 row pairs of 32: not whole workgroups
    dict(B=1, cin=16, cout=96, sp=(4, 64), G=0, want="conv_wg2_kernel"),                         # Cout % 64 != 0
    dict(B=1, cin=20, cout=64, sp=(4, 64), G=0, want="conv_kernel"),                             # Cin % 8 != 0: the direct kernel
])
def test_conv_wg2s_dispatch_is_pinned_and_matches_fp64(plan_cls, case):
    """the two-workgroups-per-CU F(2x2,3x3) kernel (round 6) by name -- rows of 128 (half-row workgroups with the
    halo column at their inner end), 64 and 32, one and two inputs, with and without the GroupNorm epilogue -- against torch in fp64
    (gate 1e-5 of the output scale); shapes outside its contract fall to conv_wg2_kernel."""
    from safediffcon_amd.engine import as5
    B, cin, cout, sp, cin1, G = case["B"], case["cin"], case["cout"], case["sp"], case.get("cin1", 0), case["G"]
    nd = len(sp)
    x, x1 = det_tensor((B, cin, *sp), 701), (det_tensor((B, cin1, *sp), 702) if cin1 else None)
    if nd == 2:
        w = det_tensor((cout, cin + cin1, 3, 3), 703, 0.2)
        k3, p3 = (1, 3, 3), (0, 1, 1)
    else:
        w = det_tensor((cout, cin + cin1, 1, 3, 3), 703, 0.2)
        k3, p3 = (1, 3, 3), (0, 1, 1)
    b = det_tensor((cout,), 704, 0.1)
    xin = x if x1 is None else torch.cat((x, x1), 1)
    ref = (F.conv2d(xin.double(), w.double(), b.double(), padding=1) if nd == 2
           else F.conv3d(xin.double(), w.double(), b.double(), padding=(0, 1, 1)))
    plan = plan_cls(DEV, precision=4)
    out = plan.conv(as5(x.to(DEV)), plan.conv_weight(w.to(DEV)), b.to(DEV), cout, k3,
                    x1=None if x1 is None else as5(x1.to(DEV)), pad=p3, gn_groups=G)
    name, share = _describe(plan)
    assert name.startswith(case["want"]), (name, case["want"])
    y = None
    if G and "wg2" in name:
        assert plan.calls[0][0] is plan.lib.sdc_conv_gn
        gam, bet = det_tensor((cout,), 705, 0.3) + 1.0, det_tensor((cout,), 706, 0.2)
        y = plan.pool.get(tuple(out.shape))
        plan.gn_silu(out, gam.to(DEV), bet.to(DEV), G, out=y)
    _run(plan)
    e = (out.cpu().reshape(ref.shape).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[measured] {name}: rel err vs fp64 {e:.2e}")
    assert e < 1e-5, (name, e)
    if y is not None:
        refn = F.silu(F.group_norm(ref, G, gam.double(), bet.double(), 1e-5))
        torch.testing.assert_close(y.cpu().reshape(ref.shape).double(), refn, rtol=1e-4, atol=2e-5)
