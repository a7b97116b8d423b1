"""Oracle: the three denoiser U-Nets as pure functions of a flat parameter dict.

``P`` maps the reference's ``state_dict`` key names to fp32 CPU tensors, so the
same dict feeds the reference module (``load_state_dict``), this oracle and the
HIP engine.  Written from SURVEY.md section 3.3 / 8a; each function cites the
reference lines whose arithmetic it restates.

  burgers  : Unet2D               1D/model/unet.py:263-426
  tokamak  : Unet1D               tokamak/model/unet.py:263-408
  smoke    : Unet3D_with_Conv3D   2d/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:357-574

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F


class _View:
    """prefix view over the flat parameter dict"""

    def __init__(self, P, prefix=""):
        self.P, self.prefix = P, prefix

    def __getitem__(self, k):
        return self.P[self.prefix + k]

    def __contains__(self, k):
        return (self.prefix + k) in self.P

    def sub(self, p):
        return _View(self.P, self.prefix + p + ".")


# ----------------------------------------------------------------------------
# shared pieces
# ----------------------------------------------------------------------------

def sinusoidal_embedding(t, dim, theta=10000.0):
    """1D/model/unet.py:81-107 (even dim branch), conv3d.py:139-151."""
    half = dim // 2
    if dim % 2 == 0:
        f = torch.exp(torch.arange(half, device=t.device) * -(math.log(theta) / (half - 1)))
        a = t[:, None] * f[None, :]
        return torch.cat((a.sin(), a.cos()), dim=-1)
    f = torch.exp(torch.arange(half, device=t.device) * -(math.log(theta) / (half - 1)))
    a = t[:, None] * f[None, :]
    half1 = (dim + 1) // 2
    f1 = torch.exp(torch.arange(half1, device=t.device) * -(math.log(theta) / (half1 - 1)))
    a1 = t[:, None] * f1[None, :]
    return torch.cat((a.sin(), a1.cos()), dim=-1)


def time_mlp(V, t, dim):
    """Sinusoidal -> Linear -> GELU(erf) -> Linear. unet.py:310-315."""
    e = sinusoidal_embedding(t, dim)
    e = F.linear(e, V["1.weight"], V["1.bias"])
    e = F.gelu(e)
    return F.linear(e, V["3.weight"], V["3.bias"])


def _conv(x, w, b, nd, **kw):
    return (F.conv1d, F.conv2d, F.conv3d)[nd - 1](x, w, b, **kw)


def conv_block(V, x, nd, groups, scale_shift=None):
    """Block: conv3 -> GroupNorm -> x*(scale+1)+shift -> SiLU.
    1D/model/unet.py:128-147, conv3d.py:189-204."""
    x = _conv(x, V["proj.weight"], V["proj.bias"], nd, padding=1)
    x = F.group_norm(x, groups, V["norm.weight"], V["norm.bias"], eps=1e-5)
    if scale_shift is not None:
        sc, sh = scale_shift
        x = x * (sc + 1) + sh
    return F.silu(x)


def resnet_block(V, x, temb, nd, groups):
    """1D/model/unet.py:149-180, conv3d.py:206-230."""
    ss = None
    if "mlp.1.weight" in V and temb is not None:
        e = F.linear(F.silu(temb), V["mlp.1.weight"], V["mlp.1.bias"])
        e = e.reshape(e.shape[0], e.shape[1], *([1] * nd))
        ss = e.chunk(2, dim=1)
    h = conv_block(V.sub("block1"), x, nd, groups, ss)
    h = conv_block(V.sub("block2"), h, nd, groups)
    if "res_conv.weight" in V:
        x = _conv(x, V["res_conv.weight"], V["res_conv.bias"], nd)
    return h + x


def chan_layernorm(x, g, eps=1e-5):
    """channel LayerNorm, gain only.  1D/model/unet.py:53-63 (rsqrt form),
    conv3d.py:165-174 (divide-by-sqrt form) -- same value to 1 ulp."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + eps).rsqrt() * g


def chan_layernorm_div(x, g, eps=1e-5):
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) / (var + eps).sqrt() * g


def chan_rmsnorm(x, g):
    """tokamak/model/unet.py:45-51."""
    return F.normalize(x, dim=1) * g * (x.shape[1] ** 0.5)


def linear_attention(V, x, nd, heads=4, dim_head=32, out_norm=None):
    """1D/model/unet.py:182-222 ; conv3d.py:232-258 (per-frame, no out norm).
    x: (B, C, *spatial).  Returns to_out(...) (no residual)."""
    B = x.shape[0]
    sp = x.shape[2:]
    qkv = _conv(x, V["to_qkv.weight"], None, nd)
    q, k, v = (t.reshape(B, heads, dim_head, -1) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, heads * dim_head, *sp)
    if out_norm is None:
        return _conv(out, V["to_out.weight"], V["to_out.bias"], nd)
    out = _conv(out, V["to_out.0.weight"], V["to_out.0.bias"], nd)
    return out_norm(out, V["to_out.1.g"])


def full_attention(V, x, nd, heads=4, dim_head=32):
    """1D/model/unet.py:224-258 (conv qkv, bias on to_out)."""
    B = x.shape[0]
    sp = x.shape[2:]
    qkv = _conv(x, V["to_qkv.weight"], None, nd)
    q, k, v = (t.reshape(B, heads, dim_head, -1) for t in qkv.chunk(3, dim=1))
    q = q * dim_head ** -0.5
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)          # (B,h,n,d)
    out = out.permute(0, 1, 3, 2).reshape(B, heads * dim_head, *sp)
    return _conv(out, V["to_out.weight"], V["to_out.bias"], nd)


def _unet_lucid(P, x, t, *, nd, dim, dim_mults, groups, prenorm, outnorm, down, up):
    """Shared skeleton of Unet2D (1D tree) / Unet1D (tokamak tree):
    forward order is 1D/model/unet.py:382-426 == tokamak/model/unet.py:359-408."""
    V = _View(P)
    n_res = len(dim_mults)
    x = _conv(x, V["init_conv.weight"], V["init_conv.bias"], nd, padding=3)
    r = x.clone()
    temb = time_mlp(V.sub("time_mlp"), t, dim)
    hs = []
    for i in range(n_res):
        L = V.sub(f"downs.{i}")
        x = resnet_block(L.sub("0"), x, temb, nd, groups)
        hs.append(x)
        x = resnet_block(L.sub("1"), x, temb, nd, groups)
        A = L.sub("2.fn")
        x = linear_attention(A.sub("fn"), prenorm(x, A["norm.g"]), nd, out_norm=outnorm) + x
        hs.append(x)
        x = down(L.sub("3"), x, i == n_res - 1)
    x = resnet_block(V.sub("mid_block1"), x, temb, nd, groups)
    A = V.sub("mid_attn.fn")
    x = full_attention(A.sub("fn"), prenorm(x, A["norm.g"]), nd) + x
    x = resnet_block(V.sub("mid_block2"), x, temb, nd, groups)
    for i in range(n_res):
        L = V.sub(f"ups.{i}")
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(L.sub("0"), x, temb, nd, groups)
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(L.sub("1"), x, temb, nd, groups)
        A = L.sub("2.fn")
        x = linear_attention(A.sub("fn"), prenorm(x, A["norm.g"]), nd, out_norm=outnorm) + x
        x = up(L.sub("3"), x, i == n_res - 1)
    x = torch.cat((x, r), dim=1)
    x = resnet_block(V.sub("final_res_block"), x, temb, nd, groups)
    return _conv(x, V["final_conv.weight"], V["final_conv.bias"], nd)


# ----------------------------------------------------------------------------
# 1D Burgers: Unet2D over the (time=16, space=128) image
# ----------------------------------------------------------------------------

def unet_burgers(P, x, t, *, dim, dim_mults=(1, 2, 4, 8), groups=1):
    """Unet2D.forward, 1D/model/unet.py:382-426.  x (B,3,16,128), t (B,)."""

    def down(V, x, last):
        if last:
            return F.conv2d(x, V["weight"], V["bias"], padding=1)
        # Downsample2d = pixel-unshuffle 'b c (h p1) (w p2) -> b (c p1 p2) h w' + 1x1 conv (:39-43)
        B, C, H, W = x.shape
        x = x.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, C * 4, H // 2, W // 2)
        return F.conv2d(x, V["1.weight"], V["1.bias"])

    def up(V, x, last):
        if last:
            return F.conv2d(x, V["weight"], V["bias"], padding=1)
        x = F.interpolate(x, scale_factor=2, mode="nearest")            # :33-37
        return F.conv2d(x, V["1.weight"], V["1.bias"], padding=1)

    return _unet_lucid(P, x, t, nd=2, dim=dim, dim_mults=dim_mults, groups=groups,
                       prenorm=chan_layernorm, outnorm=chan_layernorm, down=down, up=up)


# ----------------------------------------------------------------------------
# tokamak: Unet1D along time
# ----------------------------------------------------------------------------

def unet_tokamak(P, x, t, *, dim, dim_mults=(1, 2, 4, 8), groups=1):
    """Unet1D.forward, tokamak/model/unet.py:359-408.  x (B,12,128), t (B,)."""

    def down(V, x, last):
        if last:
            return F.conv1d(x, V["weight"], V["bias"], padding=1)
        return F.conv1d(x, V["weight"], V["bias"], stride=2, padding=1)  # k=4 (:30-31)

    def up(V, x, last):
        if last:
            return F.conv1d(x, V["weight"], V["bias"], padding=1)
        x = F.interpolate(x, scale_factor=2, mode="nearest")            # :24-28
        return F.conv1d(x, V["1.weight"], V["1.bias"], padding=1)

    return _unet_lucid(P, x, t, nd=1, dim=dim, dim_mults=dim_mults, groups=groups,
                       prenorm=chan_rmsnorm, outnorm=chan_rmsnorm, down=down, up=up)


# ----------------------------------------------------------------------------
# 2D smoke: Unet3D_with_Conv3D
# ----------------------------------------------------------------------------

def rel_pos_bias(emb_weight, n, num_buckets=32, max_distance=32):
    """T5-style bucketed bias, conv3d.py:74-112.  emb_weight (num_buckets, heads)
    -> (heads, n, n)."""
    q = torch.arange(n, device=emb_weight.device)
    rel = q[None, :] - q[:, None]                       # k_pos - q_pos
    m = -rel
    nb = num_buckets // 2
    ret = (m < 0).long() * nb
    m = m.abs()
    max_exact = nb // 2
    small = m < max_exact
    large = max_exact + (torch.log(m.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    bucket = ret + torch.where(small, m, large)
    return emb_weight[bucket].permute(2, 0, 1)


def rotary(x, freqs):
    """rotary-embedding-torch ``RotaryEmbedding.rotate_queries_or_keys`` restated
    from the package's published algorithm ('lang' freqs, interleaved pairs):
    angle[p, 2j] = angle[p, 2j+1] = p * freqs[j];  out = x*cos + rot_half(x)*sin,
    rot_half: (x0,x1) -> (-x1, x0) on adjacent pairs.  x (..., n, d); the rotated
    width is 2*len(freqs) (= d here).  Call sites: conv3d.py:320-322.
    Third-party, unpinned (requirements.txt) -> "parity unpinned" here."""
    n = x.shape[-2]
    ang = torch.arange(n, dtype=freqs.dtype, device=freqs.device)[:, None] * freqs[None, :]
    ang = ang.repeat_interleave(2, dim=-1)               # (n, rot)
    rot = ang.shape[-1]
    xr, xp = x[..., :rot], x[..., rot:]
    x2 = xr.reshape(*xr.shape[:-1], rot // 2, 2)
    half = torch.stack((-x2[..., 1], x2[..., 0]), dim=-1).reshape(xr.shape)
    out = xr * ang.cos() + half * ang.sin()
    return torch.cat((out, xp), dim=-1)


def token_attention(V, x, heads=4, dim_head=32, freqs=None, bias=None):
    """conv3d.py:277-353 with focus_present_mask all-False.  x (..., n, c)."""
    qkv = F.linear(x, V["to_qkv.weight"]).chunk(3, dim=-1)
    q, k, v = (t.reshape(*t.shape[:-1], heads, dim_head).transpose(-2, -3) for t in qkv)  # (...,h,n,d)
    q = q * dim_head ** -0.5
    if freqs is not None:
        q = rotary(q, freqs)
        k = rotary(k, freqs)
    sim = torch.einsum("...hid,...hjd->...hij", q, k)
    if bias is not None:
        sim = sim + bias
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("...hij,...hjd->...hid", attn, v)
    out = out.transpose(-2, -3).reshape(*x.shape[:-1], heads * dim_head)
    return F.linear(out, V["to_out.weight"])


def _temporal_attn(A, x, freqs, bias):
    """Residual(PreNorm(LN, 'b c f h w -> b (h w) f c' Attention)), conv3d.py:262-275,:383."""
    B, C, Fr, H, W = x.shape
    y = chan_layernorm_div(x, A["norm.gamma"])
    y = y.permute(0, 3, 4, 2, 1).reshape(B, H * W, Fr, C)
    y = token_attention(A.sub("fn.fn"), y, freqs=freqs, bias=bias)
    y = y.reshape(B, H, W, Fr, C).permute(0, 4, 3, 1, 2)
    return y + x


def _spatial_linear_attn(A, x):
    """Residual(PreNorm(LN, SpatialLinearAttention)), conv3d.py:232-258."""
    B, C, Fr, H, W = x.shape
    y = chan_layernorm_div(x, A["norm.gamma"])
    y = y.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
    y = linear_attention(A.sub("fn"), y, 2)
    y = y.reshape(B, Fr, C, H, W).permute(0, 2, 1, 3, 4)
    return y + x


def _spatial_full_attn(A, x):
    """mid: Residual(PreNorm(LN, 'b c f h w -> b f (h w) c' Attention)), conv3d.py:450-452."""
    B, C, Fr, H, W = x.shape
    y = chan_layernorm_div(x, A["norm.gamma"])
    y = y.permute(0, 2, 3, 4, 1).reshape(B, Fr, H * W, C)
    y = token_attention(A.sub("fn.fn"), y)
    y = y.reshape(B, Fr, H, W, C).permute(0, 4, 1, 2, 3)
    return y + x


def unet_smoke(P, x, t, *, dim=64, dim_mults=(1, 2, 4), groups=8):
    """Unet3D_with_Conv3D.forward, conv3d.py:487-574.  x (B,F,7,H,W) frame-major."""
    V = _View(P)
    n_res = len(dim_mults)
    x = x.permute(0, 2, 1, 3, 4)
    Fr = x.shape[2]
    bias = rel_pos_bias(V["time_rel_pos_bias.relative_attention_bias.weight"], Fr)
    freqs = V["init_temporal_attn.fn.fn.fn.rotary_emb.freqs"]
    x = F.conv3d(x, V["init_conv.weight"], V["init_conv.bias"], padding=3)
    x = _temporal_attn(V.sub("init_temporal_attn.fn"), x, freqs, bias)
    r = x.clone()
    temb = time_mlp(V.sub("time_mlp"), t, dim)
    hs = []
    for i in range(n_res):
        L = V.sub(f"downs.{i}")
        x = resnet_block(L.sub("0"), x, temb, 3, groups)
        x = resnet_block(L.sub("1"), x, temb, 3, groups)
        x = _spatial_linear_attn(L.sub("2.fn"), x)
        x = _temporal_attn(L.sub("3.fn"), x, freqs, bias)
        hs.append(x)
        if i < n_res - 1:
            x = F.conv3d(x, L["4.weight"], L["4.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
    x = resnet_block(V.sub("mid_block1"), x, temb, 3, groups)
    x = _spatial_full_attn(V.sub("mid_spatial_attn.fn"), x)
    x = _temporal_attn(V.sub("mid_temporal_attn.fn"), x, freqs, bias)
    x = resnet_block(V.sub("mid_block2"), x, temb, 3, groups)
    for i in range(n_res):
        L = V.sub(f"ups.{i}")
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(L.sub("0"), x, temb, 3, groups)
        x = resnet_block(L.sub("1"), x, temb, 3, groups)
        x = _spatial_linear_attn(L.sub("2.fn"), x)
        x = _temporal_attn(L.sub("3.fn"), x, freqs, bias)
        if i < n_res - 1:
            x = F.conv_transpose3d(x, L["4.weight"], L["4.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
    x = torch.cat((x, r), dim=1)
    x = resnet_block(V.sub("final_conv.0"), x, None, 3, groups)
    x = F.conv3d(x, V["final_conv.1.weight"], V["final_conv.1.bias"])
    return x.permute(0, 2, 1, 3, 4)
