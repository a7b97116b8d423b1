"""Differentiable execution of the drop-in U-Nets: the fine-tuning path.

``net.forward_train(x, t)`` returns eps WITH an autograd graph over the net's ``nn.Parameter``s, so that the reference's
fine-tuning loops -- ``loss = (weight * diffusion(state, mean=False)).mean(); loss.backward(); optimizer.step()``
(1D/inference/inference_ft.py:183-187, tokamak/inference/pipeline.py:238-263, 2d/inference_2d.py:267-279) -- run on the
drop-in classes without a second torch copy of the network.

Every node of the graph is a ``torch.autograd.Function`` whose forward launches the same libsdc_hip.so kernels the
samplers use:
  * convs / Linear / transposed convs (84 % of the FLOPs): forward ``sdc_conv`` (Winograd forms included); data gradient =
    ``sdc_conv`` again with flipped / transposed taps (so it runs on the same MFMA kernels); weight and bias gradient =
    ``sdc_conv_wgrad`` (fp32 MFMA, csrc/sdc_grad.hip);
  * GroupNorm + scale/shift + SiLU (+ residual): ``sdc_gn_stats`` / ``sdc_gn_apply`` forward, ``sdc_gn_silu_bwd`` backward;
  * SiLU / GELU of the time MLP: ``sdc_act`` / ``sdc_act_bwd``;
  * the attention blocks (PreNorm + LinearAttention / temporal / full attention + residual): chains of HIP nodes -- channel
    LayerNorm / RMSNorm (``sdc_chan_norm`` / ``sdc_chan_norm_bwd``), the 1x1 projections on the conv node above, the attention
    cores ``sdc_linattn`` / ``sdc_attn`` (rotary + relative-position bias) with ``sdc_linattn_bwd`` / ``sdc_attn_bwd``.
The only torch arithmetic on the path is glue of negligible size: the residual adds, the gather of the (heads, F, F)
relative-position bias from its 32 x 4 embedding, O(B C) parameter-gradient sums of the GroupNorm row table, and the
element-wise loss.
PyTorch is otherwise plumbing (device memory, the current stream, the autograd tape).  There is no CPU path.
"""
import contextlib
import ctypes as C
import functools
import math

import threading

import torch
from torch.autograd import Function

from . import _lib, grad_ops
from ._lib import SdcConvDesc, check
from .engine import as5, pack_conv_weight

HEADS, DIM_HEAD = 4, 32
HID = HEADS * DIM_HEAD


SPLIT_SMALL_GRIDS = True        # conv_raw: sdc_conv_splitk where it applies (A/B switch for tools/ft_time.py)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


# --------------------------------------------------------------------------------------------------- raw conv launch
def conv_raw(x, wp, bias, cout, k, *, x1=None, stride=(1, 1, 1), pad=(0, 0, 0), up=(1, 1, 1), up_mode=0, out=None, residual=None):
    """one sdc_conv launch on torch's current stream (the eager twin of engine.Plan.conv); 5-D views, any strides"""
    lib = _lib.get_lib()
    B, c0, iD, iH, iW = x.shape
    c1 = 0 if x1 is None else x1.shape[1]

    def osz(i, u, kk, s, p):
        v = (i - 1) * u + 1 if up_mode else i * u
        return (v + 2 * p - kk) // s + 1

    if out is None:
        out = torch.empty((B, cout, *(osz(i, u, kk, s, p) for i, u, kk, s, p in zip((iD, iH, iW), up, k, stride, pad))),
                          dtype=torch.float32, device=x.device)
    nw = k[0] * k[1] * k[2] * (c0 + c1) * cout
    n = wp.numel()
    prec = 0
    if k[2] == 3 and n != nw:
        prec = 2 if n == nw + nw // 3 * 4 else (3 if n == nw + nw // 3 * 4 + nw // 9 * 16 else 4)
        if tuple(k) == (1, 1, 3) and n == nw + nw // 3 * 4 + nw // 3 * 6:
            prec = 5                                                     # 1-D conv packed with its F(4,3) taps
        assert prec != 4 or n == nw + nw // 3 * 4 + nw // 9 * 16 + nw // 27 * 64, (n, nw)
    else:
        assert n == nw, (n, nw, k, c0, c1, cout)
    d = SdcConvDesc()
    d.B, d.Cin0, d.Cin1, d.Cout = B, c0, c1, cout
    d.iD, d.iH, d.iW = iD, iH, iW
    d.oD, d.oH, d.oW = out.shape[2:]
    d.kD, d.kH, d.kW = k
    d.sD, d.sH, d.sW = stride
    d.pD, d.pH, d.pW = pad
    d.uD, d.uH, d.uW = up
    d.up_mode, d.precision = up_mode, prec
    d.x0s[:] = tuple(int(s) for s in x.stride())
    d.x1s[:] = tuple(int(s) for s in x1.stride()) if x1 is not None else (0,) * 5
    d.ys[:] = tuple(int(s) for s in out.stride())
    d.rs[:] = tuple(int(s) for s in residual.stride()) if residual is not None else (0,) * 5
    p = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
    # the fine-tuning step's batch leaves the deep 3x3 convs of the Burgers net on 32-128 workgroups: the input channels are then
    # split over several workgroups per tile (sdc_conv_splitk; the samplers never split -- a sample's rounding would depend on
    # the batch it rides in)
    nsplit = int(lib.sdc_conv_splitk_bytes(C.byref(d))) if (SPLIT_SMALL_GRIDS and residual is None
                                                             and not getattr(_BATCH_INVARIANT, "on", False)) else 0
    if nsplit:
        work = torch.empty(nsplit // 4, dtype=torch.float32, device=x.device)
        check(lib.sdc_conv_splitk(C.byref(d), p(x), p(x1), p(wp), p(bias), p(out), work.data_ptr(), nsplit, _stream(x)), "sdc_conv_splitk")
        return out
    check(lib.sdc_conv(C.byref(d), p(x), p(x1), p(wp), p(bias), p(residual), p(out), _stream(x)), "sdc_conv")
    return out


_BATCH_INVARIANT = threading.local()


@contextlib.contextmanager
def batch_invariant():
    """inside: conv_raw never splits the input channels over workgroups (the split factor depends on the batch).  The samplers'
    differentiable last DDIM step runs forward_train under it, so that a sampled trajectory's bits do not depend on the batch it
    rides in (include/sdc.h: "the samplers use sdc_conv only")."""
    prev = getattr(_BATCH_INVARIANT, "on", False)
    _BATCH_INVARIANT.on = True
    try:
        yield
    finally:
        _BATCH_INVARIANT.on = prev


def _k5(w):
    """kernel size of an nn.Conv weight as (kD, kH, kW)"""
    shp = tuple(w.shape[2:])
    return (1,) * (3 - len(shp)) + shp


def _transposed_422(x, w, bias, cout):
    """ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1) with weight w (Cin, Cout, 1, 4, 4) as four 2x2 sub-pixel convs"""
    B, _, D, H, W = x.shape
    out = torch.empty((B, cout, D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    for ph in (0, 1):
        for pw in (0, 1):
            conv_raw(x, pack_conv_weight(w, ("convT_sub", ph, pw)), bias, cout, (1, 2, 2), pad=(0, 1 - ph, 1 - pw),
                     out=out[:, :, :, ph::2, pw::2])
    return out


class _ArenaSlot(threading.local):
    """the PackArena of the net whose training forward is being recorded (Trainer.step) -- per thread: two nets trained from two
    host threads must not see each other's arena (the backward of a recorded graph carries its arena in ctx, not in this slot)"""
    cur = None


_ARENA = _ArenaSlot()

# The weight gradient and the data gradient of a conv both start from gy and meet only in autograd's accumulation: on the small
# grids of the 1-D nets neither fills the chip (a C3 layer is 12 800 positions = 100 workgroups for 256 CUs), so the backward of
# a conv node forks -- weight gradient on a side stream, data gradient on the node's stream -- and joins before it returns.
# Under stream capture (train_graph.GraphedLossStep) the fork / join becomes two branches of the hipGraph.
OVERLAP_WGRAD = True


class _SideSlot(threading.local):
    streams = None


_SIDE = _SideSlot()


def _side_stream(dev):
    if _SIDE.streams is None:
        _SIDE.streams = {}
    s = _SIDE.streams.get(dev.index)
    if s is None:
        s = _SIDE.streams[dev.index] = torch.cuda.Stream(dev)
    return s


class ConvFn(Function):
    """y = conv(x [| x1], w) + b for every conv form of the three U-Nets.  cfg = (kind, stride, pad, up, precision):
       kind 'conv'      nn.Conv1d/2d/3d / nn.Linear (1x1) with stride / pad; up = nearest upsampling of x folded into the read
                        (nn.Upsample + conv, 1D/model/unet.py:33-37, tokamak/model/unet.py:24-28)
       kind 'convT422'  nn.ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1) (conv3d.py:159-160)
       kind 'unshuffle' pixel-unshuffle + 1x1 conv = 2x2 stride-2 conv (Downsample2d, 1D/model/unet.py:39-43)"""

    @staticmethod
    def forward(ctx, x, x1, w, b, cfg):
        kind, stride, pad, up, prec = cfg
        ctx.cfg = cfg
        ctx.save_for_backward(x, x1, w)
        ctx.has_bias = b is not None
        bb = None if b is None else b.detach().contiguous()
        wd = w.detach()
        ctx.arena = arena = _ARENA.cur if (_ARENA.cur is not None and grad_ops.PackArena.cacheable(w)) else None
        if kind == "conv":
            return conv_raw(x, grad_ops.pack_conv_weight(wd, prec, arena=arena), bb, w.shape[0], _k5(w), x1=x1, stride=stride, pad=pad, up=up)
        if kind == "convT422":
            return _transposed_422(x, wd, bb, w.shape[1])
        if kind == "unshuffle":
            return conv_raw(x, pack_conv_weight(wd, "unshuffle"), bb, w.shape[0], (1, 2, 2), stride=(1, 2, 2))
        raise ValueError(kind)

    @staticmethod
    def backward(ctx, gy):
        x, x1, w = ctx.saved_tensors
        kind, stride, pad, up, prec = ctx.cfg
        need_x, need_x1, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        wd = w.detach()
        gx = gx1 = gw = gb = None
        k = _k5(w)
        c0 = x.shape[1]
        if kind == "conv":
            fork = None
            if need_w or ctx.has_bias:
                if OVERLAP_WGRAD and (need_x or need_x1):
                    main = torch.cuda.current_stream(gy.device)
                    fork = _side_stream(gy.device)
                    fork.wait_stream(main)
                with torch.cuda.stream(fork) if fork is not None else contextlib.nullcontext():
                    gw, gb = grad_ops.conv_wgrad(gy, x, k, stride, pad, up, bias=ctx.has_bias)
                    if x1 is not None:
                        gw1, _ = grad_ops.conv_wgrad(gy, x1, k, stride, pad, up, bias=False)
                        gw = torch.cat((gw, gw1), dim=1)
                    gw = gw.reshape(w.shape)
            if need_x or need_x1:
                w5 = as5(wd)
                if stride == (1, 1, 1):
                    # correlation with the flipped, transposed taps; pad k - 1 - p restores the input size
                    ga = conv_raw(gy, grad_ops.pack_conv_weight(w5, prec, flip=True, arena=ctx.arena), None, w5.shape[1], k,
                                  pad=tuple(kk - 1 - p for kk, p in zip(k, pad)))
                    if up != (1, 1, 1):
                        ga = grad_ops.sumpool2(ga, up[1], up[2])       # VJP of the folded nearest upsampling
                elif k == (1, 4, 4) and stride == (1, 2, 2) and pad == (0, 1, 1):
                    ga = _transposed_422(gy, w5, None, w5.shape[1])    # the conv weight (Cout, Cin, k) read as a ConvTranspose weight
                else:
                    # generic strided conv: conv over the zero-stuffed gradient with flipped taps (engine kind 'convT')
                    ga = conv_raw(gy, pack_conv_weight(w5, "convT"), None, w5.shape[1], k, up=stride, up_mode=1,
                                  pad=tuple(kk - 1 - p for kk, p in zip(k, pad)),
                                  out=torch.empty((gy.shape[0], w5.shape[1], *x.shape[2:]), dtype=torch.float32, device=gy.device))
                gx = ga[:, :c0]
                gx1 = ga[:, c0:] if x1 is not None else None
            if fork is not None:
                main.wait_stream(fork)                     # join: gy / x stay referenced by this frame until here
                for g_ in (gw, gb):
                    if g_ is not None:
                        g_.record_stream(main)             # allocated on the side stream, consumed (accumulated) on the node's
        elif kind == "convT422":
            # y[2i - 1 + k] += x[i] w[k]:  dw = wgrad(G = x, X = gy);  dx[i] = sum_k gy[2i - 1 + k] w[k] (a stride-2 conv, no flip)
            if need_w or ctx.has_bias:
                gw, _ = grad_ops.conv_wgrad(x, gy, (1, 4, 4), (1, 2, 2), (0, 1, 1), bias=False)
                gw = gw.reshape(w.shape)
                if ctx.has_bias:
                    gb = gy.sum((0, 2, 3, 4))
            if need_x:
                gx = conv_raw(gy, pack_conv_weight(wd, "conv"), None, w.shape[0], (1, 4, 4), stride=(1, 2, 2), pad=(0, 1, 1))
        elif kind == "unshuffle":
            if need_w or ctx.has_bias:
                gw, gb = grad_ops.conv_wgrad(gy, x, (1, 2, 2), (1, 2, 2), (0, 0, 0), bias=ctx.has_bias)
                gw = gw.reshape(w.shape)                      # (co, c, 1, 2, 2) -> (co, c*4, 1, 1): the (c, p1, p2) order of the unshuffle
            if need_x:
                co, c4 = w.shape[0], w.shape[1]
                w4 = wd.reshape(co, c4 // 4, 2, 2)
                gx = torch.empty_like(x)
                for p1 in (0, 1):
                    for p2 in (0, 1):
                        wt = w4[:, :, p1, p2].t().reshape(c4 // 4, co, 1, 1, 1).contiguous()
                        conv_raw(gy, pack_conv_weight(wt, "conv"), None, c4 // 4, (1, 1, 1), out=gx[:, :, :, p1::2, p2::2])
        return gx, gx1, gw, gb, None


def _linear_ok(x, w):
    """sdc_linear's shapes: nn.Linear on (B, K[, 1, 1, 1]) rows, K and M multiples of 4 (every MLP of the three U-Nets at dim >= 4)"""
    return (x.dim() in (2, 5) and all(s == 1 for s in x.shape[2:]) and all(s == 1 for s in w.shape[2:])
            and w.shape[1] % 4 == 0 and w.shape[0] % 4 == 0 and x.shape[1] == w.shape[1]
            # sdc_linear reads float4s: a misaligned view (a slice of a bigger tensor) takes the 1x1x1 conv node instead
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


class LinearFn(Function):
    """y = x W^T + b on a batch of rows: the time MLP and the ResnetBlocks' scale/shift MLPs (1D/model/unet.py:300-305, :158-162;
    conv3d.py:212-216, :399-404) on sdc_linear / sdc_linear_dgrad / sdc_linear_wgrad -- kernels shaped by the weight matrix
    (16 of its rows per workgroup) instead of the conv kernels' position tiles, which see 64 positions here.  The parameter is read
    where it lives: nothing is packed."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.get_lib()
        B, K, M = x.shape[0], w.shape[1], w.shape[0]
        x2 = x.detach().reshape(B, K).contiguous()
        w2 = w.detach().reshape(M, K).contiguous()
        y = torch.empty((B, M), dtype=torch.float32, device=x.device)
        check(lib.sdc_linear(x2.data_ptr(), w2.data_ptr(), 0 if b is None else b.detach().contiguous().data_ptr(), y.data_ptr(),
                             B, K, M, K, M, _stream(x)), "sdc_linear")
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        return y.reshape(B, M, *x.shape[2:])

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.get_lib()
        x2, w = ctx.saved_tensors
        B, K = x2.shape
        M = w.shape[0]
        g2 = gy.reshape(B, M).contiguous()
        gx = gw = gb = None
        fork = None
        if ctx.needs_input_grad[1] or ctx.has_bias:
            if OVERLAP_WGRAD and ctx.needs_input_grad[0]:
                main = torch.cuda.current_stream(gy.device)
                fork = _side_stream(gy.device)
                fork.wait_stream(main)
            with torch.cuda.stream(fork) if fork is not None else contextlib.nullcontext():
                gw = torch.empty((M, K), dtype=torch.float32, device=gy.device)
                gb = torch.empty(M, dtype=torch.float32, device=gy.device) if ctx.has_bias else None
                check(lib.sdc_linear_wgrad(g2.data_ptr(), x2.data_ptr(), gw.data_ptr(), 0 if gb is None else gb.data_ptr(), B, K, M, M, K,
                                           _stream(gy)), "sdc_linear_wgrad")
                gw = gw.reshape(w.shape)
        if ctx.needs_input_grad[0]:
            w2 = w.detach().reshape(M, K).contiguous()
            gx = torch.empty((B, K), dtype=torch.float32, device=gy.device)
            check(lib.sdc_linear_dgrad(g2.data_ptr(), w2.data_ptr(), gx.data_ptr(), B, K, M, M, K, _stream(gy)), "sdc_linear_dgrad")
            gx = gx.reshape(ctx.xshape)
        if fork is not None:
            main.wait_stream(fork)
            for g_ in (gw, gb):
                if g_ is not None:
                    g_.record_stream(main)
        return gx, gw, gb


def linear(x, w, b):
    """nn.Linear node: LinearFn where its kernels take the shape, the 1x1x1 conv node otherwise (any shape)"""
    if _linear_ok(x, w):
        return LinearFn.apply(x, w, b)
    return ConvFn.apply(x, None, w, b, ("conv", (1, 1, 1), (0, 0, 0), (1, 1, 1), 0))


class GNSiLUFn(Function):
    """SiLU(GroupNorm(h) (scale + 1) + shift) (+ residual) -- Block, conv3d.py:189-204 / 1D/model/unet.py:128-147"""

    @staticmethod
    def forward(ctx, h, gamma, beta, ss, residual, groups):
        h = h.contiguous()
        st = grad_ops.gn_stats(h, groups)
        g_, b_ = gamma.detach().contiguous(), beta.detach().contiguous()
        ssd = None if ss is None else ss.detach().contiguous()
        ctx.save_for_backward(h, st, g_, b_, ssd)
        ctx.groups, ctx.has_res = groups, residual is not None
        return grad_ops.gn_apply(h, st, g_, b_, groups, ssd, None if residual is None else residual.contiguous())

    @staticmethod
    def backward(ctx, gy):
        h, st, g_, b_, ssd = ctx.saved_tensors
        gh, dg, db, dss = grad_ops.gn_silu_bwd(h, gy, st, g_, b_, ctx.groups, ssd)
        return gh, dg, db, dss, (gy if ctx.has_res else None), None


class ActFn(Function):
    """kind 0 SiLU, 1 GELU (exact erf) -- time_mlp (1D/model/unet.py:300-305, conv3d.py:391-396)"""

    @staticmethod
    def forward(ctx, x, kind):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.kind = kind
        y = torch.empty_like(x)
        check(_lib.get_lib().sdc_act(x.data_ptr(), y.data_ptr(), x.numel(), kind, _stream(x)), "sdc_act")
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return grad_ops.act_bwd(x, gy, ctx.kind), None


class ChanNormFn(Function):
    """channel LayerNorm (mode 0, gain only) / RMSNorm (mode 1): 1D/model/unet.py:53-63, tokamak/model/unet.py:45-51, conv3d.py:165-174"""

    @staticmethod
    def forward(ctx, x, g, mode):
        x = x.contiguous()
        ctx.save_for_backward(x, g.detach())
        ctx.mode, ctx.gshape = mode, g.shape
        return grad_ops.chan_norm(x, g.detach(), mode)

    @staticmethod
    def backward(ctx, gy):
        x, g = ctx.saved_tensors
        gx, dg = grad_ops.chan_norm_bwd(x, gy, g, ctx.mode)
        return gx, dg.reshape(ctx.gshape), None


class AttnCoreFn(Function):
    """softmax attention core over channel-major qkv (sdc_attn / sdc_attn_bwd); geom = (heads, outer, inner, ntok, q strides,
    out strides, out shape); rot = [ntok][16][2] cos / sin table (no gradient), bias = (heads, ntok, ntok) or None"""

    @staticmethod
    def forward(ctx, qkv, bias, rot, geom):
        heads, outer, inner, ntok, qs, os_, oshape = geom
        qkv = qkv.contiguous()
        bd = None if bias is None else bias.detach().contiguous()
        out = torch.empty(oshape, dtype=torch.float32, device=qkv.device)
        grad_ops.attn_core(qkv, out, heads, outer, inner, ntok, qs, os_, rot, bd)
        ctx.save_for_backward(qkv, bd, rot)
        ctx.geom = geom
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, bd, rot = ctx.saved_tensors
        heads, outer, inner, ntok, qs, os_, _ = ctx.geom
        dqkv, dbias = grad_ops.attn_core_bwd(qkv, gout.contiguous(), heads, outer, inner, ntok, qs, os_, rot, bd)
        return dqkv, dbias, None, None


class LinAttnCoreFn(Function):
    """linear attention core (sdc_linattn / sdc_linattn_bwd); geom = (heads, outer, inner, n, q strides, out strides, out shape)"""

    @staticmethod
    def forward(ctx, qkv, geom):
        heads, outer, inner, n, qs, os_, oshape = geom
        qkv = qkv.contiguous()
        out = torch.empty(oshape, dtype=torch.float32, device=qkv.device)
        grad_ops.linattn_core(qkv, out, heads, outer, inner, n, qs, os_)
        ctx.save_for_backward(qkv)
        ctx.geom = geom
        return out

    @staticmethod
    def backward(ctx, gout):
        (qkv,) = ctx.saved_tensors
        heads, outer, inner, n, qs, os_, _ = ctx.geom
        return grad_ops.linattn_core_bwd(qkv, gout.contiguous(), heads, outer, inner, n, qs, os_), None


# --------------------------------------------------------------------------------------------------- helpers
@functools.lru_cache(maxsize=8)
def _relpos_buckets(n, device, num_buckets=32, max_distance=32):
    """RelativePositionBias bucket indices, conv3d.py:74-112 (integer arithmetic, no gradient; cached per (n, device): the
    host-to-device copy of the table would otherwise drain the stream once per training forward)"""
    q = torch.arange(n)
    m = -(q[None, :] - q[:, None])
    nb = num_buckets // 2
    ret = (m < 0).long() * nb
    m = m.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(m.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (ret + torch.where(m < max_exact, m, large)).to(device)


# --------------------------------------------------------------------------------------------------- the differentiable forward
class Trainer:
    """Walks a drop-in U-Net in the reference's forward order and records an autograd graph of HIP-backed nodes."""

    def __init__(self, net):
        self.net = net
        self.prec = net.precision
        self.arena = grad_ops.PackArena()

    @contextlib.contextmanager
    def step(self, device):
        """one training forward: every packed conv weight refreshed by one launch, conv nodes bound to this net's arena"""
        self.arena.begin(device)
        prev, _ARENA.cur = _ARENA.cur, self.arena
        try:
            yield
        finally:
            _ARENA.cur = prev

    def P(self, key):
        return self.net.P(key)

    # ---- nodes
    def conv(self, prefix, x, x1=None, kind="conv", stride=(1, 1, 1), pad=None, up=(1, 1, 1), bias=True):
        w = self.P(f"{prefix}.weight")
        b = self.P(f"{prefix}.bias") if bias else None
        if pad is None:
            pad = tuple(kk // 2 for kk in _k5(w))
        return ConvFn.apply(x, x1, w, b, (kind, tuple(stride), tuple(pad), tuple(up), self.prec))

    def gn(self, prefix, h, ss=None, residual=None):
        return GNSiLUFn.apply(h, self.P(f"{prefix}.weight"), self.P(f"{prefix}.bias"), ss, residual, self.net.groups)

    def resnet(self, prefix, x, cond, x1=None):
        """ResnetBlock: conv3d.py:206-230 / 1D/model/unet.py:149-180"""
        ss = None
        if cond is not None and self.net.has(f"{prefix}.mlp.1.weight"):
            e = linear(cond, self.P(f"{prefix}.mlp.1.weight"), self.P(f"{prefix}.mlp.1.bias"))
            ss = e.reshape(e.shape[0], e.shape[1])
        h = self.gn(f"{prefix}.block1.norm", self.conv(f"{prefix}.block1.proj", x, x1), ss)
        g = self.conv(f"{prefix}.block2.proj", h)
        if self.net.has(f"{prefix}.res_conv.weight"):
            r = self.conv(f"{prefix}.res_conv", x, x1)
        else:
            r = x
        return self.gn(f"{prefix}.block2.norm", g, None, r)

    def time_cond(self, t):
        """SinusoidalPosEmb -> Linear -> GELU -> Linear, then the SiLU every block MLP starts with"""
        from .unet import _sinusoid
        dim = self.net.dim
        emb = _sinusoid(t.detach().to(self.net.device()).float(), dim).reshape(t.shape[0], dim, 1, 1, 1).contiguous()
        h = linear(emb, self.P("time_mlp.1.weight").reshape(4 * dim, dim, 1, 1, 1), self.P("time_mlp.1.bias"))
        h = ActFn.apply(h, 1)
        h = linear(h, self.P("time_mlp.3.weight").reshape(4 * dim, 4 * dim, 1, 1, 1), self.P("time_mlp.3.bias"))
        return ActFn.apply(h, 0)

    # ---- attention blocks as chains of HIP nodes (norm -> 1x1 conv -> core -> 1x1 conv [-> norm] + x)
    def _pw(self, key, x, bias_key=None):
        w = self.P(key)
        w5 = w.reshape(w.shape[0], w.shape[1], 1, 1, 1)
        return ConvFn.apply(x, None, w5, None if bias_key is None else self.P(bias_key), ("conv", (1, 1, 1), (0, 0, 0), (1, 1, 1), 0))

    def linattn_lucid(self, p, x, mode):
        """Residual(PreNorm(LinearAttention)): 1D/model/unet.py:182-222 (LayerNorm), tokamak/model/unet.py:182-222 (RMSNorm)"""
        B, Cc = x.shape[:2]
        n = x.numel() // (B * Cc)
        xn = ChanNormFn.apply(x, self.P(f"{p}.fn.norm.g"), mode)
        qkv = self._pw(f"{p}.fn.fn.to_qkv.weight", xn)
        o = LinAttnCoreFn.apply(qkv, (HEADS, B, 1, n, (3 * HID * n, n, 0), (HID * n, n, 0), (B, HID, *x.shape[2:])))
        y = self._pw(f"{p}.fn.fn.to_out.0.weight", o, f"{p}.fn.fn.to_out.0.bias")
        return ChanNormFn.apply(y, self.P(f"{p}.fn.fn.to_out.1.g"), mode) + x

    def fullattn_lucid(self, p, x, mode):
        """Residual(PreNorm(Attention)): 1D/model/unet.py:224-258"""
        B, Cc = x.shape[:2]
        n = x.numel() // (B * Cc)
        xn = ChanNormFn.apply(x, self.P(f"{p}.fn.norm.g"), mode)
        qkv = self._pw(f"{p}.fn.fn.to_qkv.weight", xn)
        o = AttnCoreFn.apply(qkv, None, None, (HEADS, B, 1, n, (3 * HID * n, n, 0, 1), (HID * n, n, 0, 1), (B, HID, *x.shape[2:])))
        return self._pw(f"{p}.fn.fn.to_out.weight", o, f"{p}.fn.fn.to_out.bias") + x

    def spatial_linear(self, p, x):
        """Residual(PreNorm(SpatialLinearAttention)), per frame: conv3d.py:232-258"""
        B, Cc, Fr, H, W = x.shape
        hw = H * W
        xn = ChanNormFn.apply(x, self.P(f"{p}.fn.norm.gamma"), 0)
        qkv = self._pw(f"{p}.fn.fn.to_qkv.weight", xn)
        o = LinAttnCoreFn.apply(qkv, (HEADS, B, Fr, hw, (3 * HID * Fr * hw, Fr * hw, hw), (HID * Fr * hw, Fr * hw, hw), (B, HID, Fr, H, W)))
        return self._pw(f"{p}.fn.fn.to_out.weight", o, f"{p}.fn.fn.to_out.bias") + x

    def _temporal_tables(self, Fr, dev):
        """rotary cos / sin table (no gradient: rotary-embedding-torch keeps freqs with requires_grad = False unless learned_freq)
        and the relative-position bias (heads, F, F) gathered from the embedding (gradient through the gather)"""
        fr = self.P("init_temporal_attn.fn.fn.fn.rotary_emb.freqs").detach().float()
        ang = torch.arange(Fr, dtype=torch.float32, device=dev)[:, None] * fr[None, :]
        rot = torch.stack((ang.cos(), ang.sin()), dim=-1).reshape(-1).contiguous()
        bias = self.P("time_rel_pos_bias.relative_attention_bias.weight")[_relpos_buckets(Fr, dev)].permute(2, 0, 1).contiguous()
        return rot, bias

    def temporal(self, p, x, tables):
        """Residual(PreNorm('b c f h w -> b (h w) f c' Attention)) with rotary + relative position bias: conv3d.py:262-353"""
        B, Cc, Fr, H, W = x.shape
        hw = H * W
        rot, bias = tables
        xn = ChanNormFn.apply(x, self.P(f"{p}.fn.norm.gamma"), 0)
        qkv = self._pw(f"{p}.fn.fn.fn.to_qkv.weight", xn)
        o = AttnCoreFn.apply(qkv, bias, rot, (HEADS, B, hw, Fr, (3 * HID * Fr * hw, Fr * hw, 1, hw), (HID * Fr * hw, Fr * hw, 1, hw),
                                              (B, HID, Fr, H, W)))
        return self._pw(f"{p}.fn.fn.fn.to_out.weight", o) + x

    def spatial_full(self, p, x):
        """mid: Residual(PreNorm('b c f h w -> b f (h w) c' Attention)): conv3d.py:450-452"""
        B, Cc, Fr, H, W = x.shape
        hw = H * W
        xn = ChanNormFn.apply(x, self.P(f"{p}.fn.norm.gamma"), 0)
        qkv = self._pw(f"{p}.fn.fn.fn.to_qkv.weight", xn)
        o = AttnCoreFn.apply(qkv, None, None, (HEADS, B, Fr, hw, (3 * HID * Fr * hw, Fr * hw, hw, 1), (HID * Fr * hw, Fr * hw, hw, 1),
                                               (B, HID, Fr, H, W)))
        return self._pw(f"{p}.fn.fn.fn.to_out.weight", o) + x


def _w5(p):
    return p.reshape(*p.shape[:2], *([1] * (5 - p.dim()))) if p.dim() < 5 else p


def forward_train_lucid(net, x, t):
    """Unet2D / Unet1D: 1D/model/unet.py:382-426 == tokamak/model/unet.py:359-408"""
    T = net._trainer()
    mode, nd = net.NORM_MODE, net.ND
    nres = len(net.dim_mults)
    x5 = as5(x)
    cond = T.time_cond(t)
    h = T.conv("init_conv", x5)
    r = h
    hs = []
    for i in range(nres):
        p = f"downs.{i}"
        h = T.resnet(f"{p}.0", h, cond)
        hs.append(h)
        h = T.resnet(f"{p}.1", h, cond)
        h = T.linattn_lucid(f"{p}.2", h, mode)
        hs.append(h)
        last = i == nres - 1
        if last:
            h = T.conv(f"{p}.3", h)
        elif nd == 2:
            h = T.conv(f"{p}.3.1", h, kind="unshuffle")
        else:
            h = T.conv(f"{p}.3", h, stride=(1, 1, 2), pad=(0, 0, 1))
    h = T.resnet("mid_block1", h, cond)
    h = T.fullattn_lucid("mid_attn", h, mode)
    h = T.resnet("mid_block2", h, cond)
    for i in range(nres):
        p = f"ups.{i}"
        h = T.resnet(f"{p}.0", h, cond, x1=hs.pop())
        h = T.resnet(f"{p}.1", h, cond, x1=hs.pop())
        h = T.linattn_lucid(f"{p}.2", h, mode)
        last = i == nres - 1
        if last:
            h = T.conv(f"{p}.3", h)
        else:
            h = T.conv(f"{p}.3.1", h, up=(1, 2, 2) if nd == 2 else (1, 1, 2))
    h = T.resnet("final_res_block", h, cond, x1=r)
    out = T.conv("final_conv", h)
    return out.reshape(x.shape[0], -1, *x.shape[2:])


def forward_train_smoke(net, x, t):
    """Unet3D_with_Conv3D: conv3d.py:487-574; x (B, F, C, H, W) frame-major"""
    T = net._trainer()
    nres = len(net.dim_mults)
    x5 = x.permute(0, 2, 1, 3, 4)
    cond = T.time_cond(t)

    tables = T._temporal_tables(x5.shape[2], x.device)
    temporal = lambda pre, h: T.temporal(pre, h, tables)        # noqa: E731
    spatial = T.spatial_linear

    h = T.conv("init_conv", x5)
    h = temporal("init_temporal_attn", h)
    r = h
    hs = []
    for i in range(nres):
        p = f"downs.{i}"
        h = T.resnet(f"{p}.0", h, cond)
        h = T.resnet(f"{p}.1", h, cond)
        h = spatial(f"{p}.2", h)
        h = temporal(f"{p}.3", h)
        hs.append(h)
        if i < nres - 1:
            h = T.conv(f"{p}.4", h, stride=(1, 2, 2), pad=(0, 1, 1))
    h = T.resnet("mid_block1", h, cond)
    h = T.spatial_full("mid_spatial_attn", h)
    h = temporal("mid_temporal_attn", h)
    h = T.resnet("mid_block2", h, cond)
    for i in range(nres):
        p = f"ups.{i}"
        h = T.resnet(f"{p}.0", h, cond, x1=hs.pop())
        h = T.resnet(f"{p}.1", h, cond)
        h = spatial(f"{p}.2", h)
        h = temporal(f"{p}.3", h)
        if i < nres - 1:
            h = T.conv(f"{p}.4", h, kind="convT422")
    h = T.resnet("final_conv.0", h, None, x1=r)
    out = T.conv("final_conv.1", h)
    return out.permute(0, 2, 1, 3, 4)
