"""Drop-in denoisers: Unet2D (1D Burgers), Unet1D (tokamak), Unet3D_with_Conv3D (2D smoke).

Same constructor arguments, same ``state_dict`` key names and the same
``forward(x, time)`` contract as the reference classes
  Unet2D               1D/model/unet.py:263-426
  Unet1D               tokamak/model/unet.py:263-408
  Unet3D_with_Conv3D   2d/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:357-574
but ``forward`` runs a pre-bound list of libsdc_hip.so kernels (engine.Plan); there
is no torch.nn op and no CPU path.  Weights live in ordinary ``nn.Parameter``s, so
``load_state_dict`` of a reference checkpoint works unchanged; they are repacked
into kernel layouts when a plan is built (``refresh()`` after weights change).
"""
import ctypes as C
import functools
import math
import os

import torch
from torch import nn

from ._lib import check, get_lib
from .engine import Plan, as5

HEADS, DIM_HEAD = 4, 32
HID = HEADS * DIM_HEAD


# --------------------------------------------------------------------------- parameter specs
def _resnet_spec(p, cin, cout, time_dim, k):
    s = []
    if time_dim is not None:
        s += [(f"{p}.mlp.1.weight", (cout * 2, time_dim)), (f"{p}.mlp.1.bias", (cout * 2,))]
    for blk, ci in (("block1", cin), ("block2", cout)):
        s += [(f"{p}.{blk}.proj.weight", (cout, ci, *k)), (f"{p}.{blk}.proj.bias", (cout,)),
              (f"{p}.{blk}.norm.weight", (cout,)), (f"{p}.{blk}.norm.bias", (cout,))]
    if cin != cout:
        s += [(f"{p}.res_conv.weight", (cout, cin, *([1] * len(k)))), (f"{p}.res_conv.bias", (cout,))]
    return s


def spec_lucid(dim, dim_mults, channels, nd):
    """state_dict layout of Unet2D (nd=2) / Unet1D (nd=1), in the reference's registration order."""
    one, k3, k7 = [1] * nd, [3] * nd, [7] * nd
    g = (1, 0, *one)                     # gain shape (1, C, 1[,1]) with C filled below
    td = dim * 4
    dims = [dim] + [dim * m for m in dim_mults]
    io = list(zip(dims[:-1], dims[1:]))
    gshape = lambda c: (1, c, *one)  # noqa: E731
    s = []
    if nd == 1:
        s += [("init_conv.weight", (dim, channels, *k7)), ("init_conv.bias", (dim,))]
    s += [("time_mlp.1.weight", (td, dim)), ("time_mlp.1.bias", (td,)),
          ("time_mlp.3.weight", (td, td)), ("time_mlp.3.bias", (td,))]
    if nd == 2:
        s += [("init_conv.weight", (dim, channels, *k7)), ("init_conv.bias", (dim,))]

    def lin_attn(p, c):
        return [(f"{p}.fn.fn.to_qkv.weight", (HID * 3, c, *one)), (f"{p}.fn.fn.to_out.0.weight", (c, HID, *one)),
                (f"{p}.fn.fn.to_out.0.bias", (c,)), (f"{p}.fn.fn.to_out.1.g", gshape(c)), (f"{p}.fn.norm.g", gshape(c))]

    def full_attn(p, c):
        return [(f"{p}.fn.fn.to_qkv.weight", (HID * 3, c, *one)), (f"{p}.fn.fn.to_out.weight", (c, HID, *one)),
                (f"{p}.fn.fn.to_out.bias", (c,)), (f"{p}.fn.norm.g", gshape(c))]

    downs, ups = [], []
    for i, (ci, co) in enumerate(io):
        last = i == len(io) - 1
        p = f"downs.{i}"
        downs += _resnet_spec(f"{p}.0", ci, ci, td, k3) + _resnet_spec(f"{p}.1", ci, ci, td, k3) + lin_attn(f"{p}.2", ci)
        if last:
            downs += [(f"{p}.3.weight", (co, ci, *k3)), (f"{p}.3.bias", (co,))]
        elif nd == 2:
            downs += [(f"{p}.3.1.weight", (co, ci * 4, *one)), (f"{p}.3.1.bias", (co,))]
        else:
            downs += [(f"{p}.3.weight", (co, ci, 4)), (f"{p}.3.bias", (co,))]
    mid = dims[-1]
    mids = _resnet_spec("mid_block1", mid, mid, td, k3) + full_attn("mid_attn", mid) + _resnet_spec("mid_block2", mid, mid, td, k3)
    for i, (ci, co) in enumerate(reversed(io)):
        last = i == len(io) - 1
        p = f"ups.{i}"
        ups += _resnet_spec(f"{p}.0", co + ci, co, td, k3) + _resnet_spec(f"{p}.1", co + ci, co, td, k3) + lin_attn(f"{p}.2", co)
        if last:
            ups += [(f"{p}.3.weight", (ci, co, *k3)), (f"{p}.3.bias", (ci,))]
        else:
            ups += [(f"{p}.3.1.weight", (ci, co, *k3)), (f"{p}.3.1.bias", (ci,))]
    fin = _resnet_spec("final_res_block", dim * 2, dim, td, k3) + [("final_conv.weight", (channels, dim, *one)),
                                                                   ("final_conv.bias", (channels,))]
    if nd == 2:
        return s + downs + mids + ups + fin
    # Unet1D registers downs, ups (both ModuleLists created first), then mid blocks
    return s + downs + ups + mids + fin


def spec_smoke(dim, dim_mults, channels):
    """state_dict layout of Unet3D_with_Conv3D(dim, dim_mults, channels)."""
    td = dim * 4
    dims = [dim] + [dim * m for m in dim_mults]
    io = list(zip(dims[:-1], dims[1:]))
    k3, one = (3, 3, 3), (1, 1, 1)
    rot = ("rotary_emb.freqs", (DIM_HEAD // 2,))

    def t_attn(p, c):      # Residual(PreNorm(EinopsToAndFrom(Attention)))
        return [(f"{p}.fn.fn.fn.{rot[0]}", rot[1]), (f"{p}.fn.fn.fn.to_qkv.weight", (HID * 3, c)),
                (f"{p}.fn.fn.fn.to_out.weight", (c, HID)), (f"{p}.fn.norm.gamma", (1, c, 1, 1, 1))]

    def s_lin(p, c):       # Residual(PreNorm(SpatialLinearAttention))
        return [(f"{p}.fn.fn.to_qkv.weight", (HID * 3, c, 1, 1)), (f"{p}.fn.fn.to_out.weight", (c, HID, 1, 1)),
                (f"{p}.fn.fn.to_out.bias", (c,)), (f"{p}.fn.norm.gamma", (1, c, 1, 1, 1))]

    s = [("time_rel_pos_bias.relative_attention_bias.weight", (32, HEADS)),
         ("init_conv.weight", (dim, channels, 7, 7, 7)), ("init_conv.bias", (dim,))]
    s += t_attn("init_temporal_attn", dim)
    s += [("time_mlp.1.weight", (td, dim)), ("time_mlp.1.bias", (td,)), ("time_mlp.3.weight", (td, td)),
          ("time_mlp.3.bias", (td,))]
    downs, ups = [], []
    for i, (ci, co) in enumerate(io):
        p = f"downs.{i}"
        downs += _resnet_spec(f"{p}.0", ci, co, td, k3) + _resnet_spec(f"{p}.1", co, co, td, k3) + s_lin(f"{p}.2", co) + t_attn(f"{p}.3", co)
        if i < len(io) - 1:
            downs += [(f"{p}.4.weight", (co, co, 1, 4, 4)), (f"{p}.4.bias", (co,))]
    for i, (ci, co) in enumerate(reversed(io)):
        p = f"ups.{i}"
        ups += _resnet_spec(f"{p}.0", co * 2, ci, td, k3) + _resnet_spec(f"{p}.1", ci, ci, td, k3) + s_lin(f"{p}.2", ci) + t_attn(f"{p}.3", ci)
        if i < len(io) - 1:
            ups += [(f"{p}.4.weight", (ci, ci, 1, 4, 4)), (f"{p}.4.bias", (ci,))]
    mid = dims[-1]
    mids = _resnet_spec("mid_block1", mid, mid, td, k3)
    mids += [("mid_spatial_attn.fn.fn.fn.to_qkv.weight", (HID * 3, mid)), ("mid_spatial_attn.fn.fn.fn.to_out.weight", (mid, HID)),
             ("mid_spatial_attn.fn.norm.gamma", (1, mid, 1, 1, 1))]
    mids += t_attn("mid_temporal_attn", mid) + _resnet_spec("mid_block2", mid, mid, td, k3)
    fin = _resnet_spec("final_conv.0", dim * 2, dim, None, k3) + [("final_conv.1.weight", (channels, dim, *one)),
                                                                 ("final_conv.1.bias", (channels,))]
    return s + downs + ups + mids + fin


# --------------------------------------------------------------------------- nn.Module shell
def _register(root, key, value):
    *mods, leaf = key.split(".")
    m = root
    for name in mods:
        if name not in m._modules:
            m.add_module(name, nn.Module())
        m = m._modules[name]
    m.register_parameter(leaf, nn.Parameter(value))


def _default_init(spec, gen):
    """torch.nn default-style init (uniform +-1/sqrt(fan_in); gains 1, norm biases 0)."""
    out = {}
    for key, shape in spec:
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("rotary_emb.freqs"):
            d = 2 * shape[0]
            v = 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))
        elif key.endswith("relative_attention_bias.weight"):
            v = torch.randn(shape, generator=gen)
        elif leaf in ("g", "gamma") or key.endswith("norm.weight"):
            v = torch.ones(shape)
        elif key.endswith("norm.bias"):
            v = torch.zeros(shape)
        else:
            wshape = shape if leaf != "bias" else None
            if wshape is None:   # bias: bound from the sibling weight's fan_in
                wkey = key[: -len("bias")] + "weight"
                wshape = dict(spec)[wkey]
            bound = 1.0 / math.sqrt(max(1, math.prod(wshape[1:])))
            v = (torch.rand(shape, generator=gen) * 2 - 1) * bound
        out[key] = v
    return out


def sinusoid_table(times, dim, theta=10000.0):
    """SinusoidalPosEmb rows for the given timesteps, evaluated on the host exactly like the reference
    does on CPU (1D/model/unet.py:81-107, conv3d.py:139-151): fp32 exp / sin / cos of t * freq."""
    return _sinusoid(times.detach().cpu(), dim, theta)


@functools.lru_cache(maxsize=32)
def _sinusoid_freqs(n, device, theta):
    """exp(-log(theta) / (n - 1) * arange(n)) formed on the host in fp32 like the reference's, kept per device: a fresh
    host-to-device copy per call would drain the stream and cannot be recorded by a stream capture (safediffcon_amd/train_graph.py)"""
    return torch.exp(torch.arange(n) * -(math.log(theta) / (n - 1))).to(device)


def _sinusoid(t, dim, theta=10000.0):
    """rows on t's device; the frequency vector is always formed on the host (fp32 exp), like the LUT's"""
    half = dim // 2
    f = _sinusoid_freqs(half, t.device, theta)
    a = t[:, None] * f[None, :]
    if dim % 2 == 0:
        return torch.cat((a.sin(), a.cos()), dim=-1)
    half1 = (dim + 1) // 2
    f1 = _sinusoid_freqs(half1, t.device, theta)
    return torch.cat((a.sin(), (t[:, None] * f1[None, :]).cos()), dim=-1)


class _Builder:
    """Emits the stage calls of one U-Net forward into a Plan."""

    def __init__(self, net, plan):
        self.net, self.plan = net, plan
        self.cond_sites = []       # (index of the sdc_gn_apply call, offset into the conditioning row)

    def W(self, key, kind="conv"):
        return self.plan.conv_weight(lambda: self.net.P(key), kind)

    def V(self, key):
        return self.plan.vec(lambda: self.net.P(key))

    def ksize(self, key):
        shp = tuple(self.net.P(key).shape[2:])
        return (1,) * (3 - len(shp)) + shp

    def conv(self, x, prefix, *, x1=None, stride=1, pad=None, up=(1, 1, 1), residual=None, out=None, kind="conv",
             bias=True, gn_groups=0):
        wkey = f"{prefix}.weight"
        w = self.net.P(wkey)
        if kind == "convT":
            cout, k = w.shape[1], self.ksize(wkey)
        elif kind == "unshuffle":
            cout, k = w.shape[0], (1, 2, 2)
        else:
            cout, k = w.shape[0], self.ksize(wkey)
        if isinstance(stride, int):
            stride = tuple(stride if kk > 1 else 1 for kk in k)
        if pad is None:
            pad = tuple(kk // 2 for kk in k)
        b = self.V(f"{prefix}.bias") if bias else None
        return self.plan.conv(x, self.W(wkey, kind), b, cout, k, x1=x1, stride=stride, pad=pad, up=up,
                              up_mode=1 if kind == "convT" else 0, residual=residual, out=out, gn_groups=gn_groups)

    def final_conv(self, res_prefix, conv_prefix, h, groups, x1, out):
        """last ResnetBlock + the 1x1(x1) output conv into `out`; fused (Plan.gn_pointwise_out) where the net allows and the
        kernel's contract holds, else the block's own GroupNorm pass followed by the conv"""
        pool = self.plan.pool
        # (decided up front from the geometry: below 1024 positions per sample the block's own one-launch GroupNorm is the better form)
        if self.net.fuse_final_conv and h.shape[2] * h.shape[3] * h.shape[4] >= 1024:
            g, gn, rbuf = self.resnet(res_prefix, h, groups, x1=x1, defer_gn=True)
            wv, bv = self.V(f"{conv_prefix}.weight"), self.V(f"{conv_prefix}.bias")
            if self.plan.gn_pointwise_out(g, gn, wv, bv, out) is not None:
                pool.put(g)
                if rbuf is not None:
                    pool.put(rbuf)
                return
            self.plan.gn_apply_deferred(g, gn)       # outside the kernel's contract: the block's own pass from the same statistics
            if rbuf is not None:
                pool.put(rbuf)
            f = g
        else:
            f = self.resnet(res_prefix, h, groups, x1=x1)
        self.conv(f, conv_prefix, out=out)
        pool.put(f)

    def gn(self, x, prefix, groups, residual=None, cond_off=None):
        self.plan.gn_silu(x, self.V(f"{prefix}.weight"), self.V(f"{prefix}.bias"), groups, residual=residual)
        if cond_off is not None:
            self.cond_sites.append((len(self.plan.calls) - 1, cond_off))

    def resnet(self, prefix, x, groups, *, x1=None, defer_gn=False):
        """defer_gn: leave block2's GroupNorm + SiLU + residual add to the consumer: returns (raw conv output, gn tuple for
        Plan.linattn_block, residual buffer to recycle afterwards or None)"""
        net, pool = self.net, self.plan.pool
        h = self.conv(x, f"{prefix}.block1.proj", x1=x1, gn_groups=groups)
        self.gn(h, f"{prefix}.block1.norm", groups, cond_off=net.cond_offsets.get(prefix))
        g = self.conv(h, f"{prefix}.block2.proj", gn_groups=groups)
        pool.put(h)
        own = net.has(f"{prefix}.res_conv.weight")
        if own:
            r = self.conv(x, f"{prefix}.res_conv", x1=x1)
        else:
            assert x1 is None
            r = x
        if defer_gn:
            st = self.plan.gn_stats_deferred(g, groups)
            return g, (st, self.V(f"{prefix}.block2.norm.weight"), self.V(f"{prefix}.block2.norm.bias"), groups, r), (r if own else None)
        self.gn(g, f"{prefix}.block2.norm", groups, residual=r)
        if own:
            pool.put(r)
        return g


class _HipUNet(nn.Module):
    """Common shell: parameters under the reference's key names + a cache of engine plans."""

    def __init__(self, spec, dim, seed=None):
        super().__init__()
        gen = torch.Generator()
        gen.manual_seed(int(torch.initial_seed()) & 0x7FFFFFFF if seed is None else seed)
        for k, v in _default_init(spec, gen).items():
            _register(self, k, v)
        self._spec = spec
        self._plans = {}
        # conv algorithm (include/sdc.h): 4 (default) = fp32 Winograd wherever a kernel covers the shape -- F(2x2x2,3x3x3) over
        # (D, H, W) for the 3x3x3 stride-1 convs (8/27 of the direct form's MFMA work), F(2x2,3x3) over (H, W) for the 3x3 ones
        # (4/9), F(2,3) along W for the other 3-tap convs (2/3); 3 = without the depth transform; 2 = F(2,3) along W only;
        # 0 = fp32 direct everywhere.  All modes are fp32 end to end (rounding order differs).
        self.precision = 4
        # LinearAttention blocks of width 64 / 128 as the fused 3-launch form (csrc/sdc_lablock.hip); False = the
        # unfused chain norm -> 1x1 -> attention core -> 1x1 -> norm (kept for wider layers and for A/B checks)
        self.fuse_linattn = True
        # smoke net: the ResnetBlock in front of a fused LinearAttention block hands it its raw conv output; GroupNorm + SiLU +
        # residual add happen on the attention kernels' tile loads (False = a separate sdc_gn_apply pass, for A/B checks)
        self.fuse_gn_into_linattn = True
        # entry() compares a device checksum of the parameters with the packed plan's (sees `.data` writes; one small launch +
        # an 8-byte read-back per forward / sample call, never inside the graph-replayed loop)
        self.content_stamp = True
        # nearest-x2 upsample + 3x3 conv as four sub-pixel 2x2 convs with merged taps (4/9 of the multiply-adds; the merged
        # weights change the summation order by ~1e-7 relative); False = one conv with the upsampling folded into its gather
        self.subpixel_upsample = True
        # small-batch samplers (an 8-way shard of the 1-D configs leaves B = 16-32 per GPU): where a 3-tap conv's grid
        # leaves more than half of the CUs idle its input channels are split over 2-8 workgroups per tile (sdc_conv_splitk, summed
        # in split order: deterministic).  The split factor depends on the batch, so with it ON a trajectory's rounding depends on
        # the batch it rides in (fp32 sums in another order: ~1e-7); OFF (default) the samplers use sdc_conv only.
        self.split_small_grids = False
        # the last ResnetBlock's GroupNorm apply + SiLU + residual inside the 1x1 output conv that is its only reader (one streaming
        # pass, the normalised tensor is never written: sdc_gn_pointwise_out); False = sdc_gn_apply + sdc_conv (A/B checks)
        self.fuse_final_conv = True
        self.forward_graph = True    # model(x, t) replays a captured hipGraph; False = launch the call list every time
        self._side = None
        self.dim = dim
        self.self_condition = False
        # every time-conditioned ResnetBlock gets a slot [scale | shift] in the conditioning row
        self.cond_offsets, off = {}, 0
        for k, shape in spec:
            if k.endswith(".mlp.1.weight"):
                self.cond_offsets[k[: -len(".mlp.1.weight")]] = off
                off += shape[0]
        self.cond_width = off

    def P(self, key):
        m = self
        *mods, leaf = key.split(".")
        for name in mods:
            m = m._modules[name]
        return m._parameters[leaf]

    def has(self, key):
        try:
            return self.P(key) is not None
        except KeyError:
            return False

    def _trainer(self):
        if getattr(self, "_train_ctx", None) is None:
            from .autograd import Trainer
            self._train_ctx = Trainer(self)
        return self._train_ctx

    def forward_train(self, x, time):
        """eps = model(x, t) WITH an autograd graph over the parameters (fine-tuning path): every node runs
        libsdc_hip.so kernels in both directions (safediffcon_amd/autograd.py).  ``forward`` stays the graph-replayed,
        gradient-free sampler call."""
        if not x.is_cuda:
            raise RuntimeError("safediffcon_amd runs on MI355X only: tensors must be on a cuda (HIP) device; "
                               "there is no CPU fallback")
        from . import autograd
        fn = autograd.forward_train_smoke if isinstance(self, Unet3D_with_Conv3D) else autograd.forward_train_lucid
        with self._trainer().step(x.device):
            return fn(self, x.to(torch.float32), time)

    def _drop_plans(self):
        """device moves / dtype casts invalidate bound pointers: destroy the captured graphs, forget the plans"""
        self._train_ctx = None
        for e in self._plans.values():
            g = e.get("graph")
            if g is not None:
                torch.cuda.synchronize(e["x"].device)
                e["plan"].lib.sdc_graph_destroy(g)
                e["graph"] = None
        self._plans = {}
        self._side = None

    def _apply(self, fn, *a, **k):
        self._drop_plans()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, sd, strict=True, **kw):
        # rotary buffers added by newer rotary-embedding-torch releases are tolerated
        sd = {k: v for k, v in sd.items() if ("rotary_emb." not in k) or k.endswith("rotary_emb.freqs")}
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.refresh()
        return r

    def _stamp_tensors(self):
        return [t for t in list(self.parameters()) + list(self.buffers()) if t.is_cuda and t.numel() and t.element_size() == 4]

    def _content_stamp(self):
        """64-bit checksum of every parameter and buffer as they are in device memory now (`sdc_checksum_spans`: one launch,
        one 8-byte read-back).  The only stamp that sees writes made behind autograd's back: `p.data.copy_()` / `p.data.lerp_()`
        (ema_pytorch, used by 1D/model/trainer.py and tokamak/model/trainer.py) bump no version counter, and `p.data = ...`
        (the 2-D tree's EMA, video_diffusion_pytorch_conv3d.py:121-124) leaves `_version` at 0 while the allocator hands the
        same two addresses back in turn."""
        ts = self._stamp_tensors()
        if not ts:
            return 0
        from ._lib import SdcSpan
        dev = ts[0].device
        key = tuple((t.data_ptr(), t.numel()) for t in ts)
        tab = getattr(self, "_span_tab", None)
        if tab is None or tab[0] != key:
            arr = (SdcSpan * len(ts))(*[SdcSpan(t.data_ptr(), t.numel()) for t in ts])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            tab = self._span_tab = (key, host.to(dev), torch.zeros(1, dtype=torch.int64, device=dev))
        lib = get_lib()
        check(lib.sdc_checksum_spans(tab[1].data_ptr(), len(ts), tab[2].data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
              "sdc_checksum_spans")
        return int(tab[2].item())

    def _weights_stamp(self):
        """(cheap host part, content part).  The host part -- every parameter's (address, autograd version) -- moves on
        optimizer.step(), load_state_dict and re-seating to a new address; the content part (`content_stamp`, default on) also
        catches the writes the host part cannot see (see `_content_stamp`).  With `content_stamp = False` a caller that writes
        through `.data` must call `refresh()` itself."""
        host = hash(tuple((p.data_ptr(), p._version) for p in self.parameters()))
        dev = next(self.parameters()).device
        if not (self.content_stamp and dev.type == "cuda"):
            return (host, 0)
        # Cost: one small launch + one blocking 8-byte read-back per model(x, t) / sample() call (not per denoising step).  Under a
        # caller's stream capture a read-back is illegal: the last checksum taken outside the capture stands (a capture records
        # the packed weights' addresses, which a re-pack refreshes in place, so replays still see later re-packs).
        if torch.cuda.is_current_stream_capturing():
            return (host, getattr(self, "_last_content", 0))
        self._last_content = self._content_stamp()
        return (host, self._last_content)

    def _refresh_entry(self, e, stamp):
        e["plan"].refresh_weights()
        e["cond"].refresh_weights()
        e["lut_valid"] = False
        e["wstamp"] = stamp

    def refresh(self):
        """Re-pack weights of every cached plan now.  Optional while `content_stamp` is on (the default): `entry()` -- hence
        `forward`, `sample` and the fine-tuning path -- compares a content checksum of the parameters with the one taken when a
        plan packed them and re-packs by itself (the reference's loops call optimizer.step() / ema.update() and sample again
        without any such call: 1D/inference/inference_ft.py:183-187, tokamak/inference/pipeline.py:238-263,
        2d/inference_2d.py:267-279).  Mandatory after `.data` writes when `content_stamp` has been switched off."""
        stamp = self._weights_stamp()
        for e in self._plans.values():
            self._refresh_entry(e, stamp)

    # ------------------------------------------------------------------ plans
    def device(self):
        return next(self.parameters()).device

    def entry(self, shape, rows, lut=False):
        """Plan for an input of `shape` whose conditioning table has `rows` rows: one row per sample
        (lut=False, forward(x, time)) or one row per timestep read through a device-side t (lut=True, samplers)."""
        key = (tuple(shape), rows, bool(lut), int(self.precision), bool(self.fuse_linattn), bool(self.subpixel_upsample),
               bool(self.fuse_gn_into_linattn), bool(self.split_small_grids), bool(self.fuse_final_conv))
        stamp = self._weights_stamp()
        ent = self._plans.get(key)
        if ent is not None and ent["wstamp"] != stamp:        # parameters changed since this plan packed them
            self._refresh_entry(ent, stamp)
        if key not in self._plans:
            dev = self.device()
            if dev.type != "cuda":
                raise RuntimeError("safediffcon_amd runs on MI355X only: move the model to a cuda (HIP) device; "
                                   "there is no CPU fallback")
            plan = Plan(dev, precision=self.precision)
            plan.split_small_grids = bool(self.split_small_grids)
            x = torch.zeros(shape, dtype=torch.float32, device=dev)
            eps = torch.zeros(shape, dtype=torch.float32, device=dev)
            b = _Builder(self, plan)
            self._build(b, x, eps)
            plan.cond_sites = b.cond_sites
            # conditioning pipeline: sinusoid rows -> Linear -> GELU -> Linear -> SiLU -> all block MLPs at once
            cond = Plan(dev)
            td = self.dim * 4
            emb = torch.zeros((rows, self.dim, 1, 1, 1), dtype=torch.float32, device=dev)
            cb = _Builder(self, cond)
            h = cb.conv(emb, "time_mlp.1")
            cond.act(h, 1)
            temb = cb.conv(h, "time_mlp.3")
            cond.act(temb, 0)
            prefixes = list(self.cond_offsets)
            wcat = cond.conv_weight(lambda: torch.cat([self.P(f"{p}.mlp.1.weight") for p in prefixes], 0))
            bcat = cond.vec(lambda: torch.cat([self.P(f"{p}.mlp.1.bias") for p in prefixes], 0))
            ss = cond.conv(temb, wcat, bcat, self.cond_width, (1, 1, 1))
            assert temb.shape[1] == td
            self._plans[key] = dict(plan=plan, cond=cond, x=x, eps=eps, emb=emb, ss=ss, rows=rows, lut_valid=False,
                                    t_dev=None, wstamp=stamp)
        return self._plans[key]

    def bind_cond(self, ent, t_dev):
        """Point the conditioned GroupNorm calls at ent['ss']: per-sample rows (t_dev None) or LUT[*t_dev]."""
        plan, ss, W = ent["plan"], ent["ss"], self.cond_width
        for idx, off in plan.cond_sites:
            fn, a = plan.calls[idx]
            a = list(a)
            # sdc_gn_apply(x, stats, gamma, beta, ss, t_dev, ss_t_stride, ss_b_stride, ss_off, residual, y, B, C, G, S)
            # sdc_gn_fused(x, gamma, beta, ss, t_dev, ss_t_stride, ss_b_stride, ss_off, residual, y, B, C, G, S, eps)
            i0 = 3 if fn is plan.lib.sdc_gn_fused else 4
            if t_dev is None:
                a[i0:i0 + 5] = ss.data_ptr(), 0, 0, W, off
            else:
                a[i0:i0 + 5] = ss.data_ptr(), t_dev.data_ptr(), W, 0, off
            plan.calls[idx] = (fn, tuple(a))
        ent["t_dev"] = t_dev

    def fill_cond(self, ent, times, stream):
        ent["emb"].copy_(sinusoid_table(times, self.dim).reshape(ent["emb"].shape), non_blocking=False)
        ent["cond"].run(stream)

    def forward(self, x, time, x_self_cond=None, **unused):
        """eps = model(x, t) -- reference contract; x on the HIP device, time (B,) long/float."""
        if not x.is_cuda:
            raise RuntimeError("safediffcon_amd runs on MI355X only: tensors must be on a cuda (HIP) device; "
                               "there is no CPU fallback")
        ent = self.entry(tuple(x.shape), x.shape[0])
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.no_grad():
            if not ent.get("bound"):
                self.bind_cond(ent, None)
                ent["bound"] = True
            ent["x"].copy_(x)
            # sin / cos of the embedding on the device (no host round trip, no sync; host timesteps are shipped first)
            emb = _sinusoid(time.detach().to(x.device), self.dim)
            ent["emb"].copy_(emb.reshape(ent["emb"].shape))
            if not self.forward_graph:
                ent["cond"].run(stream)
                ent["plan"].run(stream)
                return ent["eps"].clone()
            # ~330 launches per forward: replayed as one hipGraph (captured once per input shape on a private stream;
            # the repacked weights, the conditioning table and x / eps live in buffers the graph keeps pointing at)
            cur = torch.cuda.current_stream(x.device)
            if self._side is None or self._side.device != x.device:
                self._side = torch.cuda.Stream(x.device)
            side = self._side
            side.wait_stream(cur)
            lib = ent["plan"].lib
            if ent.get("graph") is None:
                check(lib.sdc_graph_begin(side.cuda_stream), "sdc_graph_begin")
                try:
                    ent["cond"].run(side.cuda_stream)
                    ent["plan"].run(side.cuda_stream)
                finally:
                    g = C.c_void_p()
                    rc = lib.sdc_graph_end(side.cuda_stream, C.byref(g))
                check(rc, "sdc_graph_end")
                ent["graph"] = g
            check(lib.sdc_graph_launch(ent["graph"], side.cuda_stream), "sdc_graph_launch")
            cur.wait_stream(side)
            return ent["eps"].clone()


class _LucidUNet(_HipUNet):
    """Unet2D / Unet1D skeleton: forward order of 1D/model/unet.py:382-426 == tokamak/model/unet.py:359-408."""
    ND = 2
    NORM_MODE = 0      # 0 channel LayerNorm (1D tree), 1 RMSNorm (tokamak tree)

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, sinusoidal_pos_emb_theta=10000,
                 attn_dim_head=32, attn_heads=4, condition_on_residual=None):
        unsupported = dict(init_dim=init_dim, out_dim=out_dim, self_condition=self_condition or None,
                           learned_variance=learned_variance or None,
                           learned_sinusoidal_cond=learned_sinusoidal_cond or None,
                           random_fourier_features=random_fourier_features or None,
                           condition_on_residual=condition_on_residual)
        bad = [k for k, v in unsupported.items() if v is not None]
        if bad or attn_dim_head != 32 or attn_heads != 4 or sinusoidal_pos_emb_theta != 10000:
            raise NotImplementedError(f"options never enabled by the reference configs are not built: {bad}")
        super().__init__(spec_lucid(dim, tuple(dim_mults), channels, self.ND), dim)
        self.channels, self.dim_mults, self.groups = channels, tuple(dim_mults), resnet_block_groups
        self.out_dim = channels
        self.random_or_learned_sinusoidal_cond = False

    def _lin_attn(self, b, prefix, x):
        plan, pool = b.plan, b.plan.pool
        B, C = x.shape[0], x.shape[1]
        n = x.numel() // (B * C)
        if self.fuse_linattn and C in (64, 128) and n % 64 == 0:
            return plan.linattn_block(x, b.V(f"{prefix}.fn.norm.g"), b.W(f"{prefix}.fn.fn.to_qkv.weight"),
                                      b.W(f"{prefix}.fn.fn.to_out.0.weight"), b.V(f"{prefix}.fn.fn.to_out.0.bias"),
                                      b.V(f"{prefix}.fn.fn.to_out.1.g"), B, 1, n, (C * n, n, 0), self.NORM_MODE, self.NORM_MODE)
        xn = plan.chan_norm(x, b.V(f"{prefix}.fn.norm.g"), self.NORM_MODE)
        qkv = b.conv(xn, f"{prefix}.fn.fn.to_qkv", bias=False)
        pool.put(xn)
        o = pool.get((B, HID, *x.shape[2:]))
        plan.linattn(qkv, HEADS, B, 1, n, (3 * HID * n, n, 0), o, (HID * n, n, 0))
        pool.put(qkv)
        y = b.conv(o, f"{prefix}.fn.fn.to_out.0")
        pool.put(o)
        plan.chan_norm(y, b.V(f"{prefix}.fn.fn.to_out.1.g"), self.NORM_MODE, residual=x, out=y)
        return y

    def _full_attn(self, b, prefix, x):
        plan, pool = b.plan, b.plan.pool
        B, C = x.shape[0], x.shape[1]
        n = x.numel() // (B * C)
        xn = plan.chan_norm(x, b.V(f"{prefix}.fn.norm.g"), self.NORM_MODE)
        qkv = b.conv(xn, f"{prefix}.fn.fn.to_qkv", bias=False)
        pool.put(xn)
        o = pool.get((B, HID, *x.shape[2:]))
        plan.attn(qkv, o, HEADS, B, 1, n, (3 * HID * n, n, 0, 1), (HID * n, n, 0, 1))
        pool.put(qkv)
        y = b.conv(o, f"{prefix}.fn.fn.to_out", residual=x)
        pool.put(o)
        return y

    def _build(self, b, x, eps):
        pool, G = b.plan.pool, self.groups
        nres = len(self.dim_mults)
        x5, e5 = as5(x), as5(eps)
        h = b.conv(x5, "init_conv")
        r = h
        hs = []
        for i in range(nres):
            p = f"downs.{i}"
            x1 = b.resnet(f"{p}.0", h, G)
            if h is not r:
                pool.put(h)
            hs.append(x1)
            x2 = b.resnet(f"{p}.1", x1, G)
            x3 = self._lin_attn(b, f"{p}.2", x2)
            pool.put(x2)
            hs.append(x3)
            h = self._down(b, f"{p}.3", x3, i == nres - 1)
        m = b.resnet("mid_block1", h, G)
        pool.put(h)
        a = self._full_attn(b, "mid_attn", m)
        pool.put(m)
        h = b.resnet("mid_block2", a, G)
        pool.put(a)
        for i in range(nres):
            p = f"ups.{i}"
            s = hs.pop()
            y1 = b.resnet(f"{p}.0", h, G, x1=s)
            pool.put(h), pool.put(s)
            s = hs.pop()
            y2 = b.resnet(f"{p}.1", y1, G, x1=s)
            pool.put(y1), pool.put(s)
            y3 = self._lin_attn(b, f"{p}.2", y2)
            pool.put(y2)
            h = self._up(b, f"{p}.3", y3, i == nres - 1)
            pool.put(y3)
        b.final_conv("final_res_block", "final_conv", h, G, r, e5)
        pool.put(h), pool.put(r)


class Unet2D(_LucidUNet):
    """1D Burgers denoiser over the (time, space) image -- 1D/model/unet.py:263-426."""
    ND, NORM_MODE = 2, 0

    def _down(self, b, p, x, last):
        if last:
            return b.conv(x, p)
        return b.conv(x, f"{p}.1", kind="unshuffle", stride=(1, 2, 2), pad=(0, 0, 0))   # Downsample2d :39-43

    def _up(self, b, p, x, last):
        if last:
            return b.conv(x, p)
        if self.subpixel_upsample:                                                      # Upsample2d :33-37
            return b.plan.upsample2_conv3(x, lambda p=p: self.P(f"{p}.1.weight"), b.V(f"{p}.1.bias"), self.P(f"{p}.1.weight").shape[0])
        return b.conv(x, f"{p}.1", up=(1, 2, 2))


class Unet1D(_LucidUNet):
    """Tokamak denoiser along time -- tokamak/model/unet.py:263-408."""
    ND, NORM_MODE = 1, 1

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=12, **kw):
        super().__init__(dim, init_dim, out_dim, dim_mults, channels, **kw)

    def _down(self, b, p, x, last):
        if last:
            return b.conv(x, p)
        return b.conv(x, p, stride=(1, 1, 2), pad=(0, 0, 1))                              # Conv1d(k4,s2,p1) :30-31

    def _up(self, b, p, x, last):
        if last:
            return b.conv(x, p)
        return b.conv(x, f"{p}.1", up=(1, 1, 2))                                          # Upsample :24-28


def rel_pos_bias_table(weight, n, num_buckets=32, max_distance=32):
    """RelativePositionBias.forward (conv3d.py:74-112): integer bucket indices on the host, gather of the
    (32, heads) embedding -> (heads, n, n)."""
    q = torch.arange(n)
    rel = q[None, :] - q[:, None]
    m = -rel
    nb = num_buckets // 2
    ret = (m < 0).long() * nb
    m = m.abs()
    max_exact = nb // 2
    small = m < max_exact
    large = max_exact + (torch.log(m.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    bucket = (ret + torch.where(small, m, large)).to(weight.device)
    return weight[bucket].permute(2, 0, 1).contiguous()


class Unet3D_with_Conv3D(_HipUNet):
    """2D smoke denoiser -- conv3d.py:357-574.  Input/Output (B, F, C, H, W) frame-major."""

    def __init__(self, dim, cond_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=6, attn_heads=4,
                 attn_dim_head=32, use_bert_text_cond=False, init_dim=None, init_kernel_size=7,
                 use_sparse_linear_attn=True, block_type="resnet", resnet_groups=8):
        if (cond_dim is not None or out_dim is not None or use_bert_text_cond or init_dim is not None
                or init_kernel_size != 7 or not use_sparse_linear_attn or attn_heads != 4 or attn_dim_head != 32):
            raise NotImplementedError("options never enabled by the reference's 2D scripts are not built")
        super().__init__(spec_smoke(dim, tuple(dim_mults), channels), dim)
        self.channels, self.dim_mults, self.groups = channels, tuple(dim_mults), resnet_groups
        self.has_cond = False
        # the reference shares ONE RotaryEmbedding instance between all temporal-attention layers
        # (conv3d.py:381-383): tie the eight state_dict entries to a single Parameter
        shared = self.P("init_temporal_attn.fn.fn.fn.rotary_emb.freqs")
        shared.requires_grad_(False)      # rotary-embedding-torch: nn.Parameter(freqs, requires_grad=learned_freq), learned_freq=False
        for k, _ in self._spec:
            if k.endswith("rotary_emb.freqs"):
                m = self
                for name in k.split(".")[:-1]:
                    m = m._modules[name]
                m._parameters["freqs"] = shared

    # ---- attention blocks
    def _attn_tables(self, b, F):
        plan = b.plan
        if not hasattr(plan, "_rot"):
            def rot():
                fr = self.P("init_temporal_attn.fn.fn.fn.rotary_emb.freqs").detach().float().cpu()
                ang = torch.arange(F, dtype=torch.float32)[:, None] * fr[None, :]
                return torch.stack((ang.cos(), ang.sin()), dim=-1).reshape(-1)
            plan._rot = plan.packed(rot)
            plan._bias = plan.packed(lambda: rel_pos_bias_table(
                self.P("time_rel_pos_bias.relative_attention_bias.weight").detach(), F).reshape(-1))
        return plan._rot, plan._bias

    def _temporal(self, b, prefix, x):
        plan, pool = b.plan, b.plan.pool
        B, C, F, H, W = x.shape
        hw = H * W
        rot, bias = self._attn_tables(b, F)
        if self.fuse_linattn and C == 64 and F == 32 and hw % 8 == 0 and x.is_contiguous():
            return plan.tattn_block(x, b.V(f"{prefix}.fn.norm.gamma"), b.W(f"{prefix}.fn.fn.fn.to_qkv.weight"),
                                    b.W(f"{prefix}.fn.fn.fn.to_out.weight"), rot, bias)
        xn = plan.chan_norm(x, b.V(f"{prefix}.fn.norm.gamma"), 0)
        qkv = b.conv(xn, f"{prefix}.fn.fn.fn.to_qkv", bias=False)
        pool.put(xn)
        o = pool.get((B, HID, F, H, W))
        plan.attn(qkv, o, HEADS, B, hw, F, (3 * HID * F * hw, F * hw, 1, hw), (HID * F * hw, F * hw, 1, hw), rot, bias)
        pool.put(qkv)
        y = b.conv(o, f"{prefix}.fn.fn.fn.to_out", bias=False, residual=x)
        pool.put(o)
        return y

    def _la_fusable(self, shape):
        B, C, F, H, W = shape
        return self.fuse_linattn and C in (64, 128) and (H * W) % 64 == 0 and B * F < 65536

    def _resnet_then_spatial_linear(self, b, res_prefix, la_prefix, x, G):
        """ResnetBlock -> SpatialLinearAttention (conv3d.py:537-545, :552-557).  The ResnetBlock's output feeds nothing but the
        attention block, so where the fused attention kernels run its last GroupNorm + SiLU + residual add rides on their tile
        loads (sdc_linattn_block_gn) instead of a pass of its own over HBM."""
        pool = b.plan.pool
        cout = self.P(f"{res_prefix}.block2.proj.weight").shape[0]
        if self.fuse_gn_into_linattn and self._la_fusable((x.shape[0], cout, *x.shape[2:])):
            g, gn, r = b.resnet(res_prefix, x, G, defer_gn=True)
            y = self._spatial_linear(b, la_prefix, g, gn=gn)
            if r is not None:
                pool.put(r)
            return y, g
        a = b.resnet(res_prefix, x, G)
        return self._spatial_linear(b, la_prefix, a), a

    def _spatial_linear(self, b, prefix, x, gn=None):
        plan, pool = b.plan, b.plan.pool
        B, C, F, H, W = x.shape
        hw = H * W
        if self._la_fusable(x.shape):
            return plan.linattn_block(x, b.V(f"{prefix}.fn.norm.gamma"), b.W(f"{prefix}.fn.fn.to_qkv.weight"),
                                      b.W(f"{prefix}.fn.fn.to_out.weight"), b.V(f"{prefix}.fn.fn.to_out.bias"), None,
                                      B, F, hw, (C * F * hw, F * hw, hw), 0, -1, gn=gn)
        assert gn is None
        xn = plan.chan_norm(x, b.V(f"{prefix}.fn.norm.gamma"), 0)
        qkv = b.conv(xn, f"{prefix}.fn.fn.to_qkv", bias=False)
        pool.put(xn)
        o = pool.get((B, HID, F, H, W))
        plan.linattn(qkv, HEADS, B, F, hw, (3 * HID * F * hw, F * hw, hw), o, (HID * F * hw, F * hw, hw))
        pool.put(qkv)
        y = b.conv(o, f"{prefix}.fn.fn.to_out", residual=x)
        pool.put(o)
        return y

    def _spatial_full(self, b, prefix, x):
        plan, pool = b.plan, b.plan.pool
        B, C, F, H, W = x.shape
        hw = H * W
        xn = plan.chan_norm(x, b.V(f"{prefix}.fn.norm.gamma"), 0)
        qkv = b.conv(xn, f"{prefix}.fn.fn.fn.to_qkv", bias=False)
        pool.put(xn)
        o = pool.get((B, HID, F, H, W))
        plan.attn(qkv, o, HEADS, B, F, hw, (3 * HID * F * hw, F * hw, hw, 1), (HID * F * hw, F * hw, hw, 1))
        pool.put(qkv)
        y = b.conv(o, f"{prefix}.fn.fn.fn.to_out", bias=False, residual=x)
        pool.put(o)
        return y

    def _build(self, b, x, eps):
        pool, G = b.plan.pool, self.groups
        nres = len(self.dim_mults)
        x5 = x.permute(0, 2, 1, 3, 4)          # (B,C,F,H,W) view of the frame-major state: strides do the permute
        e5 = eps.permute(0, 2, 1, 3, 4)
        h0 = b.conv(x5, "init_conv")
        h = self._temporal(b, "init_temporal_attn", h0)
        pool.put(h0)
        r = h
        hs = []
        for i in range(nres):
            p = f"downs.{i}"
            a1 = b.resnet(f"{p}.0", h, G)
            if h is not r:
                pool.put(h)
            a3, a2 = self._resnet_then_spatial_linear(b, f"{p}.1", f"{p}.2", a1, G)
            pool.put(a1)
            pool.put(a2)
            a4 = self._temporal(b, f"{p}.3", a3)
            pool.put(a3)
            hs.append(a4)
            h = a4
            if i < nres - 1:
                h = b.conv(a4, f"{p}.4", stride=(1, 2, 2), pad=(0, 1, 1))
        m1 = b.resnet("mid_block1", h, G)        # h (== hs[-1]) stays alive as a skip
        m2 = self._spatial_full(b, "mid_spatial_attn", m1)
        pool.put(m1)
        m3 = self._temporal(b, "mid_temporal_attn", m2)
        pool.put(m2)
        h = b.resnet("mid_block2", m3, G)
        pool.put(m3)
        for i in range(nres):
            p = f"ups.{i}"
            s = hs.pop()
            u1 = b.resnet(f"{p}.0", h, G, x1=s)
            pool.put(h), pool.put(s)
            u3, u2 = self._resnet_then_spatial_linear(b, f"{p}.1", f"{p}.2", u1, G)
            pool.put(u1)
            pool.put(u2)
            u4 = self._temporal(b, f"{p}.3", u3)
            pool.put(u3)
            h = u4
            if i < nres - 1:
                # ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1): four 2x2 sub-pixel convs (the zero-stuffed single conv,
                # kind="convT", does 4x the MFMA work)
                h = b.plan.conv_transpose_422(u4, lambda p=p: self.P(f"{p}.4.weight"), b.V(f"{p}.4.bias"),
                                              self.P(f"{p}.4.weight").shape[1])
                pool.put(u4)
        b.final_conv("final_conv.0", "final_conv.1", h, G, r, e5)
        pool.put(h), pool.put(r)
