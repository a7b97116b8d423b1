"""Stage-level execution engine: turns a U-Net description into a flat list of
libsdc_hip.so calls with every pointer, stride and size bound ahead of time.

PyTorch is plumbing here (device memory, streams): tensors are allocated once per
plan from a reuse pool, handed to the kernels as raw pointers, and the recorded
call list is either replayed from Python or captured into one hipGraph.
Nothing in this file computes on the CPU or through torch ops on the hot path.
"""
import ctypes as C
import functools
import os
import math

import torch

from . import _lib
from ._lib import SdcConvDesc, check


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _s5(t):
    assert t.dim() == 5, t.shape
    return tuple(int(s) for s in t.stride())


def as5(t):
    """(B,C,L) / (B,C,H,W) / (B,C,D,H,W) -> 5-D view (B,C,D,H,W)."""
    while t.dim() < 5:
        t = t.unsqueeze(2)
    return t


@functools.lru_cache(maxsize=16)
def _up2_merge(parity, device):
    """tap-merging matrix of the sub-pixel form of nearest-x2 upsampling + 3-tap conv, kept per device (no host-to-device copy per call)"""
    return torch.tensor([[1, 0, 0], [0, 1, 1]] if parity == 0 else [[1, 1, 0], [0, 0, 1]], dtype=torch.float64, device=device)


def pack_conv_weight(t, kind="conv", precision=0):
    """Kernel layout of a conv weight (include/sdc.h, SdcConvDesc.precision):
    nn.Conv{1,2,3}d weight (Cout,Cin,*k) -> Wp [taps*Cin][Cout], followed for precision >= 2 by the Winograd taps;
    kind 'convT': nn.ConvTranspose3d weight (Cin,Cout,*k) -> flipped-tap conv weight;
    kind 'unshuffle': 1x1 conv after 'b c (h p1) (w p2) -> b (c p1 p2) h w' -> 2x2 stride-2 conv;
    kind ('convT_sub', ph, pw) / ('up2_sub', ph, pw): the sub-pixel forms below."""
    def base():
        if kind == "convT":
            t5 = as5(t)                                   # (Cin, Cout, kD, kH, kW)
            t5 = t5.flip(2, 3, 4).permute(2, 3, 4, 0, 1)  # (kD,kH,kW,Cin,Cout)
            return t5.reshape(-1, t5.shape[-1]).contiguous()
        if isinstance(kind, tuple) and kind[0] == "convT_sub":
            # sub-pixel form of ConvTranspose (1,4,4)/(1,2,2)/(0,1,1): output parity (ph, pw) is a 2x2 conv of the
            # input with taps kh in (3,1) [ph=0] / (2,0) [ph=1] (same along W): out[2j+ph] = sum_t x[j+t-(1-ph)] W[kh_t]
            _, ph, pw = kind
            t5 = as5(t)                                   # (Cin, Cout, 1, 4, 4)
            kh = (3, 1) if ph == 0 else (2, 0)
            kw = (3, 1) if pw == 0 else (2, 0)
            # (taps (3, 1) = the odd taps reversed, (2, 0) = the even ones reversed -- as slices: indexing with a Python list
            # builds an index tensor on the host and copies it over, which drains the stream every call and cannot be recorded by a
            # stream capture (safediffcon_amd/train_graph.py))
            sub = t5[:, :, 0][:, :, (1 - ph)::2][:, :, :, (1 - pw)::2].flip(2, 3)          # (Cin, Cout, 2, 2)
            assert kh == ((3, 1) if ph == 0 else (2, 0)) and kw == ((3, 1) if pw == 0 else (2, 0))
            return sub.permute(2, 3, 0, 1).reshape(-1, sub.shape[1]).contiguous()
        if isinstance(kind, tuple) and kind[0] == "up2_sub":
            # nearest x2 upsampling + 3x3 conv (pad 1), output parity (ph, pw): rows 2i+ph of the upsampled image see
            # x[i-1], x[i], x[i] (ph = 0) or x[i], x[i], x[i+1] (ph = 1) -> a 2-tap kernel with merged weights
            # (W0, W1+W2) on rows (i-1, i)  /  (W0+W1, W2) on rows (i, i+1); same along W.  9 taps -> 4.
            _, ph, pw = kind
            t5 = as5(t).to(torch.float64)                 # (Cout, Cin, 1, 3, 3)
            mh, mw = _up2_merge(ph, t5.device), _up2_merge(pw, t5.device)
            sub = torch.einsum("ah,bw,oihw->oiab", mh, mw, t5[:, :, 0])                 # (Cout, Cin, 2, 2)
            return sub.permute(2, 3, 1, 0).reshape(-1, sub.shape[0]).to(torch.float32).contiguous()
        if kind == "unshuffle":
            co, c4 = t.shape[0], t.shape[1]
            t4 = t.reshape(co, c4 // 4, 2, 2)             # (Cout, C, p1, p2)
            return t4.permute(2, 3, 1, 0).reshape(-1, co).contiguous()
        t5 = as5(t)                                       # (Cout, Cin, kD, kH, kW)
        return t5.permute(2, 3, 4, 1, 0).reshape(-1, t5.shape[0]).contiguous()
    wp = base().to(torch.float32)
    if precision not in (2, 3, 4, 5) or kind != "conv" or t.shape[-1] != 3:
        return wp
    # fp32 Wp followed by the Winograd F(2,3) taps along W, [(kd*kH + kh)*4 + xi][Cin][Cout]:
    # G g with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], formed in fp64 and rounded once;
    # precision 3 appends, for 3x3 (kH = kW = 3) taps, the F(2x2,3x3) taps G g G^T over (H, W): [kd][Cin][Cout][j*4 + xi]
    t5 = as5(t).to(torch.float64)                 # (Cout, Cin, kD, kH, 3)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64, device=t5.device)
    u = torch.einsum("xk,oidhk->dhxio", G, t5)    # (kD, kH, 4, Cin, Cout)
    parts = [wp.reshape(-1), u.reshape(-1).to(torch.float32)]
    if precision == 5 and t5.shape[2] == 1 and t5.shape[3] == 1:
        # precision 5, 1-D convs: the F(4,3) taps G43 g, [xi][Cin][Cout] with xi < 6, after everything else (csrc/sdc_conv.hip conv_wg_kernel NX = 6)
        G43 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                            [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64, device=t5.device)
        u43 = torch.einsum("xk,oik->xio", G43, t5[:, :, 0, 0])
        return torch.cat(parts + [u43.reshape(-1).to(torch.float32)])
    if precision >= 3 and t5.shape[3] == 3:
        u2 = torch.einsum("jh,xk,oidhk->diojx", G, G, t5)   # (kD, Cin, Cout, 4, 4): 16 components contiguous
        parts.append(u2.reshape(-1).to(torch.float32))
        if precision >= 4 and t5.shape[2] == 3:
            # F(2x2x2,3x3x3) taps, G along the depth as well: [jd][Cin][Cout][j*4 + xi]
            u3 = torch.einsum("zd,jh,xk,oidhk->ziojx", G, G, G, t5)
            parts.append(u3.reshape(-1).to(torch.float32))
    return torch.cat(parts)


class Pool:
    """Size-keyed free list so that activation buffers are reused along the plan."""

    def __init__(self, device):
        self.device = device
        self.free = {}
        self.all = []
        self.bytes = 0

    def get(self, shape):
        n = int(math.prod(shape))
        lst = self.free.get(n)
        if lst:
            return lst.pop().view(shape)
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        self.all.append(t)
        self.bytes += n * 4
        return t.view(shape)

    def put(self, t):
        self.free.setdefault(t.numel(), []).append(t.reshape(-1))


class Plan:
    """Recorded kernel calls; `run(stream)` replays them (the samplers capture that replay into a hipGraph)."""

    def __init__(self, device, precision=0):
        self.device = torch.device(device)
        self.lib = _lib.get_lib()
        # conv algorithm (include/sdc.h): 0 direct fp32 MFMA | 2 fp32 Winograd F(2,3) along W | 3 F(2x2,3x3) over (H, W) where
        # covered, else as 2 | 4 (default of the nets) F(2x2x2,3x3x3) over (D, H, W) where covered, else as 3 | 5 (opt-in) as 4,
        # with F(4,3) along W for the (1,1,3) convs (Conv1d k3: half the direct MFMA work at 3x the rounding error)
        self.precision = int(precision)
        if self.precision not in (0, 2, 3, 4, 5):
            raise ValueError(f"precision must be 0, 2, 3, 4 or 5 (got {precision})")
        self.calls = []          # (fn, args, keepalive)
        self.pool = Pool(self.device)
        self.keep = []           # descriptors / tensors that must outlive the plan
        self.repackers = []      # (dst tensor, fn() -> src tensor) to refresh repacked weights
        self._stats = None
        self._ctx = None
        self._gn_parts = {}      # out.data_ptr() -> (partial-sum buffer, parts per (sample, group), groups) left by sdc_conv_gn
        self.fuse_gn_stats = True    # False: separate statistics pass after every conv (A/B checks)
        self.fuse_gn_small = True    # False: small groups take the three-launch path (partial, finalize, apply)
        self.split_small_grids = False   # True: sdc_conv_splitk where a conv's grid leaves most CUs idle (net.split_small_grids)

    # ------------------------------------------------------------------ execution
    def run(self, stream):
        for fn, args in self.calls:
            rc = fn(*args, stream)
            if rc:
                check(rc, fn.__name__)

    def refresh_weights(self):
        """Re-run every weight repack (after an optimiser step / load_state_dict)."""
        with torch.no_grad():
            for dst, fn in self.repackers:
                dst.copy_(fn())

    # ------------------------------------------------------------------ scratch
    def _stats_buf(self, B, G):
        need = int(self.lib.sdc_gn_stats_bytes(B, G))
        if self._stats is None or self._stats.numel() * 4 < need:
            self._stats = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
            self.keep.append(self._stats)
        return self._stats

    def _ctx_buf(self, n):
        if self._ctx is None or self._ctx.numel() < n:
            self._ctx = torch.empty(n, dtype=torch.float32, device=self.device)
            self.keep.append(self._ctx)
        return self._ctx

    def _emit(self, fn, *args):
        self.calls.append((fn, args))

    # ------------------------------------------------------------------ weights
    def packed(self, src_fn):
        """Register a repacked weight: src_fn() -> tensor in kernel layout; refreshed by refresh_weights()."""
        with torch.no_grad():
            w = src_fn().detach().to(self.device, torch.float32).contiguous().clone()
        self.repackers.append((w, lambda: src_fn().detach().to(self.device, torch.float32)))
        self.keep.append(w)
        return w

    def conv_weight(self, w, kind="conv"):
        """Register the kernel layout of a conv weight (pack_conv_weight), refreshed by refresh_weights()."""
        return self.packed(lambda: pack_conv_weight(w() if callable(w) else w, kind, self.precision))

    def vec(self, p):
        return self.packed(lambda: (p() if callable(p) else p).reshape(-1))

    # ------------------------------------------------------------------ stages
    def conv(self, x, wp, bias, cout, k, *, x1=None, stride=(1, 1, 1), pad=(0, 0, 0), up=(1, 1, 1), up_mode=0,
             residual=None, out=None, gn_groups=0):
        """x (and optional x1, channel-concatenated) are 5-D views; returns out (B,cout,oD,oH,oW).
        gn_groups > 0: a GroupNorm over `out` follows -- where the conv epilogue can sum its statistics (sdc_conv_gn) the
        partial sums are kept for the gn_silu call on `out`, which then skips its own pass over the tensor."""
        B, c0, iD, iH, iW = x.shape
        c1 = 0 if x1 is None else x1.shape[1]

        def osz(i, u, kk, s, p):
            v = (i - 1) * u + 1 if up_mode else i * u
            return (v + 2 * p - kk) // s + 1

        o = tuple(osz(i, u, kk, s, p) for i, u, kk, s, p in zip((iD, iH, iW), up, k, stride, pad))
        if out is None:
            out = self.pool.get((B, cout, *o))
        else:
            # a caller-sized output may differ from the symmetric-padding size by a one-sided margin (include/sdc.h); the
            # library validates it
            assert tuple(out.shape[:2]) == (B, cout) and out.dim() == 5, (out.shape, (B, cout, *o))
            o = tuple(out.shape[2:])
        nw = k[0] * k[1] * k[2] * (c0 + c1) * cout
        wino = wp.dim() == 1 and k[2] == 3 and wp.numel() == nw + nw // 3 * 4
        wino43 = wp.dim() == 1 and tuple(k) == (1, 1, 3) and wp.numel() == nw + nw // 3 * 4 + nw // 3 * 6   # precision 5: + F(4,3) taps
        wino2 = wp.dim() == 1 and k[1] == 3 and k[2] == 3 and wp.numel() == nw + nw // 3 * 4 + nw // 9 * 16
        wino3 = wp.dim() == 1 and tuple(k) == (3, 3, 3) and wp.numel() == nw + nw // 3 * 4 + nw // 9 * 16 + nw // 27 * 64
        assert wino or wino2 or wino3 or wino43 or wp.numel() == nw, (wp.shape, k, c0, c1, cout)
        d = SdcConvDesc()
        d.B, d.Cin0, d.Cin1, d.Cout = B, c0, c1, cout
        d.iD, d.iH, d.iW = iD, iH, iW
        d.oD, d.oH, d.oW = o
        d.kD, d.kH, d.kW = k
        d.sD, d.sH, d.sW = stride
        d.pD, d.pH, d.pW = pad
        d.uD, d.uH, d.uW = up
        d.up_mode, d.precision = up_mode, (5 if wino43 else 4 if wino3 else 3 if wino2 else 2 if wino else 0)
        d.x0s[:] = _s5(x)
        d.x1s[:] = _s5(x1) if x1 is not None else (0,) * 5
        d.ys[:] = _s5(out)
        d.rs[:] = _s5(residual) if residual is not None else (0,) * 5
        if residual is not None:
            assert tuple(residual.shape) == tuple(out.shape)
        self.keep += [d, x, x1, wp, bias, residual, out]     # the call list holds raw pointers only
        if self.split_small_grids and residual is None:
            # a grid that leaves most of the chip idle (small batch, deep level): Cin split over several workgroups per tile.  The
            # statistics of a following GroupNorm then come from its own kernel (these tensors are small: sdc_gn_fused)
            nsplit = int(self.lib.sdc_conv_splitk_bytes(C.byref(d)))
            if nsplit:
                work = torch.empty(nsplit // 4, dtype=torch.float32, device=self.device)
                self.keep.append(work)
                self._emit(self.lib.sdc_conv_splitk, C.byref(d), _ptr(x), _ptr(x1), _ptr(wp), _ptr(bias), _ptr(out), _ptr(work), nsplit)
                return out
        nparts = int(self.lib.sdc_conv_gnparts(C.byref(d), gn_groups)) if (gn_groups and self.fuse_gn_stats) else 0
        if nparts > 0:
            parts = torch.empty(B * gn_groups * nparts * 2, dtype=torch.float64, device=self.device)
            self.keep.append(parts)
            self._gn_parts[out.data_ptr()] = (parts, nparts, gn_groups)
            self._emit(self.lib.sdc_conv_gn, C.byref(d), _ptr(x), _ptr(x1), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(out),
                       _ptr(parts), gn_groups)
            return out
        self._emit(self.lib.sdc_conv, C.byref(d), _ptr(x), _ptr(x1), _ptr(wp), _ptr(bias), _ptr(residual), _ptr(out))
        return out

    def conv_transpose_422(self, x, w, bias, cout):
        """nn.ConvTranspose3d(C, cout, (1,4,4), (1,2,2), (0,1,1)) as four 2x2 stride-1 convs, one per output parity,
        each writing its quarter of the output through strides (conv3d.py:159-160).  The zero-stuffed form spends
        3/4 of its MFMA work on inserted zeros."""
        B, c0, iD, iH, iW = x.shape
        out = self.pool.get((B, cout, iD, 2 * iH, 2 * iW))
        for ph in (0, 1):
            for pw in (0, 1):
                wp = self.conv_weight(w, ("convT_sub", ph, pw))
                self.conv(x, wp, bias, cout, (1, 2, 2), pad=(0, 1 - ph, 1 - pw), out=out[:, :, :, ph::2, pw::2])
        return out

    def upsample2_conv3(self, x, w, bias, cout):
        """nn.Upsample(scale_factor=2, mode='nearest') + Conv2d(3x3, pad 1) (1D/model/unet.py:33-37) as four 2x2 convs of the
        low-resolution input, one per output parity, with the taps that land on the same source pixel merged."""
        B, c0, iD, iH, iW = x.shape
        out = self.pool.get((B, cout, iD, 2 * iH, 2 * iW))
        for ph in (0, 1):
            for pw in (0, 1):
                wp = self.conv_weight(w, ("up2_sub", ph, pw))
                self.conv(x, wp, bias, cout, (1, 2, 2), pad=(0, 1 - ph, 1 - pw), out=out[:, :, :, ph::2, pw::2])
        return out

    def gn_silu(self, x, gamma, beta, groups, *, ss=None, t_dev=None, ss_t_stride=0, ss_b_stride=0, ss_off=0,
                residual=None, out=None, eps=1e-5):
        """GroupNorm -> (scale+1, shift) -> SiLU (+ residual); x contiguous 5-D; in place by default."""
        assert x.is_contiguous()
        B, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (B * Cc)
        st = self._stats_buf(B, groups)
        out = x if out is None else out
        self.keep += [x, gamma, beta, ss, t_dev, residual, out]
        fused = self._gn_parts.pop(x.data_ptr(), None)
        if fused is not None and fused[2] == groups:
            self._emit(self.lib.sdc_gn_finalize, _ptr(fused[0]), _ptr(st), B, groups, fused[1], (Cc // groups) * S, eps)
        elif self.fuse_gn_small and self.lib.sdc_gn_fused_ok(B, Cc, groups, S):
            # small groups without conv-epilogue sums (the deep levels of the 1-D nets): statistics + apply in one launch
            # (argument positions 3.. match sdc_gn_apply's 4..: bind_cond patches either form)
            self._emit(self.lib.sdc_gn_fused, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(ss), _ptr(t_dev), ss_t_stride, ss_b_stride,
                       ss_off, _ptr(residual), _ptr(out), B, Cc, groups, S, eps)
            return out
        else:
            self._emit(self.lib.sdc_gn_stats, _ptr(x), _ptr(st), B, Cc, groups, S, eps)
        self._emit(self.lib.sdc_gn_apply, _ptr(x), _ptr(st), _ptr(gamma), _ptr(beta), _ptr(ss), _ptr(t_dev),
                   ss_t_stride, ss_b_stride, ss_off, _ptr(residual), _ptr(out), B, Cc, groups, S)
        return out

    def chan_norm(self, x, g, mode, *, residual=None, out=None, eps=1e-5):
        assert x.is_contiguous()
        B, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (B * Cc)
        if out is None:
            out = self.pool.get(tuple(x.shape))
        self.keep += [x, g, residual, out]
        self._emit(self.lib.sdc_chan_norm, _ptr(x), _ptr(g), _ptr(residual), _ptr(out), B, Cc, S, mode, eps)
        return out

    def act(self, x, kind, out=None):
        out = x if out is None else out
        self.keep += [x, out]
        self._emit(self.lib.sdc_act, _ptr(x), _ptr(out), x.numel(), kind)
        return out

    def linattn(self, qkv, heads, outer, inner, n, q_strides, out, o_strides):
        ctx = self._ctx_buf(outer * inner * heads * 32 * 32)
        self.keep += [qkv, out]
        self._emit(self.lib.sdc_linattn, _ptr(qkv), _ptr(ctx), _ptr(out), outer, inner, heads, n, *q_strides, *o_strides)
        return out

    def gn_stats_deferred(self, x, groups, eps=1e-5):
        """GroupNorm statistics of x -- (mean, rstd) per (sample, group) -- WITHOUT the apply pass: the consumer
        (linattn_block(..., gn=...)) normalises, activates and adds the residual while it loads its tiles."""
        assert x.is_contiguous()
        B, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (B * Cc)
        # its own buffer (read two launches later), sized like _stats_buf: sdc_gn_stats keeps its fp64 partials behind the (mean, rstd) pairs
        st = torch.empty((int(self.lib.sdc_gn_stats_bytes(B, groups)) + 3) // 4, dtype=torch.float32, device=self.device)
        self.keep += [x, st]
        fused = self._gn_parts.pop(x.data_ptr(), None)
        if fused is not None and fused[2] == groups:
            self._emit(self.lib.sdc_gn_finalize, _ptr(fused[0]), _ptr(st), B, groups, fused[1], (Cc // groups) * S, eps)
        else:
            self._emit(self.lib.sdc_gn_stats, _ptr(x), _ptr(st), B, Cc, groups, S, eps)
        return st

    def gn_apply_deferred(self, x, gn):
        """the apply pass of a GroupNorm whose statistics gn_stats_deferred took: x <- SiLU(GN(x)) (+ residual), in place"""
        st, gamma, beta, groups, res = gn
        B, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (B * Cc)
        self.keep += [x, st, gamma, beta, res]
        self._emit(self.lib.sdc_gn_apply, _ptr(x), _ptr(st), _ptr(gamma), _ptr(beta), 0, 0, 0, 0, 0, _ptr(res), _ptr(x), B, Cc, groups, S)
        return x

    def gn_pointwise_out(self, x, gn, w, bias, out):
        """final_conv on the RAW conv output x of the last ResnetBlock: gn = (stats, gamma, beta, groups, residual) as handed out by
        resnet(defer_gn=True); GroupNorm apply + SiLU + residual happen on the kernel's loads (sdc_gn_pointwise_out).  w = the conv
        weight (Cout, C) flat, out a 5-D view whose (H, W) planes are dense.  None when the shape is outside the kernel's contract."""
        st, gamma, beta, groups, res = gn
        B, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (B * Cc)
        cout = out.shape[1]
        plane = out.shape[3] * out.shape[4]
        ys = _s5(out)
        # (a thread walks all C channels of its four positions: worth it where a sample has >= 1024 positions -- the tokamak net's 128
        # positions per sample leave one 32-lane block per sample on a 256-channel walk: 150 us against ~30 for the two small kernels)
        ok = (x.is_contiguous() and (res is None or res.is_contiguous()) and cout <= 16 and Cc % 4 == 0 and Cc <= 2048 and S >= 1024 and plane % 4 == 0 and ys[4] == 1 and
              ys[3] == out.shape[4] and all(v % 4 == 0 for v in ys[:3]) and out.data_ptr() % 16 == 0 and
              tuple(out.shape[2:]) == tuple(x.shape[2:]))
        if not ok:
            return None
        self.keep += [x, st, gamma, beta, res, w, bias, out]
        self._emit(self.lib.sdc_gn_pointwise_out, _ptr(x), _ptr(st), _ptr(gamma), _ptr(beta), _ptr(res), _ptr(w), _ptr(bias), _ptr(out),
                   B, Cc, groups, cout, S, plane, ys[0], ys[1], ys[2])
        return out

    def linattn_block(self, x, g_pre, wqkv, wo, bo, g_post, outer, inner, n, strides, pre_mode, post_mode, eps=1e-5, gn=None):
        """Residual(PreNorm(LinearAttention)) in one call (dim 64 / 128, n % 64 == 0); returns y shaped like x.
        gn = (stats, gamma, beta, groups, residual): x is the RAW conv output of the producing ResnetBlock, whose GroupNorm +
        SiLU + residual add is applied on load (sdc_linattn_block_gn); `outer` must then be the batch axis."""
        Cc = x.shape[1]
        y = self.pool.get(tuple(x.shape))
        work = torch.empty(int(self.lib.sdc_linattn_block_bytes(outer, inner, Cc, n)) // 4, dtype=torch.float32, device=self.device)
        self.keep += [x, g_pre, wqkv, wo, bo, g_post, work, y]
        if gn is not None:
            st, gamma, beta, groups, res = gn
            assert outer == x.shape[0] and (res is None or (tuple(res.shape) == tuple(x.shape) and res.stride() == x.stride()))
            self.keep += [st, gamma, beta, res]
            self._emit(self.lib.sdc_linattn_block_gn, _ptr(x), _ptr(st), _ptr(gamma), _ptr(beta), groups, _ptr(res), _ptr(g_pre),
                       _ptr(wqkv), _ptr(wo), _ptr(bo), _ptr(g_post), _ptr(work), _ptr(y), outer, inner, Cc, n, *strides, pre_mode,
                       post_mode, eps)
            return y
        self._emit(self.lib.sdc_linattn_block, _ptr(x), _ptr(g_pre), _ptr(wqkv), _ptr(wo), _ptr(bo), _ptr(g_post), _ptr(work),
                   _ptr(y), outer, inner, Cc, n, *strides, pre_mode, post_mode, eps)
        return y

    def tattn_block(self, x, g_pre, wqkv, wo, rot, bias, eps=1e-5):
        """Residual(PreNorm(temporal Attention)) of the smoke net in one launch: x (B, 64, 32, H, W) contiguous."""
        B, Cc, Fr, H, W = x.shape
        assert x.is_contiguous()
        y = self.pool.get(tuple(x.shape))
        self.keep += [x, g_pre, wqkv, wo, rot, bias, y]
        self._emit(self.lib.sdc_tattn_block, _ptr(x), _ptr(g_pre), _ptr(wqkv), _ptr(wo), _ptr(rot), _ptr(bias), _ptr(y),
                   B, H * W, Cc, Fr, Cc * Fr * H * W, Fr * H * W, H * W, eps)
        return y

    def attn(self, qkv, out, heads, outer, inner, ntok, q_strides, o_strides, rot=None, bias=None):
        self.keep += [qkv, out, rot, bias]
        self._emit(self.lib.sdc_attn, _ptr(qkv), _ptr(out), _ptr(rot), _ptr(bias), outer, inner, heads, ntok,
                   *q_strides, *o_strides)
        return out
