"""Thin tensor-level wrappers of the backward (VJP) entry points of libsdc_hip.so (include/sdc.h, "backward").

PyTorch is plumbing here: device memory and the current stream.  Every function takes / returns fp32 CUDA (HIP) tensors
and launches asynchronously on torch's current stream; there is no CPU path.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import SdcPackItem, SdcWgradDesc, check
from .engine import as5


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("safediffcon_amd runs on MI355X only: tensors must be on a cuda (HIP) device; there is no CPU fallback")


def conv_wgrad(g, x, k, stride=(1, 1, 1), pad=(0, 0, 0), up=(1, 1, 1), bias=True):
    """dw[m][n][k...] = sum_{b,pos} g[b][m][pos] x[b][n][pos*s - p + tap]   (sdc_conv_wgrad)
    g (B, M, oD, oH, oW) and x (B, N, iD, iH, iW) are 5-D views (any strides); returns (dw (M, N, kD, kH, kW), dbias (M,) | None)."""
    _need_cuda(g, x)
    g, x = as5(g), as5(x)
    if tuple(k) == (1, 1, 1) and g.shape[2:] == (1, 1, 1) and x.shape[2:] == (1, 1, 1) and g.shape[0] > 1:
        # nn.Linear over a batch of rows: the batch is the position axis (16 rows per MFMA chunk instead of one)
        g = g.reshape(g.shape[0], g.shape[1]).t().reshape(1, g.shape[1], 1, 1, g.shape[0])
        x = x.reshape(x.shape[0], x.shape[1]).t().reshape(1, x.shape[1], 1, 1, x.shape[0])
    lib = _lib.get_lib()
    d = SdcWgradDesc()
    d.B, d.M, d.N = g.shape[0], g.shape[1], x.shape[1]
    d.oD, d.oH, d.oW = g.shape[2:]
    d.iD, d.iH, d.iW = x.shape[2:]
    d.kD, d.kH, d.kW = k
    d.sD, d.sH, d.sW = stride
    d.pD, d.pH, d.pW = pad
    d.uD, d.uH, d.uW = up
    d.gs[:] = tuple(int(s) for s in g.stride())
    d.xs[:] = tuple(int(s) for s in x.stride())
    nbytes = int(lib.sdc_conv_wgrad_bytes(C.byref(d)))
    work = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=g.device)
    dw = torch.empty((d.M, d.N, *k), dtype=torch.float32, device=g.device)
    db = torch.empty(d.M, dtype=torch.float32, device=g.device) if bias else None
    check(lib.sdc_conv_wgrad(C.byref(d), g.data_ptr(), x.data_ptr(), dw.data_ptr(), 0 if db is None else db.data_ptr(),
                             work.data_ptr(), nbytes, _stream(g)), "sdc_conv_wgrad")
    return dw, db


def gn_stats(h, groups, eps=1e-5):
    """(mean, rstd) table of sdc_gn_stats for a contiguous (B, C, ...) tensor"""
    _need_cuda(h)
    lib = _lib.get_lib()
    B, Cc = h.shape[0], h.shape[1]
    S = h.numel() // (B * Cc)
    st = torch.empty((int(lib.sdc_gn_stats_bytes(B, groups)) + 3) // 4, dtype=torch.float32, device=h.device)
    check(lib.sdc_gn_stats(h.data_ptr(), st.data_ptr(), B, Cc, groups, S, eps, _stream(h)), "sdc_gn_stats")
    return st


def gn_apply(h, st, gamma, beta, groups, ss=None, residual=None):
    """y = SiLU(GN(h) (scale + 1) + shift) (+ residual), out of place; ss (B, 2C) rows [scale | shift] or None"""
    lib = _lib.get_lib()
    B, Cc = h.shape[0], h.shape[1]
    S = h.numel() // (B * Cc)
    y = torch.empty_like(h)
    check(lib.sdc_gn_apply(h.data_ptr(), st.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0 if ss is None else ss.data_ptr(), 0,
                           0, 0 if ss is None else ss.stride(0), 0, 0 if residual is None else residual.data_ptr(), y.data_ptr(),
                           B, Cc, groups, S, _stream(h)), "sdc_gn_apply")
    return y


def gn_silu_bwd(h, gy, st, gamma, beta, groups, ss=None):
    """-> (gh, dgamma, dbeta, dss | None): backward of gn_apply (the residual's gradient is gy itself)"""
    _need_cuda(h, gy)
    lib = _lib.get_lib()
    B, Cc = h.shape[0], h.shape[1]
    S = h.numel() // (B * Cc)
    gy = gy.contiguous()
    buf = torch.empty(int(lib.sdc_gn_silu_bwd_floats(B, Cc, groups, S)), dtype=torch.float32, device=h.device)   # row sums + kernel scratch
    gh = torch.empty_like(h)
    dgb = torch.empty((2, Cc), dtype=torch.float32, device=h.device)
    dss = None if ss is None else torch.empty((B, 2 * Cc), dtype=torch.float32, device=h.device)
    check(lib.sdc_gn_silu_bwd(h.data_ptr(), gy.data_ptr(), st.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                              0 if ss is None else ss.data_ptr(), 0 if ss is None else ss.stride(0), buf.data_ptr(), gh.data_ptr(),
                              dgb[0].data_ptr(), dgb[1].data_ptr(), 0 if dss is None else dss.data_ptr(),
                              B, Cc, groups, S, _stream(h)), "sdc_gn_silu_bwd")
    return gh, dgb[0], dgb[1], dss


def chan_norm_bwd(x, gy, g, mode, eps=1e-5):
    """-> (gx, dgain): backward of sdc_chan_norm (mode 0 channel LayerNorm, 1 RMSNorm) for contiguous (B, C, ...) tensors"""
    _need_cuda(x, gy)
    lib = _lib.get_lib()
    B, Cc = x.shape[0], x.shape[1]
    S = x.numel() // (B * Cc)
    gy = gy.contiguous()
    nparts = int(lib.sdc_chan_norm_bwd_parts(B, S))
    gpart = torch.empty(int(lib.sdc_chan_norm_bwd_bytes(B, Cc, S)) // 4, dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    gv = g.reshape(-1).contiguous()
    check(lib.sdc_chan_norm_bwd(x.data_ptr(), gy.data_ptr(), gv.data_ptr(), gx.data_ptr(), gpart.data_ptr(), B, Cc, S, mode, eps,
                                _stream(x)), "sdc_chan_norm_bwd")
    return gx, gpart[: Cc * nparts].view(Cc, nparts).sum(1)


def act_bwd(x, gy, kind):
    lib = _lib.get_lib()
    gy = gy.contiguous()
    gx = torch.empty_like(x)
    check(lib.sdc_act_bwd(x.data_ptr(), gy.data_ptr(), gx.data_ptr(), x.numel(), kind, _stream(x)), "sdc_act_bwd")
    return gx


def sumpool2(g, fh, fw):
    """VJP of nearest upsampling by (fh, fw) over the last two axes of a contiguous tensor"""
    lib = _lib.get_lib()
    g = g.contiguous()
    H, W = g.shape[-2] // fh, g.shape[-1] // fw
    out = torch.empty((*g.shape[:-2], H, W), dtype=torch.float32, device=g.device)
    rows = out.numel() // (H * W)
    check(lib.sdc_sumpool2(g.data_ptr(), out.data_ptr(), rows, H, W, fh, fw, _stream(g)), "sdc_sumpool2")
    return out


def chan_norm(x, g, mode, eps=1e-5):
    """channel LayerNorm (mode 0) / RMSNorm (mode 1) of a contiguous (B, C, ...) tensor (sdc_chan_norm)"""
    _need_cuda(x)
    lib = _lib.get_lib()
    B, Cc = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    gv = g.reshape(-1).contiguous()
    check(lib.sdc_chan_norm(x.data_ptr(), gv.data_ptr(), 0, y.data_ptr(), B, Cc, x.numel() // (B * Cc), mode, eps, _stream(x)), "sdc_chan_norm")
    return y


def attn_core(qkv, out, heads, outer, inner, ntok, qs, os_, rot=None, bias=None):
    """softmax attention core (sdc_attn): strides (so, sc, si, st) of qkv and out in elements"""
    check(_lib.get_lib().sdc_attn(qkv.data_ptr(), out.data_ptr(), 0 if rot is None else rot.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                  outer, inner, heads, ntok, *qs, *os_, _stream(qkv)), "sdc_attn")
    return out


def attn_core_bwd(qkv, dout, heads, outer, inner, ntok, qs, os_, rot=None, bias=None):
    """-> (dqkv, dbias | None)"""
    lib = _lib.get_lib()
    dqkv = torch.empty_like(qkv)
    dbias = work = None
    if bias is not None:
        dbias = torch.empty_like(bias)
        work = torch.empty(int(lib.sdc_attn_bwd_bytes(outer, inner, heads, ntok)) // 4, dtype=torch.float32, device=qkv.device)
    check(lib.sdc_attn_bwd(qkv.data_ptr(), dout.data_ptr(), 0 if rot is None else rot.data_ptr(), 0 if bias is None else bias.data_ptr(),
                           dqkv.data_ptr(), 0 if dbias is None else dbias.data_ptr(), 0 if work is None else work.data_ptr(),
                           outer, inner, heads, ntok, *qs, *os_, _stream(qkv)), "sdc_attn_bwd")
    return dqkv, dbias


def linattn_core(qkv, out, heads, outer, inner, n, qs, os_):
    """linear attention core (sdc_linattn): strides (so, sc, si) of qkv and out"""
    lib = _lib.get_lib()
    ctx = torch.empty(outer * inner * heads * 32 * 32, dtype=torch.float32, device=qkv.device)
    check(lib.sdc_linattn(qkv.data_ptr(), ctx.data_ptr(), out.data_ptr(), outer, inner, heads, n, *qs, *os_, _stream(qkv)), "sdc_linattn")
    return out


def linattn_core_bwd(qkv, dout, heads, outer, inner, n, qs, os_):
    lib = _lib.get_lib()
    dqkv = torch.empty_like(qkv)
    work = torch.empty(int(lib.sdc_linattn_bwd_bytes(outer, inner, heads, n)) // 4, dtype=torch.float32, device=qkv.device)
    check(lib.sdc_linattn_bwd(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), work.data_ptr(), outer, inner, heads, n, *qs, *os_,
                              _stream(qkv)), "sdc_linattn_bwd")
    return dqkv


class PackArena:
    """The packed conv weights of ONE net, refreshed by one launch per training forward (sdc_pack_batch_run).

    A fine-tuning step needs every conv weight in kernel layout twice (forward form, data-gradient form) and the optimiser
    changes them between steps, so they are re-packed every step: 187 launches of ~13 us on the tokamak net.  The arena
    records which (parameter, precision, flip) layouts a step asked for; from the next step on ``begin()`` packs all of
    them in one launch into one buffer and ``get()`` hands out slices.  Only ``nn.Parameter`` storage is cached (a weight
    computed inside the graph has no stable address); the arena holds a reference to every source tensor, so a table entry
    never points at freed memory, and entries nobody asked for in two consecutive steps are dropped -- unless a captured
    graph replays the arena's launch (`freeze`): then buffer and table stay exactly as captured."""

    def __init__(self):
        self.index = {}        # key -> (offset, numel) in self.buf
        self.meta = {}         # key -> (source tensor, precision, flip)
        self.idle = {}         # key -> steps since the entry was last asked for
        self.pending = {}      # asked for, not in the table yet
        self.used = set()
        self.buf = self.table = None
        self.launch = None     # (n, total_blocks, lds_bytes)
        self.fresh = False     # the buffer holds this step's weights
        self.frozen = 0        # captured graphs that replay sdc_pack_batch_run on this buf / table (GraphedLossStep)

    def freeze(self):
        """a captured graph has baked in the addresses of buf and table: from now on the arena neither drops nor reallocates
        them (layouts asked for later are packed one by one by the caller, as in the first step) until every holder thaws"""
        self.frozen += 1
        return self.buf, self.table

    def thaw(self):
        self.frozen = max(0, self.frozen - 1)

    @staticmethod
    def cacheable(w):
        b = w._base if w._base is not None else w
        return isinstance(b, torch.nn.Parameter) and w.is_contiguous() and w.is_cuda

    @staticmethod
    def _key(w5, precision, flip):
        return (w5.data_ptr(), tuple(w5.shape), int(precision), bool(flip))

    def _rebuild(self, device):
        lib = _lib.get_lib()
        for k in [k for k, n in self.idle.items() if n >= 2]:
            del self.meta[k], self.idle[k]
        for k, m in self.pending.items():
            self.meta[k] = m
            self.idle[k] = 0
        self.pending = {}
        self.index, off = {}, 0
        items = (SdcPackItem * max(len(self.meta), 1))()
        for it, (k, (w5, prec, flip)) in zip(items, self.meta.items()):
            co, ci, kD, kH, kW = w5.shape
            if flip:
                co, ci = ci, co
            n = int(lib.sdc_pack_conv_weight_floats(co, ci, kD, kH, kW, prec))
            self.index[k] = (off, n)
            it.w = w5.data_ptr()
            it.Cout, it.Cin, it.kD, it.kH, it.kW, it.precision, it.flip = co, ci, kD, kH, kW, prec, 1 if flip else 0
            it.out = off                                   # offset in floats for now
            off += (n + 3) & ~3                            # 16-byte aligned sections
        if not self.meta:
            self.buf = self.table = self.launch = None
            return
        self.buf = torch.empty(off, dtype=torch.float32, device=device)
        base = self.buf.data_ptr()
        for it in items:
            it.out = base + 4 * (it.out or 0)
        nb, lds = C.c_int(0), C.c_int(0)
        check(lib.sdc_pack_batch_plan(items, len(self.meta), C.byref(nb), C.byref(lds)), "sdc_pack_batch_plan")
        self.table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
        self.launch = (len(self.meta), nb.value, lds.value)

    def begin(self, device):
        """start of a training forward: fold the last step's requests into the table, then pack everything once"""
        for k in self.idle:
            self.idle[k] = 0 if k in self.used else self.idle[k] + 1
        self.used = set()
        if not self.frozen and (self.pending or any(n >= 2 for n in self.idle.values())):
            self._rebuild(device)
        self.fresh = False
        if self.table is not None:
            n, nb, lds = self.launch
            cur = torch.cuda.current_stream(device)
            # the buffer may have been allocated under another stream (the sampler's private one in the differentiable last DDIM
            # step, the default one in p_losses): tell the allocator about every stream that packs or reads it, so that a later
            # _rebuild cannot hand its memory out while this stream's work is pending
            self.buf.record_stream(cur)
            self.table.record_stream(cur)
            check(_lib.get_lib().sdc_pack_batch_run(self.table.data_ptr(), n, nb, lds, cur.cuda_stream), "sdc_pack_batch_run")
            self.fresh = True

    def get(self, w5, precision, flip):
        """this step's packed layout, or None (then the caller packs it alone and the next step's table holds it)"""
        k = self._key(w5, precision, flip)
        ent = self.index.get(k)
        if ent is not None and self.fresh and self.buf.device == w5.device:
            self.used.add(k)
            self.buf.record_stream(torch.cuda.current_stream(w5.device))
            return self.buf[ent[0]:ent[0] + ent[1]]
        if k not in self.meta and not self.frozen:       # (frozen: no table change is coming, do not pin the source tensor)
            self.pending[k] = (w5, int(precision), bool(flip))
        return None


def pack_conv_weight(w, precision, flip=False, arena=None):
    """kernel layout of a contiguous nn.Conv weight (Cout, Cin, *k) on the device in one launch (sdc_pack_conv_weight);
    flip: the data-gradient weight (transposed channels, flipped taps).  arena: the net's PackArena when w is parameter storage."""
    _need_cuda(w)
    lib = _lib.get_lib()
    if arena is not None:
        hit = arena.get(as5(w.detach()), precision, flip)
        if hit is not None:
            return hit
    w5 = as5(w.detach()).contiguous()
    co, ci, kD, kH, kW = w5.shape
    if flip:
        co, ci = ci, co
    out = torch.empty(int(lib.sdc_pack_conv_weight_floats(co, ci, kD, kH, kW, precision)), dtype=torch.float32, device=w.device)
    check(lib.sdc_pack_conv_weight(w5.data_ptr(), out.data_ptr(), co, ci, kD, kH, kW, precision, 1 if flip else 0, _stream(w)),
          "sdc_pack_conv_weight")
    return out
