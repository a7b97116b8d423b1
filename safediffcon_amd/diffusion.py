"""Drop-in ``GaussianDiffusion`` samplers for the three SafeDiffCon tasks.

  GaussianDiffusionBurgers   <- 1D/model/diffusion.py:21-607      (sample / p_sample_loop)
  GaussianDiffusionTokamak   <- tokamak/model/diffusion.py:19-539
  GaussianDiffusionSmoke     <- 2d/ddpm/diffusion_2d.py:111-414

Same constructor keywords and ``sample(...)`` signatures as the reference.  One
denoising step = the U-Net plan + sdc_guide_reduce + sdc_step_update +
sdc_advance, all on the caller's HIP stream; with in-kernel Philox noise the
whole step is captured once into a hipGraph and replayed for every timestep
(the timestep and the noise-draw counter live in device memory).

Guidance: the three closed forms the reference's pipelines use are described by
``GuidanceSpec`` objects (callable, so they also work as plain ``nablaJ`` /
``design_fn`` closures); the sampler recognises them and fuses the gradient into
the update kernel.  Any other callable is honoured too: the step is then split
around the callable (x0 kernel -> callable on device tensors -> update kernel).
"""
import ctypes as C
import math

import torch
from torch import nn

from . import _lib
from ._lib import SdcStepDesc, check
from .engine import Plan

_MODEL = {"burgers": _lib.SDC_MODEL_BURGERS, "tokamak": _lib.SDC_MODEL_TOKAMAK, "smoke": _lib.SDC_MODEL_SMOKE}

BURGERS_SCALER = 10.0                                               # 1D/utils/common.py:17
TOKAMAK_SCALER = (2, 7, 2, 1, 2, 2, 2, 2, 1, 1, 2, 3)              # tokamak/utils/common.py:16
SMOKE_RESCALER = (2, 19, 20, 17, 20, 1, 1)                          # 2d/ddpm/data_2d.py:38


# --------------------------------------------------------------------------- schedules (fp64 on host, cast at the end)
def _betas(kind, T):
    if kind == "linear":                                           # 1D/model/model_utils.py:142-146
        s = 1000.0 / T
        return torch.linspace(s * 1e-4, s * 2e-2, T, dtype=torch.float64)
    if kind == "cosine":                                           # 1D/model/model_utils.py:148-158
        x = torch.linspace(0, T, T + 1, dtype=torch.float64)
        ac = torch.cos(((x / T) + 0.008) / 1.008 * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)
    if kind == "sigmoid":                                          # 2d/ddpm/diffusion_2d.py:95-108 (start -3, end 3, tau 1)
        t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
        v0, v1 = torch.tensor(-3.0).sigmoid(), torch.tensor(3.0).sigmoid()
        ac = (-(t * 6 - 3).sigmoid() + v1) / (v1 - v0)
        ac = ac / ac[0]
        return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)
    raise ValueError(f"unknown beta schedule {kind}")


def schedule_tables(kind, T):
    """The reference's registered buffers (1D/model/diffusion.py:111-156), fp32."""
    b = _betas(kind, T)
    a = 1.0 - b
    ac = torch.cumprod(a, dim=0)
    acp = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])
    pv = b * (1.0 - acp) / (1.0 - ac)
    tabs = dict(betas=b, alphas_cumprod=ac, alphas_cumprod_prev=acp, sqrt_alphas_cumprod=ac.sqrt(),
                sqrt_one_minus_alphas_cumprod=(1.0 - ac).sqrt(), log_one_minus_alphas_cumprod=(1.0 - ac).log(),
                sqrt_recip_alphas_cumprod=(1.0 / ac).sqrt(), sqrt_recipm1_alphas_cumprod=(1.0 / ac - 1).sqrt(),
                posterior_variance=pv, posterior_log_variance_clipped=pv.clamp(min=1e-20).log(),
                posterior_mean_coef1=b * acp.sqrt() / (1.0 - ac), posterior_mean_coef2=(1.0 - acp) * a.sqrt() / (1.0 - ac))
    return {k: v.to(torch.float32) for k, v in tabs.items()}


# --------------------------------------------------------------------------- guidance specs
class GuidanceSpec:
    """Closed-form guidance recognised by the fused update kernel.  Also a plain callable
    (x0 -> dJ/dx0 via autograd on the device), so it can be passed wherever the reference takes
    ``nablaJ`` / ``design_fn``."""
    kind = None

    def gpar(self):
        raise NotImplementedError

    def J(self, x):
        raise NotImplementedError

    def __call__(self, x):
        with torch.enable_grad():
            x = x.detach().requires_grad_()
            j = self.J(x)
            return torch.autograd.grad(j, x, grad_outputs=torch.ones_like(j))[0]

    @staticmethod
    def _f(v):
        return float(v.item()) if isinstance(v, torch.Tensor) else float(v)


class BurgersGuidance(GuidanceSpec):
    """J = w_score * max(f(10*x0[:,2,:11,:]) + Q - u_bound^2, 0), f = mean if use_max_safety else amax.
    1D/utils/guidance.py:58-85 (get_finetune_guidance).  ``Q`` may be reassigned between samples."""
    kind = "burgers"

    def __init__(self, Q, w_score, u_bound, use_max_safety=True):
        self.Q, self.w_score, self.u_bound, self.use_max_safety = Q, w_score, u_bound, use_max_safety

    def gpar(self):
        return [self._f(self.w_score), self._f(self.u_bound) ** 2, self._f(self.Q), BURGERS_SCALER]

    def J(self, x):
        s = (x * BURGERS_SCALER)[:, 2, :11, :]
        s = s.mean(dim=(-1, -2)) if self.use_max_safety else s.amax(dim=(-1, -2))
        return torch.clamp(s + self._f(self.Q) - self._f(self.u_bound) ** 2, min=0) * self._f(self.w_score)


class TokamakGuidance(GuidanceSpec):
    """loss = scaler * (w_obj*(mse(beta_p)+mse(l_i)) + w_safe*max(thr - min_t q95 + Q, 0)).
    tokamak/utils/guidance.py:32-73; ``target`` (B,3,nt) is cached once (the reference reloads it every step)."""
    kind = "tokamak"

    def __init__(self, target, nt=122, w_obj=0.0, w_safe=0.0, guidance_scaler=1.0, Q=0.0, safety_threshold=5.0):
        self.target, self.nt = target, nt
        self.w_obj, self.w_safe, self.guidance_scaler, self.Q, self.safety_threshold = w_obj, w_safe, guidance_scaler, Q, safety_threshold

    def gpar(self):
        return [self._f(self.w_obj), self._f(self.w_safe), self._f(self.guidance_scaler), self._f(self.safety_threshold),
                self._f(self.Q)]

    def J(self, x):
        sc = torch.tensor(TOKAMAK_SCALER, dtype=x.dtype, device=x.device).reshape(12, 1)
        st = (x * sc)[:, :3, :self.nt]
        tg = self.target.to(x.device)
        obj = (st[:, 0] - tg[:, 0]).square().mean(-1) + (st[:, 2] - tg[:, 2]).square().mean(-1)
        s = st[:, 1].amin(dim=-1)
        safe = torch.clamp(self._f(self.safety_threshold) - s + self._f(self.Q), min=0)
        return (self._f(self.w_obj) * obj + self._f(self.w_safe) * safe) * self._f(self.guidance_scaler)


class SmokeGuidance(GuidanceSpec):
    """guidance = -(1-w_safe)*mean(R5*x[:,:,5]) + w_safe*max(mean(R6*x[:,-1,6]) + Q - safe_bound, 0).
    2d/inference_2d.py:173-195 (InferencePipeline.guidance / design_fn)."""
    kind = "smoke"

    def __init__(self, Q, w_safe, safe_bound, standard_fixed_ratio=1.0):
        self.Q, self.w_safe, self.safe_bound, self.standard_fixed_ratio = Q, w_safe, safe_bound, standard_fixed_ratio

    def gpar(self):
        return [self._f(self.w_safe), self._f(self.safe_bound), self._f(self.Q), self._f(self.standard_fixed_ratio)]

    def J(self, x):
        R = torch.tensor(SMOKE_RESCALER, dtype=x.dtype, device=x.device).reshape(1, 1, 7, 1, 1)
        st = x * R
        succ = st[:, :, 5].mean((-1, -2, -3))
        safe = torch.clamp(st[:, -1, 6].mean((-1, -2)) + self._f(self.Q) - self._f(self.safe_bound), min=0)
        return -(1 - self._f(self.w_safe)) * succ + self._f(self.w_safe) * safe


def _splitmix64(v):
    v = (v + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return v ^ (v >> 31)


def _mix_rank(seed, rank=None):
    """seed ^ splitmix64(rank) for rank > 0 (rank 0 / single process: unchanged)"""
    if rank is None:
        import torch.distributed as dist
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    return seed if rank == 0 else (seed ^ _splitmix64(rank)) & ((1 << 62) - 1)


# --------------------------------------------------------------------------- sampler core
class _State:
    """Per-(batch size) device state: step counters, guidance scalars, condition buffers, captured graphs."""

    def __init__(self, dev, B, per, T):
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.idx_dev = torch.zeros(1, dtype=torch.int32, device=dev)     # DDIM: step number (row of coef / ttab)
        self.draw_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gpar = z(8)
        self.gscal = z(4 * B)
        self.noise = None            # single-slot injected-noise buffer
        self.x0 = None
        self.cond = {}               # name -> persistent device buffer


class _SamplerBase(nn.Module):
    MODEL = None

    def _init_common(self, model, timesteps, sampling_timesteps, beta_schedule, ddim_sampling_eta):
        self.model = model
        self.channels = model.channels
        self.self_condition = model.self_condition
        tabs = schedule_tables(beta_schedule, timesteps)
        self.num_timesteps = int(timesteps)
        self.sampling_timesteps = timesteps if sampling_timesteps is None else sampling_timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for k, v in tabs.items():
            self.register_buffer(k, v)
        self.register_buffer("loss_weight", torch.ones(timesteps))
        self._states = {}
        self._lib = _lib.get_lib()
        self.use_graph = True

    # -------------------------------------------------------------- fine-tuning loss
    def q_sample(self, x_start, t, noise=None):
        """1D/model/diffusion.py:629-636, 2d/ddpm/diffusion_2d.py:416-423"""
        noise = torch.randn_like(x_start) if noise is None else noise
        bshape = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
        return self.sqrt_alphas_cumprod[t].reshape(bshape) * x_start + self.sqrt_one_minus_alphas_cumprod[t].reshape(bshape) * noise

    def _loss_tail(self, model_out, target, t, mean, kind="l2"):
        loss = (model_out - target) ** 2 if kind == "l2" else (model_out - target).abs()
        loss = loss.flatten(1).mean(1)                       # reduce(loss, 'b ... -> b', 'mean')
        if self.MODEL != "smoke":                            # loss_weight = 1 for pred_noise (1D/model/diffusion.py:160-166)
            loss = loss * self.loss_weight[t]
        return loss.mean() if mean else loss

    def forward(self, img, *args, **kwargs):
        """diffusion(state[, mean=False]) -> loss at random timesteps: 1D/model/diffusion.py:735-747, 2d/ddpm/diffusion_2d.py:454-458"""
        t = torch.randint(0, self.num_timesteps, (img.shape[0],), device=img.device).long()
        return self.p_losses(img, t, *args, **kwargs)

    # -------------------------------------------------------------- helpers
    def _coef(self, J_scheduler, k_const=1.0):
        """[T][8] = {a, b, c1, c2, sigma, k, 0, 0}; sigma = exp(0.5*logvar) (0 at t=0: 'no noise if t == 0')."""
        T = self.num_timesteps
        c = torch.zeros(T, 8, dtype=torch.float32)
        c[:, 0] = self.sqrt_recip_alphas_cumprod.cpu()
        c[:, 1] = self.sqrt_recipm1_alphas_cumprod.cpu()
        c[:, 2] = self.posterior_mean_coef1.cpu()
        c[:, 3] = self.posterior_mean_coef2.cpu()
        c[:, 4] = (0.5 * self.posterior_log_variance_clipped.cpu()).exp()
        c[0, 4] = 0.0
        if J_scheduler is None:
            c[:, 5] = k_const
        else:
            c[:, 5] = torch.tensor([float(J_scheduler(t)) for t in range(T)]) * k_const
        return c.to(self.betas.device)

    def _coef_ddim(self, J_scheduler, k_const=1.0):
        """DDIM rows in step order, {a, b, sqrt(alpha_next), c, sigma, k, last, 0}, and the timestep of every step.
        Scalars are formed with fp32 tensor arithmetic exactly like ddim_sample does (1D/model/diffusion.py:500-504)."""
        T, S, eta = self.num_timesteps, self.sampling_timesteps, self.ddim_sampling_eta
        times = torch.linspace(-1, T - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        pairs = list(zip(times[:-1], times[1:]))
        ac = self.alphas_cumprod.cpu()
        c = torch.zeros(len(pairs), 8, dtype=torch.float32)
        for i, (time, nxt) in enumerate(pairs):
            c[i, 0] = self.sqrt_recip_alphas_cumprod.cpu()[time]
            c[i, 1] = self.sqrt_recipm1_alphas_cumprod.cpu()[time]
            c[i, 5] = (float(J_scheduler(time)) if J_scheduler is not None else 1.0) * k_const
            if nxt < 0:
                c[i, 6] = 1.0
                continue
            alpha, alpha_next = ac[time], ac[nxt]
            sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c[i, 2] = alpha_next.sqrt()
            c[i, 3] = (1 - alpha_next - sigma ** 2).sqrt()
            c[i, 4] = sigma
        ttab = torch.tensor([p[0] for p in pairs] + [0], dtype=torch.int32)
        return c.to(self.betas.device), ttab.to(self.betas.device)

    def _state(self, B, per):
        key = (B, per, str(self.betas.device))
        if key not in self._states:
            self._states[key] = _State(self.betas.device, B, per, self.num_timesteps)
        return self._states[key]

    def _stream(self):
        return torch.cuda.current_stream(self.betas.device).cuda_stream

    def _desc(self, B, dims, **kw):
        d = SdcStepDesc()
        d.model, d.B = _MODEL[self.MODEL], B
        d.d0, d.d1, d.d2, d.d3 = dims
        d.guide = d.impose = d.pad_zero = d.use_max = d.has_wgt = d.skip_draws = 0
        d.clip, d.cond_idx, d.seed = 1, 0, 0
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    def _persist(self, st, name, t):
        """copy a user condition tensor into a persistent device buffer (stable pointer for the graph)."""
        if t is None:
            return None
        t = t.detach().to(self.betas.device, torch.float32).contiguous()
        buf = st.cond.get(name)
        if buf is None or buf.shape != t.shape:
            buf = torch.empty_like(t)
            st.cond[name] = buf
        buf.copy_(t)
        return buf

    def _tail(self, st, ent, d, coef, guide_spec, target, c0, c1, c2, noise_buf, advance=True, ttab=None):
        """Plan with the post-U-Net part of one step: [guide_reduce] -> step_update (x in place) -> advance.
        DDPM: coefficient row = timestep (st.t_dev); DDIM (ttab given): row = step number (st.idx_dev)."""
        p = Plan(self.betas.device)
        lib = self._lib
        x, eps = ent["x"], ent["eps"]
        ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        p.keep += [d, coef, x, eps, c0, c1, c2, target, noise_buf, ttab]
        row = st.t_dev if ttab is None else st.idx_dev
        if d.guide == 1:
            p._emit(lib.sdc_guide_reduce, C.byref(d), ptr(x), ptr(eps), ptr(coef), ptr(row), ptr(st.gpar), ptr(st.gscal))
        p._emit(lib.sdc_step_update, C.byref(d), ptr(x), ptr(eps), 0, ptr(coef), ptr(row), ptr(st.draw_dev),
                ptr(noise_buf), 0, ptr(st.gpar), ptr(st.gscal), ptr(target), ptr(c0), ptr(c1), ptr(c2), ptr(x), 0)
        if advance and ttab is None:
            p._emit(lib.sdc_advance, ptr(st.t_dev), -1, ptr(st.draw_dev), 1 + d.skip_draws)
        elif advance:
            p._emit(lib.sdc_advance_table, ptr(st.idx_dev), ptr(st.t_dev), ptr(ttab), ptr(st.draw_dev), 1)
        return p

    def _dispatch(self, prepare, B, dims, **kw):
        """sample() runs the whole reverse process; sample(..., _prepare=True) only binds it and returns the _Loop
        (init/step/final/close) -- used by bench.py to time single denoising steps; the caller owns the stream."""
        if prepare:
            kw.setdefault("target", None)
            kw.setdefault("final_update", True)
            kw.pop("grad_tail", None)
            return self._setup(B, dims, **kw)
        return self._reverse_loop(B, dims, **kw)

    def _reverse_loop(self, B, dims, *, noise, guide, J_scheduler, k_const, cond, flags, impose_last, target=None,
                      final_update=True, ddim=False, control_at_end=False, grad_tail=False):
        """Shared DDPM loop.  `guide`: None | GuidanceSpec | callable.  `cond`: (c0, c1, c2) tensors or None.
        `noise`: None (Philox in-kernel) or callable i -> tensor (injected, parity runs)."""
        dev = self.betas.device
        if dev.type != "cuda":
            raise RuntimeError("safediffcon_amd samplers run on MI355X only: call .to('cuda') first; no CPU fallback")
        # hipGraph capture needs a non-default stream: run the whole loop on a private stream that is
        # ordered after / before the caller's current stream
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_side", None) is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            out = self._reverse_loop_on_stream(B, dims, noise=noise, guide=guide, J_scheduler=J_scheduler, k_const=k_const,
                                               cond=cond, flags=flags, impose_last=impose_last, target=target,
                                               final_update=final_update, ddim=ddim, control_at_end=control_at_end,
                                               grad_tail=grad_tail)
        cur.wait_stream(self._side)
        return out

    def _reverse_loop_on_stream(self, B, dims, *, noise, guide, J_scheduler, k_const, cond, flags, impose_last, target,
                                final_update, ddim, control_at_end, grad_tail=False):
        S = self._setup(B, dims, noise=noise, guide=guide, J_scheduler=J_scheduler, k_const=k_const, cond=cond,
                        flags=flags, impose_last=impose_last, target=target, final_update=final_update, ddim=ddim,
                        control_at_end=control_at_end)
        # enable_grad: the reference runs its LAST step inside torch.enable_grad() so that the fine-tuning loss can be
        # back-propagated through one U-Net evaluation (1D/model/diffusion.py:524-551, 2d/ddpm/diffusion_2d.py:314-322,:379-399)
        grad_tail = grad_tail and any(p.requires_grad for p in self.model.parameters())
        try:
            S.init()
            n_main = S.nsteps - 1 if grad_tail else S.n_main
            for _ in range(n_main):
                S.step()
            if grad_tail:
                return self._differentiable_last_step(S, guide, cond)
            S.final()
            return S.x.clone()
        finally:
            S.close()

    def _differentiable_last_step(self, S, guide, cond):
        """the last reverse step as an autograd graph over the model parameters: eps through the differentiable HIP U-Net
        (model.forward_train), the x0 / guidance / posterior arithmetic of sdc_step_update restated element-wise"""
        x = S.x.clone().view(S.shape)
        row = S.coef[S.nsteps - 1] if S.ddim else S.coef[0]
        t_last = int(S.ttab[S.nsteps - 1].item()) if S.ddim else 0
        a, b, k = row[0], row[1], row[5]
        from .autograd import batch_invariant
        with torch.enable_grad():
            with batch_invariant():           # no batch-dependent channel split: a trajectory's bits do not depend on its batch
                eps = self.model.forward_train(x, torch.full((S.B,), t_last, device=x.device, dtype=torch.long))
            x0 = (a * x - b * eps).clamp(-1.0, 1.0)
            if guide is not None:
                g = guide(x0.detach().clone().requires_grad_())        # (the reference differentiates J on a detached clone)
                g = g.detach() if isinstance(g, torch.Tensor) else g
                x0 = (a * x - b * (eps + g * k)).clamp(-1.0, 1.0)
            if S.ddim:
                out = x0                                               # time_next < 0: img = x_start
            else:
                out = row[2] * x0 + row[3] * x                         # p_sample at t = 0: posterior mean, no noise
            out = out.clone()
            if self.MODEL == "smoke":
                c0, c1 = cond[0], cond[1]
                if not S.ddim:
                    out[:, 0, 0] = c0.to(out.device)
                if c1 is not None:
                    out[:, :, 3:5] = c1.to(out.device)
        return out

    def _setup(self, B, dims, *, noise, guide, J_scheduler, k_const, cond, flags, impose_last, target=None,
               final_update=True, ddim=False, control_at_end=False):
        """Bind everything one sample() needs and return a _Loop: init() draws x_T and imposes the conditions,
        step() runs one denoising step (a hipGraph launch when possible), final() the last, un-imposed step."""
        dev = self.betas.device
        T = self.num_timesteps
        per = int(math.prod(dims))
        shape = self._sample_shape(B)
        st = self._state(B, per)
        net = self.model
        ent = net.entry(shape, T, lut=True)
        stream = self._stream()
        if not ent["lut_valid"]:
            net.fill_cond(ent, torch.arange(T), stream)
            ent["lut_valid"] = True
        if ent["t_dev"] is not st.t_dev:
            net.bind_cond(ent, st.t_dev)
        L = _Loop()
        L.gd, L.st, L.ent, L.x, L.eps, L.T, L.B, L.shape = self, st, ent, ent["x"], ent["eps"], T, B, shape
        L.ddim, L.ttab, L.control_at_end = ddim, None, control_at_end
        if ddim:
            L.coef, L.ttab = self._coef_ddim(J_scheduler, k_const)
            flags = {**flags, "ddim": 1}
            impose_last = False                      # the last DDIM step returns x_start un-imposed (:495-498)
        else:
            L.coef = self._coef(J_scheduler, k_const)
        L.nsteps = int(L.coef.shape[0])
        L.c0 = self._persist(st, "c0", cond[0])
        L.c1 = self._persist(st, "c1", cond[1])
        L.c2 = self._persist(st, "c2", cond[2])
        L.tgt = self._persist(st, "target", target)
        L.fused = isinstance(guide, GuidanceSpec) and guide.kind == self.MODEL
        L.external = (guide is not None) and not L.fused
        L.guide = guide
        if L.fused:
            gp = guide.gpar()
            st.gpar.copy_(torch.tensor(gp + [0.0] * (8 - len(gp)), dtype=torch.float32))
        # Philox key: drawn from torch's global CPU generator (so torch.manual_seed reproduces a run) and mixed with the
        # process rank: ranks of a batch-sharded run that all call torch.manual_seed(cfg.seed) must not sample their
        # shards from one and the same noise stream (dist.py)
        L.seed = _mix_rank(int(torch.randint(0, 2 ** 62, (1,)).item()))
        L.noise = noise
        if noise is not None and st.noise is None:
            st.noise = torch.empty(B * per, dtype=torch.float32, device=dev)
        L.nbuf = st.noise if noise is not None else None
        L.skip = flags.get("skip_draws", 0)
        L.impose_last, L.final_update = impose_last, final_update or ddim
        L.n_main = L.nsteps if impose_last else L.nsteps - 1
        L.mk = lambda **kw: self._desc(B, dims, **{**flags, "seed": L.seed, **kw})  # noqa: E731
        L.d_init = L.mk(impose=1)
        if not L.external:
            gmode = 1 if L.fused else 0
            L.tail = self._tail(st, ent, L.mk(guide=gmode, impose=1), L.coef, guide, L.tgt, L.c0, L.c1, L.c2, L.nbuf,
                                ttab=L.ttab)
            L.tail_last = self._tail(st, ent, L.mk(guide=gmode, impose=1 if impose_last else 0), L.coef, guide, L.tgt,
                                     L.c0, L.c1, L.c2, L.nbuf, ttab=L.ttab)
        else:
            if st.x0 is None:
                st.x0 = torch.empty_like(L.x)
            L.d_x0 = L.mk(guide=3, impose=0)
        L.use_graph = self.use_graph and noise is None and not L.external
        return L


class _Loop:
    """One bound reverse process (see _SamplerBase._setup)."""
    graph = None
    draw_i = 1
    t_host = 0

    def _ptr(self, t):
        return 0 if t is None else t.data_ptr()

    def init(self):
        lib, st, x, p = self.gd._lib, self.st, self.x, self._ptr
        stream = self.gd._stream()
        if self.noise is not None:
            x.copy_(self.noise(0).to(x.device).reshape(x.shape))
        else:
            st.draw_dev.zero_()
            check(lib.sdc_randn(x.data_ptr(), x.numel(), self.seed, st.draw_dev.data_ptr(), stream), "sdc_randn")
        check(lib.sdc_impose(C.byref(self.d_init), x.data_ptr(), p(self.c0), p(self.c1), p(self.c2), stream), "sdc_impose")
        if self.ddim:
            st.idx_dev.zero_()
            st.t_dev.copy_(self.ttab[:1])
        else:
            st.t_dev.fill_(self.T - 1)
        st.draw_dev.fill_(1)
        self.draw_i, self.t_host = 1, self.nsteps - 1            # t_host counts the steps still to run, minus one
        if self.use_graph and self.graph is None:
            # one capture per bound loop: the descriptors carry this call's Philox seed
            check(lib.sdc_graph_begin(stream), "sdc_graph_begin")
            try:
                self.ent["plan"].run(stream)
                self.tail.run(stream)
            finally:
                g = C.c_void_p()
                rc = lib.sdc_graph_end(stream, C.byref(g))
            check(rc, "sdc_graph_end")
            self.graph = g

    def _feed_noise(self):
        if self.noise is not None and self.t_host > 0:
            for _ in range(self.skip):
                self.noise(self.draw_i)            # the calibration branch's discarded draw
                self.draw_i += 1
            self.nbuf.copy_(self.noise(self.draw_i).to(self.nbuf.device).reshape(-1))
            self.draw_i += 1

    def step(self, last=False):
        """one denoising step at the current device-side timestep"""
        lib, stream = self.gd._lib, self.gd._stream()
        if self.external:
            return self._step_external(last)
        if self.graph is not None and not last:
            check(lib.sdc_graph_launch(self.graph, stream), "sdc_graph_launch")
        else:
            self._feed_noise()
            self.ent["plan"].run(stream)
            if not last:
                self.tail.run(stream)
            elif self.final_update:
                self.tail_last.run(stream)
        self.t_host -= 1

    def _step_external(self, last):
        # arbitrary guidance callable: x0 kernel -> callable (torch, on device) -> update kernel
        lib, st, p, stream = self.gd._lib, self.st, self._ptr, self.gd._stream()
        x, eps = self.x, self.eps
        row = st.idx_dev if self.ddim else st.t_dev
        self._feed_noise()
        self.ent["plan"].run(stream)
        check(lib.sdc_step_update(C.byref(self.d_x0), p(x), p(eps), 0, p(self.coef), p(row), p(st.draw_dev), 0, 0, 0,
                                  0, 0, 0, 0, 0, 0, p(st.x0), stream), "sdc_step_update[x0]")
        # the reference calls nablaJ / design_fn on ``x_start.clone().detach().requires_grad_()`` inside torch.enable_grad()
        # (1D/model/diffusion.py:261-266, 2d/ddpm/diffusion_2d.py:250-252): its own closures (get_finetune_guidance,
        # InferencePipeline.design_fn) call torch.autograd.grad on their argument directly
        with torch.enable_grad():
            xc = st.x0.view(self.shape).detach().clone().requires_grad_()
            g = self.guide(xc)
        if isinstance(g, torch.Tensor):
            g = g.detach()
        gk = (g if isinstance(g, torch.Tensor) else torch.zeros_like(x) + g).to(torch.float32).contiguous()
        d_up = self.mk(guide=2, impose=(0 if (last and not self.impose_last) else 1))
        check(lib.sdc_step_update(C.byref(d_up), p(x), p(eps), p(gk), p(self.coef), p(row), p(st.draw_dev),
                                  p(self.nbuf), 0, 0, 0, 0, p(self.c0), p(self.c1), p(self.c2), p(x), 0, stream),
              "sdc_step_update[ext]")
        if self.ddim:
            check(lib.sdc_advance_table(p(st.idx_dev), p(st.t_dev), p(self.ttab), p(st.draw_dev), 1, stream), "sdc_advance_table")
        else:
            check(lib.sdc_advance(p(st.t_dev), -1, p(st.draw_dev), 1 + self.skip, stream), "sdc_advance")
        self.t_host -= 1

    def final(self):
        if not self.impose_last:
            self.step(last=True)
        if self.control_at_end and self.c1 is not None:
            # smoke DDIM: after the loop only the control channels are written back (2d/ddpm/diffusion_2d.py:400-401)
            d = self.mk(impose=2)
            check(self.gd._lib.sdc_impose(C.byref(d), self.x.data_ptr(), self._ptr(self.c0), self._ptr(self.c1),
                                          self._ptr(self.c2), self.gd._stream()), "sdc_impose[control]")

    def close(self):
        if self.graph is not None:
            torch.cuda.current_stream(self.x.device).synchronize()
            self.gd._lib.sdc_graph_destroy(self.graph)
            self.graph = None


class GaussianDiffusionBurgers(_SamplerBase):
    """Drop-in for 1D/model/diffusion.py::GaussianDiffusion (temporal=True, use_conv2d=True)."""
    MODEL = "burgers"

    def __init__(self, model, *, seq_length, timesteps=1000, sampling_timesteps=None, objective="pred_noise",
                 beta_schedule="cosine", ddim_sampling_eta=0., auto_normalize=False, guidance_u0=True,
                 conditioned_on_residual=None, residual_on_u0=False, temporal=False, use_conv2d=False,
                 is_condition_u0=False, is_condition_uT=False, is_condition_u0_zero_pred_noise=True,
                 is_condition_uT_zero_pred_noise=True, condition_idx=10, recurrence=False, recurrence_k=1,
                 normalize_beta=False, train_on_padded_locations=False, train_on_partially_observed=None,
                 set_unobserved_to_zero_during_sampling=False, is_model_w=False, eval_two_models=False,
                 expand_condition=False, prior_beta=1):
        super().__init__()
        if (objective != "pred_noise" or auto_normalize or conditioned_on_residual is not None or recurrence
                or is_model_w or eval_two_models or expand_condition or set_unobserved_to_zero_during_sampling
                or not (temporal and use_conv2d)):
            raise NotImplementedError("only the configuration built by 1D/utils/common.py:110-137 is supported")
        assert isinstance(seq_length, tuple) and len(seq_length) == 2, "should be a tuple of (Nt, Nx)"
        self._init_common(model, timesteps, sampling_timesteps, beta_schedule, ddim_sampling_eta)
        self.temporal, self.conv2d, self.traj_size = True, True, seq_length
        self.objective = objective
        self.guidance_u0 = guidance_u0
        self.is_condition_u0, self.is_condition_uT = is_condition_u0, is_condition_uT
        self.condition_idx = condition_idx
        self.train_on_padded_locations = train_on_padded_locations

    def _sample_shape(self, B):
        return (B, self.channels, *self.traj_size)

    def p_losses(self, x_start, t, noise=None, mean=True):
        """1D/model/diffusion.py:638-733 for the configuration the reference builds (pred_noise, conditions on u0 / uT, padded
        locations not trained on): eps through the differentiable HIP U-Net (safediffcon_amd.autograd), loss per sample."""
        noise = torch.randn_like(x_start) if noise is None else noise.clone()
        x = self.q_sample(x_start, t, noise)
        ci = self.condition_idx
        if self.is_condition_u0:
            x[:, 0, 0, :] = x_start[:, 0, 0, :]
        if self.is_condition_uT:
            x[:, 0, ci, :] = x_start[:, 0, ci, :]
        if not self.train_on_padded_locations:               # set_pad_condition(x): zeros
            x[:, 0, ci + 1:, :] = 0
            x[:, 1:3, ci:, :] = 0
        model_out = self.model.forward_train(x, t)
        target = noise
        if self.is_condition_u0:                             # is_condition_u0_zero_pred_noise (default True)
            target[:, 0, 0, :] = 0
        if self.is_condition_uT:
            target[:, 0, ci, :] = 0
        if not self.train_on_padded_locations:               # set_pad_condition(model_out, origin_img=target)
            keep = torch.ones_like(target, dtype=torch.bool)
            keep[:, 0, ci + 1:, :] = False
            keep[:, 1:3, ci:, :] = False
            model_out = torch.where(keep, model_out, target)
        return self._loss_tail(model_out, target, t, mean)


    @torch.no_grad()
    def sample(self, batch_size=16, clip_denoised=True, w_groundtruth=None, enable_grad=True, noise=None, **kwargs):
        """Reference signature (1D/model/diffusion.py:557-607) + ``noise`` (i -> tensor) for injected-noise parity."""
        if "guidance_u0" in kwargs:
            self.guidance_u0 = kwargs["guidance_u0"]
        ddim = self.is_ddim_sampling        # ddim_sample (:451-555); enable_grad: its last step carries an autograd graph (:524-551)
        if not (self.is_condition_u0 and self.is_condition_uT):
            raise NotImplementedError("built for is_condition_u0 = is_condition_uT = True (1D/configs/*)")
        assert kwargs.get("u_init") is not None and kwargs.get("u_final") is not None
        nablaJ, J_sched = kwargs.get("nablaJ"), kwargs.get("J_scheduler")
        if kwargs.get("proj_guidance") is not None:
            raise NotImplementedError("proj_guidance is never enabled by the reference configs")
        C_, (H, W) = self.channels, self.traj_size
        flags = dict(clip=1 if clip_denoised else 0, cond_idx=self.condition_idx,
                     pad_zero=0 if self.train_on_padded_locations else 1, has_wgt=0 if w_groundtruth is None else 1)
        guide = nablaJ
        if isinstance(guide, BurgersGuidance):
            flags["use_max"] = 0 if guide.use_max_safety else 1
        if not self.guidance_u0:
            if nablaJ is not None and not ddim:
                raise NotImplementedError("guidance on x_t (guidance_u0=False with nablaJ) is unused by the reference pipelines")
            if not ddim:
                flags["skip_draws"] = 1      # DDPM calibration branch draws twice per step (:421-423)
            guide = None
        return self._dispatch(kwargs.get("_prepare", False), batch_size, (C_, H, W, 1), noise=noise, guide=guide,
                              J_scheduler=J_sched, k_const=1.0, cond=(kwargs["u_init"], kwargs["u_final"], w_groundtruth),
                              flags=flags, impose_last=False,
                              # DDPM quirk (:431-447): with enable_grad and guidance_u0=False the t = 0 step's result is dropped;
                              # ddim_sample always takes x_start of its last step (:493-496)
                              final_update=ddim or self.guidance_u0 or not enable_grad, ddim=ddim,
                              # ddim_sample runs its last step under torch.enable_grad() whenever enable_grad is set, with or
                              # without guidance on x0 (:524-531): the returned sample carries a graph over the model
                              grad_tail=bool(enable_grad and ddim))


class GaussianDiffusionTokamak(_SamplerBase):
    """Drop-in for tokamak/model/diffusion.py::GaussianDiffusion (temporal=False)."""
    MODEL = "tokamak"

    def __init__(self, model, *, seq_length, nt=122, timesteps=1000, sampling_timesteps=None, objective="pred_noise",
                 beta_schedule="cosine", ddim_sampling_eta=0., auto_normalize=False, guidance_u0=True,
                 residual_on_u0=False, temporal=False, use_conv2d=False, is_condition_u0=True, is_condition_uT=True,
                 is_condition_u0_zero_pred_noise=True, is_condition_uT_zero_pred_noise=True,
                 train_on_padded_locations=True, expand_condition=False):
        super().__init__()
        if objective != "pred_noise" or auto_normalize or temporal or use_conv2d or expand_condition:
            raise NotImplementedError("only the configuration built by tokamak/utils/common.py:99-127 is supported")
        self._init_common(model, timesteps, sampling_timesteps, beta_schedule, ddim_sampling_eta)
        self.seq_length, self.nt, self.temporal = seq_length, nt, False
        self.objective = objective
        self.guidance_u0 = guidance_u0
        self.is_condition_u0, self.is_condition_uT = is_condition_u0, is_condition_uT
        self.train_on_padded_locations = train_on_padded_locations

    def _sample_shape(self, B):
        return (B, self.channels, self.seq_length)

    def p_losses(self, x_start, t, noise=None, mean=True):
        """tokamak/model/diffusion.py:570-637 (pred_noise; conditions on u0 / uT; train_on_padded_locations as constructed)"""
        noise = torch.randn_like(x_start) if noise is None else noise.clone()
        x = self.q_sample(x_start, t, noise)
        nt = self.nt
        if self.is_condition_u0:
            x[:, :3, 0] = x_start[:, :3, 0]
        if self.is_condition_uT:
            x[:, 0:3:2, :nt] = x_start[:, 0:3:2, :nt]          # channels 0 and 2 (a slice: a list index costs a host-to-device copy)
        if not self.train_on_padded_locations:
            x[..., :3, nt:] = x_start[..., :3, nt:]
            x[..., 3:, nt - 1:] = x_start[..., 3:, nt - 1:]
        model_out = self.model.forward_train(x, t)
        target = noise
        if self.is_condition_u0:
            target[:, :3, 0] = 0
        if self.is_condition_uT:
            target[:, 0:3:2, :nt] = 0
        if not self.train_on_padded_locations:
            keep = torch.ones_like(target, dtype=torch.bool)
            keep[..., :3, nt:] = False
            keep[..., 3:, nt - 1:] = False
            model_out = torch.where(keep, model_out, target)
        return self._loss_tail(model_out, target, t, mean)


    @torch.no_grad()
    def sample(self, batch_size=16, clip_denoised=True, w_groundtruth=None, enable_grad=True, noise=None, **kwargs):
        """Reference signature (tokamak/model/diffusion.py:498-539)."""
        if "guidance_u0" in kwargs:
            self.guidance_u0 = kwargs["guidance_u0"]
        ddim = self.is_ddim_sampling
        if not (self.is_condition_u0 and self.is_condition_uT):
            raise NotImplementedError("built for is_condition_u0 = is_condition_uT = True")
        assert kwargs.get("u_init") is not None and kwargs.get("u_final") is not None
        if w_groundtruth is not None and not ddim:
            # the reference's DDPM path executes ``img[:,1,:,:] = w_groundtruth`` on a 3-D tensor (:335-336)
            raise IndexError("too many indices for tensor of dimension 3")
        nablaJ, J_sched = kwargs.get("nablaJ"), kwargs.get("J_scheduler")
        flags = dict(clip=1 if clip_denoised else 0, cond_idx=self.nt, pad_zero=0 if self.train_on_padded_locations else 1,
                     has_wgt=0 if w_groundtruth is None else 1)
        guide, target = nablaJ, None
        if isinstance(guide, TokamakGuidance):
            target = guide.target
            assert tuple(target.shape) == (batch_size, 3, self.nt), "target must be (B, 3, nt)"
        if not self.guidance_u0:
            if nablaJ is not None and not ddim:
                raise NotImplementedError("guidance on x_t (guidance_u0=False with nablaJ) is unused by the reference pipelines")
            if not ddim:
                flags["skip_draws"] = 1
            guide = None
        return self._dispatch(kwargs.get("_prepare", False), batch_size, (self.channels, self.seq_length, 1, 1), noise=noise,
                              guide=guide, J_scheduler=J_sched, k_const=1.0,
                              cond=(kwargs["u_init"], kwargs["u_final"], w_groundtruth), flags=flags, impose_last=False,
                              target=target, final_update=ddim or self.guidance_u0 or not enable_grad, ddim=ddim,
                              grad_tail=bool(enable_grad and ddim))    # as in the 1-D class (tokamak/model/diffusion.py:455-470)


class GaussianDiffusionSmoke(_SamplerBase):
    """Drop-in for 2d/ddpm/diffusion_2d.py::GaussianDiffusion."""
    MODEL = "smoke"

    def __init__(self, model, *, image_size, frames, timesteps=1000, sampling_timesteps=None, loss_type="l1",
                 beta_schedule="sigmoid", schedule_fn_kwargs=dict(), ddim_sampling_eta=0., min_snr_loss_weight=False,
                 min_snr_gamma=5, standard_fixed_ratio=1, device=None):
        super().__init__()
        if schedule_fn_kwargs:
            raise NotImplementedError("schedule_fn_kwargs is never set by the reference")
        self._init_common(model, timesteps, sampling_timesteps, beta_schedule, ddim_sampling_eta)
        self.image_size, self.frames, self.standard_fixed_ratio, self.loss_type = image_size, frames, standard_fixed_ratio, loss_type

    def _sample_shape(self, B):
        return (B, self.frames, self.channels, self.image_size, self.image_size)

    def p_losses(self, state_start, t, noise=None, mean=True):
        """2d/ddpm/diffusion_2d.py:434-452: frame-0 density conditioned, its noise target zeroed; l1 / l2 per sample"""
        noise = torch.randn_like(state_start) if noise is None else noise.clone()
        state = self.q_sample(state_start, t, noise)
        state[:, 0, 0] = state_start[:, 0, 0]
        noise[:, 0, 0] = 0
        model_out = self.model.forward_train(state, t)
        return self._loss_tail(model_out, noise, t, mean, kind="l2" if self.loss_type == "l2" else "l1")


    @torch.no_grad()
    def sample(self, batch_size=16, design_fn=None, enable_grad=False, init=None, control=None, device=None, noise=None,
               _prepare=False):
        """Reference signature (2d/ddpm/diffusion_2d.py:406-414)."""
        ddim = self.is_ddim_sampling                  # ddim_sample, 2d/ddpm/diffusion_2d.py:324-404
        assert init is not None and batch_size == init.shape[0]
        flags = dict(clip=1, has_wgt=0 if control is None else 1)
        S = self.image_size
        return self._dispatch(_prepare, batch_size, (self.frames, self.channels, S, S), noise=noise, guide=design_fn,
                              J_scheduler=None, k_const=float(self.standard_fixed_ratio), cond=(init, control, None),
                              flags=flags, impose_last=True, ddim=ddim, control_at_end=ddim,
                              grad_tail=bool(enable_grad and ddim))   # (the reference's DDPM p_sample is @torch.no_grad: no graph there)


# reference-compatible alias: each reference tree calls its class ``GaussianDiffusion``
GaussianDiffusion = GaussianDiffusionBurgers
