"""Oracle: DDPM reverse loops, guidance gradients and conformal reductions.

Restates (file:line relative to /root/reference)
  burgers  p_sample_loop 1D/model/diffusion.py:368-449, p_sample :299-306,
           model_predictions :226-286, q_posterior :217-224,
           set_condition :336-358, set_pad_condition :360-366
  tokamak  p_sample_loop tokamak/model/diffusion.py:310-372, set_condition :295-308
  smoke    p_sample_loop 2d/ddpm/diffusion_2d.py:288-322, p_sample :275-285,
           model_predictions :242-261
  guidance 1D/utils/guidance.py:58-85, tokamak/utils/guidance.py:32-73,
           2d/inference_2d.py:173-195
  conformal 1D/inference/conformal.py:68-117, 1D/inference/guidance.py:39-66,
           tokamak/inference/conformal.py:103-145, tokamak/utils/guidance.py:98-148,
           2d/inference_2d.py:83-165

RNG protocol: every sampler takes ``noise(i)`` -> tensor; i = 0 is the x_T draw
and i >= 1 are the per-step draws in the order the reference makes them (after
the U-Net forward, none at t == 0; the burgers/tokamak calibration branch makes
two draws per step and discards the first, 1D/model/diffusion.py:421-423).

TEST INFRASTRUCTURE -- see oracle/__init__.py.
"""
import math

import numpy as np
import torch

BURGERS_SCALER = 10.0                                                    # 1D/utils/common.py:17
TOKAMAK_SCALER = torch.tensor([2, 7, 2, 1, 2, 2, 2, 2, 1, 1, 2, 3.0]).reshape(12, 1)   # tokamak/utils/common.py:16
SMOKE_RESCALER = torch.tensor([2, 19, 20, 17, 20, 1, 1.0]).reshape(1, 1, 7, 1, 1)      # 2d/ddpm/data_2d.py:38


class _Draws:
    def __init__(self, noise):
        self.noise, self.i = noise, 0

    def __call__(self):
        z = self.noise(self.i)
        self.i += 1
        return z


def _posterior_step(tabs, x, t, eps, g, k, clip, z):
    """eps -> x0 -> eps' = eps + k*g -> x0' -> clamp -> posterior mean -> + sigma z."""
    a, b = tabs["sqrt_recip_alphas_cumprod"][t], tabs["sqrt_recipm1_alphas_cumprod"][t]
    if g is not None:
        eps = eps + g * k
    x0 = a * x - b * eps
    if clip:
        x0 = x0.clamp(-1.0, 1.0)
    mean = tabs["posterior_mean_coef1"][t] * x0 + tabs["posterior_mean_coef2"][t] * x
    if z is None:
        return mean, x0
    return mean + (0.5 * tabs["posterior_log_variance_clipped"][t]).exp() * z, x0


def _x0_from_eps(tabs, x, t, eps):
    return tabs["sqrt_recip_alphas_cumprod"][t] * x - tabs["sqrt_recipm1_alphas_cumprod"][t] * eps


# ----------------------------------------------------------------------------
# guidance (autograd of the restated J, like the reference does it)
# ----------------------------------------------------------------------------

def burgers_J(state, Q, w_score, u_bound, use_max_safety=True):
    """calculate_guidance, 1D/utils/guidance.py:58-77 (mean when use_max_safety!)."""
    s = (state * BURGERS_SCALER)[:, 2, :11, :]
    s = s.mean(dim=(-1, -2)) if use_max_safety else s.amax(dim=(-1, -2))
    return torch.maximum(s + Q - u_bound ** 2, torch.zeros_like(s)) * w_score


def burgers_guidance(Q, w_score, u_bound, use_max_safety=True):
    def nablaJ(x):
        with torch.enable_grad():
            x = x.detach().requires_grad_()
            return torch.autograd.grad(burgers_J(x, Q, w_score, u_bound, use_max_safety).sum(), x)[0]
    return nablaJ


def tokamak_J(x, target, nt, Q, thr, w_obj, w_safe):
    """GradientGuidance.calculate_loss, tokamak/utils/guidance.py:32-56."""
    st = (x * TOKAMAK_SCALER.to(x.device))[:, :3, :nt]
    obj = (st[:, 0] - target[:, 0]).square().mean(-1) + (st[:, 2] - target[:, 2]).square().mean(-1)
    s = st[:, 1].amin(dim=-1)                                            # utils/metrics.py:144-151
    safe = torch.maximum(thr - s + Q, torch.zeros_like(s))
    return w_obj * obj + w_safe * safe


def tokamak_guidance(target, nt, Q, thr, w_obj, w_safe, scaler):
    def nablaJ(x):
        with torch.enable_grad():
            x = x.detach().requires_grad_()
            loss = tokamak_J(x, target, nt, Q, thr, w_obj, w_safe) * scaler
            return torch.autograd.grad(loss, x, grad_outputs=torch.ones_like(loss))[0]
    return nablaJ


def smoke_J(x, Q, w_safe, safe_bound):
    """InferencePipeline.guidance, 2d/inference_2d.py:173-186."""
    st = x * SMOKE_RESCALER.to(x.device)
    succ = st[:, :, 5].mean((-1, -2, -3))
    safe = torch.maximum(st[:, -1, 6].mean((-1, -2)) + Q - safe_bound, torch.zeros_like(st[:, -1, 6, 0, 0]))
    return -(1 - w_safe) * succ + w_safe * safe


def smoke_guidance(Q, w_safe, safe_bound):
    def design_fn(x):
        with torch.enable_grad():
            x = x.detach().requires_grad_()
            return torch.autograd.grad(smoke_J(x, Q, w_safe, safe_bound).sum(), x)[0]
    return design_fn


# ----------------------------------------------------------------------------
# reverse loops
# ----------------------------------------------------------------------------

def _lucid_loop(eps_fn, tabs, shape, noise, impose, *, nablaJ, J_scheduler, guidance_u0,
                clip_denoised, enable_grad, T):
    """Common body of the burgers / tokamak p_sample_loop."""
    draw = _Draws(noise)
    img = draw().clone()
    for t in reversed(range(T)):
        impose(img)
        eps = eps_fn(img, torch.full((shape[0],), t, dtype=torch.long))
        k = J_scheduler(t) if J_scheduler is not None else 1.0
        if guidance_u0:
            g = nablaJ(_x0_from_eps(tabs, img, t, eps)) if nablaJ is not None else None
            img, _ = _posterior_step(tabs, img, t, eps, g, k, clip_denoised, draw() if t > 0 else None)
        else:
            # calibration branch: p_sample twice; first result only feeds nabla_J(img_curr)
            cur, _ = _posterior_step(tabs, img, t, eps, None, k, clip_denoised, draw() if t > 0 else None)
            eps2 = eps + (nablaJ(cur) * k if nablaJ is not None else 0)
            _ = eps_fn(img, torch.full((shape[0],), t, dtype=torch.long))   # discarded forward (:423)
            nxt, _ = _posterior_step(tabs, img, t, eps2, None, k, clip_denoised, draw() if t > 0 else None)
            if t != 0 or not enable_grad:
                img = nxt
            # t == 0 with enable_grad and guidance_u0 False: reference keeps img (:445-447)
    return img


def sample_burgers(eps_fn, tabs, batch, noise, *, u_init, u_final, nablaJ=None, J_scheduler=None,
                   guidance_u0=True, w_groundtruth=None, clip_denoised=True, enable_grad=True,
                   condition_idx=10, train_on_padded_locations=False, shape=(3, 16, 128), T=None):
    T = T or tabs["betas"].shape[0]

    def impose(img):
        img[:, 0, 0, :] = u_init
        img[:, 0, condition_idx, :] = u_final
        if w_groundtruth is not None:
            img[:, 1, :, :] = w_groundtruth
        if not train_on_padded_locations:
            img[:, 0, condition_idx + 1:, :] = 0
            img[:, 1, condition_idx:, :] = 0
            img[:, 2, condition_idx:, :] = 0

    return _lucid_loop(eps_fn, tabs, (batch, *shape), noise, impose, nablaJ=nablaJ, J_scheduler=J_scheduler,
                       guidance_u0=guidance_u0, clip_denoised=clip_denoised, enable_grad=enable_grad, T=T)


def sample_tokamak(eps_fn, tabs, batch, noise, *, u_init, u_final, nablaJ=None, J_scheduler=None,
                   guidance_u0=True, w_groundtruth=None, clip_denoised=True, enable_grad=True,
                   nt=122, train_on_padded_locations=True, shape=(12, 128), T=None):
    T = T or tabs["betas"].shape[0]
    if w_groundtruth is not None:
        # reference bug: ``img[:,1,:,:] = w_groundtruth`` on a 3-D tensor
        raise IndexError("too many indices for tensor of dimension 3")

    def impose(img):
        img[:, :3, 0] = u_init
        img[:, [0, 2], :nt] = u_final
        if not train_on_padded_locations:
            img[:, :3, nt:] = 0
            img[:, 3:, nt - 1:] = 0

    return _lucid_loop(eps_fn, tabs, (batch, *shape), noise, impose, nablaJ=nablaJ, J_scheduler=J_scheduler,
                       guidance_u0=guidance_u0, clip_denoised=clip_denoised, enable_grad=enable_grad, T=T)


def sample_smoke(eps_fn, tabs, batch, noise, *, init, control=None, design_fn=None, ratio=1.0,
                 shape=(32, 7, 64, 64), T=None):
    T = T or tabs["betas"].shape[0]
    draw = _Draws(noise)
    x = draw().clone()

    def impose(x):
        x[:, 0, 0] = init
        if control is not None:
            x[:, :, 3:5] = control

    impose(x)
    for t in reversed(range(T)):
        eps = eps_fn(x, torch.full((batch,), t, dtype=torch.long))
        g = design_fn(_x0_from_eps(tabs, x, t, eps)) if design_fn is not None else None
        x, _ = _posterior_step(tabs, x, t, eps, g, ratio, True, draw() if t > 0 else None)
        impose(x)
    return x


# ----------------------------------------------------------------------------
# conformal
# ----------------------------------------------------------------------------

def normalize_weights(w, smoke=False):
    w = w.clone()
    inf = torch.isinf(w)
    if inf.any():
        w[inf] = w[~inf].max()
    if w.sum() == 0:
        out = torch.ones_like(w)
    else:
        out = w.shape[0] * w / w.sum()
    if smoke:                                                            # 2d/inference_2d.py:110
        bad = torch.isinf(out)
        out[bad] = w.shape[0] / bad.sum()
    return out


def burgers_weight(state, Q, w_score, u_bound, use_max_safety=True):
    return torch.exp(-burgers_J(state, Q, w_score, u_bound, use_max_safety))


def burgers_score(pred, state, use_max_safety=True):
    f = (lambda s: s.mean(dim=(-1, -2))) if use_max_safety else (lambda s: s.amax(dim=(-1, -2)))
    return (f((pred * BURGERS_SCALER)[:, 2, :11, :]) - f((state * BURGERS_SCALER)[:, 2, :11, :])).abs()


def tokamak_weight(state, target, nt, Q, thr, w_obj, w_safe, scaler):
    return torch.exp(-tokamak_J(state, target, nt, Q, thr, w_obj, w_safe) * scaler)


def tokamak_score(pred, state, nt):
    f = lambda s: (s * TOKAMAK_SCALER)[:, 1, :nt].amin(dim=-1)
    return (f(pred) - f(state)).abs()


def smoke_weight(state, Q, w_safe, safe_bound, ratio):
    return torch.exp(-ratio * smoke_J(state, Q, w_safe, safe_bound))


def smoke_score(pred, state):
    p, s = pred * SMOKE_RESCALER.to(pred.device), state * SMOKE_RESCALER.to(state.device)
    return (p[:, -1, -1].mean((-1, -2)) - s[:, -1, -1, 0, 0]).abs()


def quantile_lucid(scores, alpha):
    """1D/inference/conformal.py:95-117 == tokamak/inference/conformal.py:121-145."""
    n = scores.shape[0]
    _, idx = torch.sort(scores)
    rank = min(int(np.ceil(alpha * (n + 1))), n) - 1
    return scores[idx[rank]]


def quantile_smoke(scores, alpha):
    """2d/inference_2d.py:150-165."""
    n = scores.shape[0]
    _, idx = torch.sort(scores)
    q = int(min(np.ceil((n + 1) * (1 - alpha)), n - 1))
    return scores[idx[q - 1]]


# ----------------------------------------------------------------------------
# DDIM: what the reference's shipped scripts actually run
# ----------------------------------------------------------------------------

def ddim_pairs(T, S):
    """[(time, time_next)] of ddim_sample: 1D/model/diffusion.py:460-462 (== tokamak, 2d)."""
    times = torch.linspace(-1, T - 1, steps=S + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def _ddim_loop(eps_fn, tabs, shape, noise, impose, finish, *, S, eta, guide, k_of_t):
    """ddim_sample: 1D/model/diffusion.py:451-555, tokamak/...:374-496, 2d/ddpm/diffusion_2d.py:324-404.
    model_predictions(clip_x_start=True, rederive_pred_noise=True): x0 clipped, guidance evaluated on the clipped x0,
    eps re-derived from the clipped guided x0."""
    T = tabs["betas"].shape[0]
    ac = tabs["alphas_cumprod"]
    draw = _Draws(noise)
    img = draw().clone()
    impose(img)
    for time, time_next in ddim_pairs(T, S):
        a, b = tabs["sqrt_recip_alphas_cumprod"][time], tabs["sqrt_recipm1_alphas_cumprod"][time]
        eps = eps_fn(img, torch.full((shape[0],), time, dtype=torch.long))
        x0 = (a * img - b * eps).clamp(-1.0, 1.0)
        if guide is not None:
            eps = eps + guide(x0) * k_of_t(time)
        x0 = (a * img - b * eps).clamp(-1.0, 1.0)
        eps = (a * img - x0) / b
        if time_next < 0:
            img = x0
            continue
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        img = x0 * alpha_next.sqrt() + c * eps + sigma * draw()
        impose(img)
    finish(img)
    return img


def ddim_burgers(eps_fn, tabs, batch, noise, *, S, eta, u_init, u_final, nablaJ=None, J_scheduler=None, guidance_u0=True,
                 w_groundtruth=None, condition_idx=10, train_on_padded_locations=False, shape=(3, 16, 128)):
    def impose(img):
        img[:, 0, 0, :] = u_init
        img[:, 0, condition_idx, :] = u_final
        if w_groundtruth is not None:
            img[:, 1, :, :] = w_groundtruth
        if not train_on_padded_locations:
            img[:, 0, condition_idx + 1:, :] = 0
            img[:, 1, condition_idx:, :] = 0
            img[:, 2, condition_idx:, :] = 0
    k = (lambda t: J_scheduler(t)) if J_scheduler is not None else (lambda t: 1.0)
    return _ddim_loop(eps_fn, tabs, (batch, *shape), noise, impose, lambda img: None, S=S, eta=eta,
                      guide=nablaJ if guidance_u0 else None, k_of_t=k)


def ddim_tokamak(eps_fn, tabs, batch, noise, *, S, eta, u_init, u_final, nablaJ=None, J_scheduler=None, guidance_u0=True,
                 w_groundtruth=None, nt=122, train_on_padded_locations=True, shape=(12, 128)):
    def impose(img):
        img[:, :3, 0] = u_init
        img[:, [0, 2], :nt] = u_final
        if not train_on_padded_locations:
            img[:, :3, nt:] = 0
            img[:, 3:, nt - 1:] = 0
        if w_groundtruth is not None:
            img[:, 3:, :] = w_groundtruth                                 # the DDIM path indexes correctly (:411,:453)
    k = (lambda t: J_scheduler(t)) if J_scheduler is not None else (lambda t: 1.0)
    return _ddim_loop(eps_fn, tabs, (batch, *shape), noise, impose, lambda img: None, S=S, eta=eta,
                      guide=nablaJ if guidance_u0 else None, k_of_t=k)


def ddim_smoke(eps_fn, tabs, batch, noise, *, S, eta, init, control=None, design_fn=None, ratio=1.0, shape=(32, 7, 64, 64)):
    def impose(x):
        x[:, 0, 0] = init
        if control is not None:
            x[:, :, 3:5] = control

    def finish(x):                                                        # 2d/ddpm/diffusion_2d.py:400-401
        if control is not None:
            x[:, :, 3:5] = control
    return _ddim_loop(eps_fn, tabs, (batch, *shape), noise, impose, finish, S=S, eta=eta, guide=design_fn,
                      k_of_t=lambda t: ratio)
