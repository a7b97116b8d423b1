"""Conformal calibration reductions (the one cross-GPU exchange of the path).

Per calibration sample the HIP kernel ``sdc_conformal_score`` produces
  score_i  = | f(pred_i) - f(truth_i) |
  weight_i = exp(-J(truth_i))
(1D/inference/conformal.py:68-85 + inference/guidance.py:9-46, tokamak/inference/conformal.py:79-109,
2d/inference_2d.py:83-92,139-144).  Each rank does that for its shard of the calibration set; one
all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) of the two length-n/world vectors follows,
and every rank then runs the identical tiny epilogue: inf / zero-sum fix-ups, n*w/sum(w), w*score,
sort, alpha-rank select (1D/inference/conformal.py:87-117, 2d/inference_2d.py:94-111,150-165) -> the
same scalar Q everywhere.  n <= 1000, so the epilogue is host-side bookkeeping on <= 8 KB.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import SdcStepDesc, check

_MODEL = {"burgers": 0, "tokamak": 1, "smoke": 2}


def scores_and_weights(model, pred, truth, gpar, *, target=None, use_max=False, nt=122):
    """(score[B], weight[B]) on the device of ``pred`` through sdc_conformal_score.
    gpar: burgers [w_score, u_bound^2, Q, 10] | tokamak [w_obj, w_safe, scaler, thr, Q] | smoke [w_safe, safe_bound, Q, ratio]."""
    if not pred.is_cuda:
        raise RuntimeError("safediffcon_amd.conformal runs on MI355X only (no CPU fallback)")
    lib = _lib.get_lib()
    dev = pred.device
    pred = pred.detach().to(torch.float32).contiguous()
    truth = truth.detach().to(dev, torch.float32).contiguous()
    assert pred.shape == truth.shape
    B = pred.shape[0]
    d = SdcStepDesc()
    d.model, d.B = _MODEL[model], B
    dims = tuple(pred.shape[1:]) + (1,) * (5 - pred.dim())
    d.d0, d.d1, d.d2, d.d3 = dims
    d.use_max = 1 if use_max else 0
    d.cond_idx = nt if model == "tokamak" else 10
    d.clip = 1
    g = torch.tensor(list(gpar) + [0.0] * (8 - len(gpar)), dtype=torch.float32, device=dev)
    tg = None if target is None else target.detach().to(dev, torch.float32).contiguous()
    score = torch.empty(B, dtype=torch.float32, device=dev)
    weight = torch.empty(B, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    check(lib.sdc_conformal_score(C.byref(d), pred.data_ptr(), truth.data_ptr(), 0 if tg is None else tg.data_ptr(),
                                  g.data_ptr(), score.data_ptr(), weight.data_ptr(), stream), "sdc_conformal_score")
    return score, weight


def all_gather_1d(t, group=None):
    """Concatenate equal-length 1-D shards over the process group (no-op when not distributed)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t.contiguous(), group=group)
    return torch.cat(out)


def normalize_weights(weights, smoke=False):
    """inf -> max finite; sum 0 -> ones; else n*w/sum(w).  1D/inference/guidance.py:48-66 (== tokamak/utils/guidance.py:130-148);
    smoke adds the trailing inf fix-up of 2d/inference_2d.py:110."""
    w = weights.clone()
    inf = torch.isinf(w)
    if inf.any():
        w[inf] = w[~inf].max()
    if w.sum() == 0:
        out = torch.ones_like(w)
    else:
        out = w.shape[0] * w / w.sum()
    if smoke:
        bad = torch.isinf(out)
        out[bad] = w.shape[0] / bad.sum()
    return out


def quantile_rank(n, alpha, smoke=False):
    """index into the ascending sort: 1D/inference/conformal.py:112 ; 2d/inference_2d.py:159-160."""
    if smoke:
        return int(min(math.ceil((n + 1) * (1 - alpha)), n - 1)) - 1
    return min(int(math.ceil(alpha * (n + 1))), n) - 1


def calculate_quantile(scores, alpha, smoke=False):
    n = scores.shape[0]
    _, idx = torch.sort(scores)
    return scores[idx[quantile_rank(n, alpha, smoke)]]


def weighted_quantile(score_shard, weight_shard, alpha, *, smoke=False, group=None):
    """shards -> all-gather -> normalise -> weighted scores -> Q (identical on every rank)."""
    s = all_gather_1d(score_shard, group)
    w = all_gather_1d(weight_shard, group)
    nw = normalize_weights(w, smoke)
    return calculate_quantile(nw * s, alpha, smoke), nw


class ConformalCalculator:
    """Drop-in for 1D/inference/conformal.py::ConformalCalculator (and the tokamak twin): same
    get_conformal_scores / calculate_quantile contract, scores and weights from the HIP kernel."""

    def __init__(self, model, config, kind="burgers"):
        self.model, self.config, self.kind = model, config, kind
        self.device = getattr(config, "device", "cuda")

    def _gpar(self, Q):
        c = self.config
        if self.kind == "burgers":
            return [c.guidance_weights["w_score"], c.u_bound ** 2, float(Q), 10.0]
        return [c.guidance_weights["w_obj"], c.guidance_weights["w_safe"], c.guidance_scaler, c.safety_threshold, float(Q)]

    def get_conformal_scores(self, dataloader, Q, cal_targets=None, group=None):
        c = self.config
        scores, weights, states = [], [], []
        for _ in range(c.num_cal_batch):
            item = next(dataloader)
            state, idx = (item if isinstance(item, (tuple, list)) else (item, None))
            states.append(state)
            state = state.to(self.device)
            if self.kind == "burgers":
                out = self.model.sample(batch_size=state.shape[0], clip_denoised=True, guidance_u0=False,
                                        u_init=state[:, 0, 0, :], u_final=state[:, 0, c.nt - 1, :],
                                        w_groundtruth=state[:, 1, :, :], nablaJ=None, J_scheduler=None, w_scheduler=None,
                                        enable_grad=False)
                s, w = scores_and_weights("burgers", out, state, self._gpar(Q), use_max=not c.use_max_safety)
                if getattr(c, "InfFT_Q", None) is not None:
                    w = w * scores_and_weights("burgers", out, state, self._gpar(c.InfFT_Q), use_max=not c.use_max_safety)[1]
            else:
                # tokamak/inference/conformal.py:62-74: the calibration samples are conditioned on the ground-truth actions
                # (DDIM samplers; the DDPM path raises the reference's IndexError)
                out = self.model.sample(batch_size=state.shape[0], clip_denoised=True, guidance_u0=False,
                                        u_init=state[:, :3, 0], u_final=state[:, [0, 2], :c.nt_total],
                                        w_groundtruth=state[:, 3:, :], nablaJ=None, J_scheduler=None, w_scheduler=None,
                                        enable_grad=False)
                tg = cal_targets[idx].to(self.device)
                s, w = scores_and_weights("tokamak", out, state, self._gpar(Q), target=tg, nt=c.nt_total)
                if getattr(c, "finetune_set", None) == "train" and getattr(c, "use_guidance", False):
                    w = w * w                        # the same calculate_weight factor a second time (:86-93)
                if getattr(c, "finetune_set", None) == "test" and not getattr(c, "wo_post_train", True):
                    fg = c.finetune_guidance_weights   # :94-102
                    g2 = [fg["w_obj"], fg["w_safe"], c.finetune_guidance_scaler, c.safety_threshold, float(c.finetune_quantile)]
                    w = w * scores_and_weights("tokamak", out, state, g2, target=tg, nt=c.nt_total)[1]
            scores.append(s)
            weights.append(w)
        s = all_gather_1d(torch.cat(scores), group)
        w = all_gather_1d(torch.cat(weights), group)
        nw = normalize_weights(w)
        return nw * s, nw, torch.cat(states)

    def calculate_quantile(self, scores, weights, states, alpha):
        return calculate_quantile(scores, alpha)


class SmokeConformal:
    """2D smoke: drop-in for the conformal / weighting methods of InferencePipeline
    (2d/inference_2d.py:83-165: get_weight, normalize_weights, get_weighted_score_set, get_quantile, guidance)."""

    def __init__(self, model, args_general, RESCALER=None):
        self.model, self.args_general = model, args_general
        self.device = getattr(args_general, "device", "cuda")
        self.Q = 0.0

    def _gpar(self, Q, ratio):
        a = self.args_general
        return [a.w_safe, a.safe_bound, float(Q), float(ratio)]

    def get_weight(self, state, mode="train"):
        a = self.args_general
        if mode == "train":
            g = self._gpar(self.Q, a.standard_fixed_ratio)
        else:   # the reference's guidance() adds self.Q even when a Q argument is passed (2d/inference_2d.py:183)
            g = self._gpar(self.Q, a.finetune_standard_fixed_ratio)
        return scores_and_weights("smoke", state, state, g)[1]

    def normalize_weights(self, weights):
        return normalize_weights(weights, smoke=True)

    def get_weighted_score_set(self, cal_dataloader, group=None):
        a = self.args_general
        scores, weights, states = [], [], []
        for _ in range(a.N_cal_batch):
            state, _sim_id = next(cal_dataloader)
            states.append(state)
            state = state.to(self.device)
            out = self.model.sample(batch_size=state.shape[0], design_fn=None, init=state[:, 0, 0], control=state[:, :, 3:5])
            s, w = scores_and_weights("smoke", out, state, self._gpar(self.Q, a.standard_fixed_ratio))
            if getattr(a, "finetune_set", "train") != "train":
                w = w * self.get_weight(state, mode="test")
            scores.append(s)
            weights.append(w)
        s = all_gather_1d(torch.cat(scores), group)
        w = all_gather_1d(torch.cat(weights), group)
        nw = normalize_weights(w, smoke=True)
        return nw * s, nw, torch.cat(states)

    def get_quantile(self, conformal_score_set, normalized_weights, ori_states, alpha):
        return calculate_quantile(conformal_score_set, alpha, smoke=True)

    def conformal_prediction(self, cal_dataloader):
        with torch.no_grad():
            s, nw, st = self.get_weighted_score_set(cal_dataloader)
        self.Q = self.get_quantile(s, nw, st, self.args_general.alpha)
        return self.Q
