"""Tokamak score check on the GPU: the KSTAR surrogate rollout and the evaluation metrics around it.

Mirrors, by name and argument meaning,
  KSTARSolver(random_seed).simulate(actions)            tokamak/kstar_solver.py:120-148, 389-428
  control_trajectories(diffused, nt_total, seed)        tokamak/utils/metrics.py:60-85
  calculate_safety_score / calculate_safety_metrics /
  calculate_reported_safe_metric / evaluate_samples     tokamak/utils/metrics.py:11-58, 87-151
The reference builds the simulator from TensorFlow/Keras networks (tokamak/common/model_structure.py:69-152) and steps one
sample at a time, 122 ``model.predict`` calls each; here the networks' weights are read straight from the Keras HDF5 files
(``h5lite``; no TensorFlow, no h5py) and ``sdc_kstar_rollout`` carries the whole batch through the 122 rows in one launch.
There is no CPU path: without ``libsdc_hip.so`` and a GPU every entry point raises.

Keras semantics restated here (load order and layer maths) are spelled out in ``load_weights`` and ``csrc/sdc_kstar.hip``;
the reference cannot be run in this image (TensorFlow absent), so this row's parity is *unpinned* -- see DESIGN.md section 9.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

from . import _lib, h5lite
from ._lib import check

# ---- constants of tokamak/kstar_solver.py:29-112 and tokamak/common/model_structure.py:100-143
SEQ_LEN, N_IN, UNITS = 10, 18, 100
YEAR_IN = 2021
LOW_ACTION = [0.3, 0.0, 0.0, 0.0, 1.6, 0.15, 0.5, 1.265, 2.14]
HIGH_ACTION = [0.8, 1.75, 1.75, 1.5, 1.95, 0.5, 0.85, 1.36, 2.3]
INPUT_INIT = [0.5, 1.8, 0.33, 1.5, 1.5, 0.5, 0.0, 0.0, 0.0, 0.0, 1.32, 2.22, 1.7, 0.3, 0.75]
LSTM_YMEAN, LSTM_YSTD = [1.4361666, 5.275876, 1.534538, 1.1268075], [0.7294007, 1.5010427, 0.6472052, 0.2331879]
NN_YMEAN, NN_YSTD = [1.22379703, 5.2361062, 1.64438005, 1.12040048], [0.72255576, 1.5622809, 0.96563557, 0.23868018]
BPW_YMEAN, BPW_YSTD = [1.02158800e+00, 1.87408512e+05], [6.43390272e-01, 1.22543529e+05]
N_STEPS = 1 + 12 * 10                                   # control + predict_0d calls after the steady-state row
OUTPUTS = ["βn", "βp", "h89", "h98", "q95", "q0", "li", "wmhd"]      # output_params2, kstar_solver.py:91
_BN_KEYS = ("gamma", "beta", "moving_mean", "moving_variance")


def _scale():
    """kstar_solver.py:35,107-113 scale by 10 ** np.log10(1000), evaluated the same way here (1000.0 with this numpy)"""
    return float(10 ** np.log10(1000))


# ------------------------------------------------------------------------------------------------ weights
def _layer_arrays(group, layer):
    """{short weight name: array} of one layer of a Keras ``model_weights`` group, in the stored order"""
    g = group[layer]
    names = g.attrs.get("weight_names")
    out = {}
    for n in ([] if names is None else list(names)):
        n = n.decode() if isinstance(n, bytes) else str(n)
        out[n.rsplit("/", 1)[-1].split(":")[0]] = np.ascontiguousarray(g[n][...], dtype=np.float32)
    return out


def _stored_layers(f):
    mw = f["model_weights"]
    names = [n.decode() if isinstance(n, bytes) else str(n) for n in mw.attrs["layer_names"]]
    return [(n, _layer_arrays(mw, n)) for n in names]


def _bn(a, eps=1e-3):
    return {"gamma": a["gamma"], "beta": a["beta"], "mean": a["moving_mean"], "var": a["moving_variance"], "eps": np.float32(eps)}


def _load_lstm_net(path):
    """``load_custom_model((10, 18), [100, 100], [50, 4], path)`` (model_structure.py:69-81): a fresh Sequential
    [BN, LSTM(100, seq), BN, LSTM(100), BN, Dense(50, sigmoid), BN, Dense(4)] whose ``load_weights`` pairs, in order, its layers
    that own weights with the stored layers that own weights -- the stored config (GaussianNoise, Lambda, TimeDistributed
    wrappers, hard_sigmoid) is never instantiated"""
    stored = [(n, a) for n, a in _stored_layers(h5lite.File(path)) if a]
    want = ["bn", "lstm", "bn", "lstm", "bn", "dense", "bn", "dense"]
    if len(stored) != len(want):
        raise ValueError(f"{path}: {len(stored)} stored layers own weights, the model has {len(want)}")
    out = {}
    counts = {"bn": 0, "lstm": 0, "dense": 0}
    for kind, (name, a) in zip(want, stored):
        keys = set(a)
        if kind == "bn" and keys == set(_BN_KEYS):
            out[f"bn{counts['bn']}"] = _bn(a)
        elif kind == "lstm" and keys == {"kernel", "recurrent_kernel", "bias"}:
            out[f"lstm{counts['lstm']}"] = a
        elif kind == "dense" and keys == {"kernel", "bias"}:
            out[f"dense{counts['dense']}"] = a
        else:
            raise ValueError(f"{path}: stored layer {name!r} holds {sorted(keys)}, the model expects a {kind} layer there")
        counts[kind] += 1
    shapes = (out["lstm0"]["kernel"].shape, out["lstm0"]["recurrent_kernel"].shape, out["lstm1"]["kernel"].shape,
              out["dense0"]["kernel"].shape, out["dense1"]["kernel"].shape)
    if shapes != ((N_IN, 4 * UNITS), (UNITS, 4 * UNITS), (UNITS, 4 * UNITS), (UNITS, 50), (50, 4)):
        raise ValueError(f"{path}: unexpected weight shapes {shapes}")
    return out


def _load_dense_net(path):
    """``models.load_model(path, compile=False)`` for the stored Sequentials of kstar_nn / bpw_nn (model_structure.py:125,143):
    BatchNormalization / Dense / Dropout layers from the file's own ``model_config``"""
    f = h5lite.File(path)
    cfg = json.loads(f.attrs["model_config"])
    if cfg.get("class_name") != "Sequential":
        raise ValueError(f"{path}: only Sequential models are supported, found {cfg.get('class_name')}")
    layers_cfg = cfg["config"]["layers"] if isinstance(cfg["config"], dict) else cfg["config"]
    arrays = dict(_stored_layers(f))
    layers = []
    for lc in layers_cfg:
        kind, c = lc["class_name"], lc["config"]
        a = arrays.get(c["name"], {})
        if kind == "BatchNormalization":
            if not (c.get("center", True) and c.get("scale", True)):
                raise ValueError(f"{path}: BatchNormalization without center/scale is not supported")
            layers.append(("bn", _bn(a, c.get("epsilon", 1e-3))))
        elif kind == "Dense":
            if c.get("activation") not in ("sigmoid", "linear") or not c.get("use_bias", True):
                raise ValueError(f"{path}: Dense activation {c.get('activation')!r} / use_bias={c.get('use_bias')} is not supported")
            layers.append(("dense", {"kernel": a["kernel"], "bias": a["bias"], "activation": c["activation"]}))
        elif kind in ("Dropout", "GaussianNoise", "InputLayer"):
            continue                                        # identities at inference
        else:
            raise ValueError(f"{path}: layer class {kind} is not supported")
    return {"layers": layers}


def load_weights(weights_dir, n_models=1):
    """the networks ``KSTARSolver.__init__`` loads (kstar_solver.py:54-57,136-143), first ``n_models`` of each ensemble the
    rollout averages (``n_model_box`` = 1 in the reference): {"lstm": [...], "nn": [one], "bpw": [...]}"""
    lstm_dir = os.path.join(weights_dir, "lstm", "v220505")
    return {"lstm": [_load_lstm_net(os.path.join(lstm_dir, f"best_model{i}")) for i in range(n_models)],
            "nn": [_load_dense_net(os.path.join(weights_dir, "nn", "best_model0"))],
            "bpw": [_load_dense_net(os.path.join(weights_dir, "bpw", f"best_model{i}")) for i in range(n_models)]}


def flatten_weights(w):
    """nested weights -> {"lstm/0/bn0/gamma": array, ...} (what tests/golden/kstar_weights.npz stores)"""
    flat = {}
    for i, m in enumerate(w["lstm"]):
        for lname, d in m.items():
            for k, v in d.items():
                flat[f"lstm/{i}/{lname}/{k}"] = np.asarray(v)
    for fam in ("nn", "bpw"):
        for i, m in enumerate(w[fam]):
            for j, (kind, d) in enumerate(m["layers"]):
                for k, v in d.items():
                    flat[f"{fam}/{i}/{j:02d}_{kind}/{k}"] = np.asarray(v)
    return flat


def unflatten_weights(flat):
    w = {"lstm": {}, "nn": {}, "bpw": {}}
    for key in sorted(flat):
        fam, i, lname, k = key.split("/")
        v = flat[key]
        v = str(v) if k == "activation" else np.asarray(v)
        w[fam].setdefault(int(i), {}).setdefault(lname, {})[k] = v
    out = {"lstm": [w["lstm"][i] for i in sorted(w["lstm"])]}
    for fam in ("nn", "bpw"):
        out[fam] = [{"layers": [(ln.split("_", 1)[1], d) for ln, d in sorted(w[fam][i].items())]} for i in sorted(w[fam])]
    return out


# ------------------------------------------------------------------------------------------------ device model
def _fold_bn(bn):
    """tf.nn.batch_normalization at inference: x * inv + (beta - mean * inv), inv = gamma * rsqrt(var + eps), in float32"""
    eps = np.float32(bn.get("eps", 1e-3))
    inv = (bn["gamma"].astype(np.float32) / np.sqrt(bn["var"].astype(np.float32) + eps)).astype(np.float32)
    return inv, (bn["beta"].astype(np.float32) - bn["mean"].astype(np.float32) * inv).astype(np.float32)


def _mlp_blocks(layers):
    """[(kind, params)] -> [(inv, off, kernel, bias, act)]: every Dense with the BatchNormalization in front of it"""
    blocks, pending = [], None
    for kind, p in layers:
        if kind == "bn":
            if pending is not None:
                raise ValueError("two BatchNormalization layers in a row are not supported")
            pending = _fold_bn(p)
        else:
            n_in = p["kernel"].shape[0]
            inv, off = pending if pending is not None else (np.ones(n_in, np.float32), np.zeros(n_in, np.float32))
            blocks.append((inv, off, p["kernel"].astype(np.float32), p["bias"].astype(np.float32), 1 if p["activation"] == "sigmoid" else 0))
            pending = None
    if pending is not None:
        raise ValueError("a trailing BatchNormalization is not supported")
    return blocks


def _pack_mlps(nets):
    """-> (float32 array (n, stride), widths, acts)"""
    rows, widths, acts = [], None, None
    for blocks in nets:
        w = [blocks[0][2].shape[0]] + [b[2].shape[1] for b in blocks]
        a = [b[4] for b in blocks]
        if widths is not None and (w, a) != (widths, acts):
            raise ValueError("the networks of an ensemble differ in shape")
        widths, acts = w, a
        rows.append(np.concatenate([np.concatenate([inv, off, k.reshape(-1), b]) for inv, off, k, b, _ in blocks]))
    return np.stack(rows).astype(np.float32), widths, acts


class KSTARModel:
    """device copy of the surrogate's networks + the constants of the rollout; ``rollout`` is the batched ``simulate``"""

    def __init__(self, weights, device="cuda", n_model_box=1):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("safediffcon_amd.kstar runs on a GPU (libsdc_hip.so); there is no CPU path")
        self.lib = _lib.get_lib()
        n = int(n_model_box)
        if n < 1 or n > len(weights["lstm"]) or n > len(weights["bpw"]):
            raise ValueError(f"n_model_box={n} but {len(weights['lstm'])} LSTM / {len(weights['bpw'])} bpw networks were loaded")
        lstm_rows, heads = [], []
        for m in weights["lstm"][:n]:
            inv0, off0 = _fold_bn(m["bn0"])
            inv1, off1 = _fold_bn(m["bn1"])
            l0, l1 = m["lstm0"], m["lstm1"]
            lstm_rows.append(np.concatenate([inv0, off0, l0["kernel"].reshape(-1), l0["recurrent_kernel"].reshape(-1), l0["bias"],
                                             inv1, off1, l1["kernel"].reshape(-1), l1["recurrent_kernel"].reshape(-1), l1["bias"]]))
            heads.append(_mlp_blocks([("bn", m["bn2"]), ("dense", dict(m["dense0"], activation="sigmoid")),
                                      ("bn", m["bn3"]), ("dense", dict(m["dense1"], activation="linear"))]))
        lstm = np.stack(lstm_rows).astype(np.float32)
        if lstm.shape[1] != self.lib.sdc_kstar_lstm_floats():
            raise ValueError(f"LSTM parameter count {lstm.shape[1]} != sdc_kstar_lstm_floats() {self.lib.sdc_kstar_lstm_floats()}")
        packs = {"head": _pack_mlps(heads), "steady": _pack_mlps([_mlp_blocks(weights["nn"][0]["layers"])]),
                 "bpw": _pack_mlps([_mlp_blocks(m["layers"]) for m in weights["bpw"][:n]])}
        self._bufs = {"lstm": torch.from_numpy(lstm).to(self.device)}
        d = _lib.SdcKstarModel()
        d.n_lstm = d.n_bpw = n
        d.lstm, d.lstm_stride = self._bufs["lstm"].data_ptr(), lstm.shape[1]
        for name, (arr, widths, acts) in packs.items():
            t = torch.from_numpy(arr).to(self.device)
            self._bufs[name] = t
            mlp = getattr(d, name)
            if len(acts) > 6 or max(widths) > 256:
                raise ValueError(f"{name} network: at most 6 Dense layers of width <= 256")
            mlp.nlayers = len(acts)
            for i, wv in enumerate(widths):
                mlp.width[i] = wv
            for i, av in enumerate(acts):
                mlp.act[i] = av
            mlp.params, mlp.stride = t.data_ptr(), arr.shape[1]
        scale = _scale()
        for i in range(4):
            d.lstm_ystd[i], d.lstm_ymean[i], d.nn_ystd[i], d.nn_ymean[i] = LSTM_YSTD[i], LSTM_YMEAN[i], NN_YSTD[i], NN_YMEAN[i]
        for i in range(2):
            d.bpw_ystd[i], d.bpw_ymean[i] = BPW_YSTD[i], BPW_YMEAN[i]
        d.scale = scale
        for i, v in enumerate(INPUT_INIT):
            d.inputs0[i] = float(int(v * scale) / scale)     # i2f(f2i(v)), kstar_solver.py:107-113,150-152
        for i in range(9):
            d.low_action[i], d.high_action[i] = LOW_ACTION[i], HIGH_ACTION[i]
        d.year_in = YEAR_IN
        self.desc = d
        self._work = torch.empty(4, dtype=torch.float64, device=self.device)

    def rollout(self, actions, n_steps=N_STEPS):
        """actions: float32 CUDA tensor viewed as (B, >= n_steps, 9) -- any strides, e.g. ``diffused[:, 3:, :nt-1].permute(0, 2, 1)``
        -> (B, n_steps + 1, 8) float64 rows [βn, βp, h89, h98, q95, q0, li, wmhd] (KSTARSolver.simulate per sample)"""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.float32):
            raise TypeError("actions must be a float32 CUDA tensor")
        if actions.dim() != 3 or actions.shape[2] != 9 or actions.shape[1] < n_steps:
            raise IndexError(f"actions must be (B, >= {n_steps}, 9), got {tuple(actions.shape)}")      # actions[idx] past the end
        if not bool(torch.isfinite(actions[:, :n_steps]).all()):
            raise ValueError("cannot convert float NaN to integer")                                  # int() in f2i, kstar_solver.py:111
        B = actions.shape[0]
        out = torch.empty((B, n_steps + 1, 8), dtype=torch.float64, device=actions.device)
        if B == 0:
            return out
        check(self.lib.sdc_kstar_rollout(C.byref(self.desc), actions.data_ptr(), actions.stride(0), actions.stride(1), actions.stride(2),
                                         out.data_ptr(), self._work.data_ptr(), B, n_steps,
                                         torch.cuda.current_stream(actions.device).cuda_stream), "sdc_kstar_rollout")
        return out


def default_weights_dir():
    """where the reference looks: ``<dir of the running script>/weights`` (kstar_solver.py:31,54-57)"""
    return os.path.join(os.path.abspath(os.path.dirname(sys.argv[0])), "weights")


_MODELS = {}


def get_model(weights=None, device="cuda", n_model_box=1):
    """a cached KSTARModel; ``weights`` = a directory laid out like tokamak/weights, a nested dict from load_weights, or None
    for the reference's default location"""
    if isinstance(weights, KSTARModel):
        return weights
    if weights is None:
        weights = default_weights_dir()
    if isinstance(weights, (str, os.PathLike)):
        key = (os.path.abspath(weights), str(device), n_model_box)
        if key not in _MODELS:
            _MODELS[key] = KSTARModel(load_weights(weights, n_model_box), device, n_model_box)
        return _MODELS[key]
    return KSTARModel(weights, device, n_model_box)


class KSTARSolver:
    """``KSTARSolver(random_seed).simulate(actions)`` of tokamak/kstar_solver.py (the seed only seeds numpy there; nothing in
    ``simulate`` draws from it).  ``simulate`` takes one (T, 9) action array like the reference and returns a numpy (122, 8);
    ``simulate_batch`` keeps a (B, T, 9) CUDA tensor on the device."""

    def __init__(self, random_seed=0, weights=None, device="cuda", n_model_box=1):
        self.random_seed = random_seed
        self.model = get_model(weights, device, n_model_box)

    def simulate_batch(self, actions):
        return self.model.rollout(actions)

    def simulate(self, actions):
        a = torch.as_tensor(np.asarray(actions), dtype=torch.float32, device=self.model.device)
        if a.dim() != 2:
            raise IndexError("actions must be (T, 9)")
        return self.model.rollout(a[None])[0].cpu().numpy()


# ------------------------------------------------------------------------------------------------ metrics.py
def control_trajectories(diffused, nt_total, seed, weights=None, n_model_box=1):
    """tokamak/utils/metrics.py:60-85: channels 3..11 of ``diffused`` (B, C, padded_T; original scale) are the 9 actuators;
    -> (B, 3, nt_total) controlled (βp, q95, li), same dtype / device as ``diffused``.  One launch for the batch."""
    if diffused.shape[1] < 12:
        raise IndexError("diffused needs 3 state + 9 action channels")
    if nt_total - 1 >= N_STEPS and nt_total != N_STEPS + 1:
        raise RuntimeError(f"the simulator produces {N_STEPS + 1} rows, nt_total={nt_total}")       # the reference's row assignment cannot broadcast
    model = get_model(weights, diffused.device, n_model_box)
    actions = diffused[:, 3:12, :nt_total - 1].permute(0, 2, 1)
    if actions.dtype != torch.float32:
        actions = actions.float()
    rows = model.rollout(actions, n_steps=N_STEPS)
    return rows[:, :, [1, 4, 6]].permute(0, 2, 1).to(diffused.dtype).contiguous()


def calculate_safety_score(x):
    """metrics.py:144-151: min_t q95"""
    return x[:, 1, :].amin(dim=-1)


def calculate_reported_safe_metric(controlled_q95, threshold):
    """metrics.py:125-142"""
    scores = controlled_q95.min(dim=1)[0]
    ratio = threshold / scores
    safe, unsafe = (scores >= threshold).float(), (scores < threshold).float()
    inside = (ratio * safe).sum() / safe.sum().clamp(min=1)
    outside = (ratio * unsafe).sum() / unsafe.sum().clamp(min=1)
    return (inside + outside).item()


def calculate_safety_metrics(controlled_q95, threshold, diffused_s):
    """metrics.py:87-123"""
    below = controlled_q95 < threshold
    score = controlled_q95.amin(dim=-1)
    return {"time_below_ratio": below.float().mean().item(),
            "sample_below_ratio": below.any(dim=-1).float().mean().item(),
            "safety_score_mean": score.mean().item(),
            "safety_score_std": score.std().item(),
            "diffused_score_mse": (diffused_s.amin(dim=-1) - score).square().mean().item(),
            "reported_safe_metric": calculate_reported_safe_metric(controlled_q95, threshold)}


def evaluate_samples(diffused, state_controlled, state_target, safety_threshold, dataset):
    """metrics.py:11-52 (``dataset`` only supplies ``nt_total``)"""
    nt = dataset.nt_total if hasattr(dataset, "nt_total") else int(dataset)
    m = {}
    d_mse = (state_controlled - diffused[:, :3, :nt]).square().mean((-1, -2))
    m["diffusion_mse_mean"], m["diffusion_mse_std"] = d_mse.mean().item(), d_mse.std().item()
    bp = (state_target[:, 0, :nt] - state_controlled[:, 0, :nt]).square().mean(-1)
    li = (state_target[:, 2, :nt] - state_controlled[:, 2, :nt]).square().mean(-1)
    m["beta_p_mse_mean"], m["beta_p_mse_std"] = bp.mean().item(), bp.std().item()
    m["l_i_mse_mean"], m["l_i_mse_std"] = li.mean().item(), li.std().item()
    m["obj_mse_mean"] = m["beta_p_mse_mean"] + m["l_i_mse_mean"]
    m["obj_mse_std"] = (bp + li).std().item()
    m.update(calculate_safety_metrics(state_controlled[:, 1, :], safety_threshold, diffused[:, 1, :nt]))
    return m
