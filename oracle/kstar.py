"""CPU restatement of the tokamak score check -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import it; the product never does).

PARITY UNPINNED.  The reference runs this path through TensorFlow 2.13 / Keras (``tokamak/common/model_structure.py:2``), which
is not installed here and cannot be (no network); the reference ships no stored outputs of the simulator either (the
known-answer check of ``tokamak/kstar_solver.py:435-450`` reads data files that are not in the tree).  What anchors it instead:
the reference's own trained controller, closed around this restatement, reaches its random targets to 1-2 % (``closed_loop``
below, tests/test_kstar_host.py; mis-restated networks miss by 15-80 %).  What this file restates:

  * the Keras layers the surrogate is built from, from Keras' published layer definitions (TF 2.13 ``keras.layers``):
      BatchNormalization at inference   y = x * inv + (beta - mean * inv),  inv = gamma / sqrt(var + eps)  [tf.nn.batch_normalization]
      Dense                             y = act(x @ kernel + bias),  act in {sigmoid, linear}
      LSTM (activation tanh, recurrent_activation sigmoid -- the defaults ``load_custom_model`` builds with,
            model_structure.py:69-81, NOT the hard_sigmoid of the config stored in the file, which load_weights ignores)
            z = x_t @ kernel + h @ recurrent_kernel + bias, gates in the order i, f, c, o;
            c' = sig(z_f) c + sig(z_i) tanh(z_c);  h' = sig(z_o) tanh(c');  h0 = c0 = 0
    all in float32 like Keras' ``predict`` (inputs are cast to float32 on entry);
  * ``kstar_v220505.predict`` / ``kstar_nn.predict`` / ``bpw_nn.predict`` (model_structure.py:100-152): mean over the first
    ``nmodels`` networks of ``prediction * ystd + ymean`` in float64;
  * ``KSTARSolver`` (kstar_solver.py:120-428): ``control`` (clip, integer quantisation ``f2i``), ``predict_0d`` (the steady-state
    network once, then the LSTM on a sliding 10-row window with output feedback, then the (beta_p, W_mhd) network and the
    H-factor formulas), ``simulate`` (1 + 1 + 12 x 10 = 122 rows of [bn, bp, h89, h98, q95, q0, li, wmhd]);
  * ``control_trajectories`` / ``calculate_safety_score`` and the evaluation metrics (tokamak/utils/metrics.py:11-151).

Scalar Python / numpy, one sample at a time, like the reference.  numpy-1.x promotion is assumed where the reference mixes
float32 actions with Python floats (TF 2.13 pins numpy < 1.25): clip and the f2i product are evaluated in float64.
"""
import numpy as np

# ---- constants of kstar_solver.py:29-112
SEQ_LEN = 10
DECIMALS = np.log10(1000)                       # kstar_solver.py:35: the scale is whatever 10 ** np.log10(1000) evaluates to (1000.0 here)
SCALE = float(10 ** DECIMALS)
YEAR_IN = 2021
LOW_ACTION = [0.3, 0.0, 0.0, 0.0, 1.6, 0.15, 0.5, 1.265, 2.14]
HIGH_ACTION = [0.8, 1.75, 1.75, 1.5, 1.95, 0.5, 0.85, 1.36, 2.3]
INPUT_PARAMS = ["Ip", "Bt", "GW", "Pnb1a", "Pnb1b", "Pnb1c", "Pec2", "Pec3", "Zec2", "Zec3", "InMid", "OutMid", "Elon", "UpTri", "LoTri"]
INPUT_INIT = [0.5, 1.8, 0.33, 1.5, 1.5, 0.5, 0.0, 0.0, 0.0, 0.0, 1.32, 2.22, 1.7, 0.3, 0.75]
ACTION_TO_INPUT = [0, 3, 4, 5, 12, 13, 14, 10, 11]      # control(): action i sets input_params[...]
LSTM_YMEAN = [1.4361666, 5.275876, 1.534538, 1.1268075]  # kstar_v220505, model_structure.py:102-104
LSTM_YSTD = [0.7294007, 1.5010427, 0.6472052, 0.2331879]
NN_YMEAN = [1.22379703, 5.2361062, 1.64438005, 1.12040048]   # kstar_nn, :122-124
NN_YSTD = [0.72255576, 1.5622809, 0.96563557, 0.23868018]
BPW_YMEAN = np.array([1.02158800e+00, 1.87408512e+05])      # bpw_nn, :141-142
BPW_YSTD = np.array([6.43390272e-01, 1.22543529e+05])
OUTPUT_ORDER = ["bn", "bp", "h89", "h98", "q95", "q0", "li", "wmhd"]   # output_params2, kstar_solver.py:91


def i2f(i):
    return float(i / SCALE)


def f2i(f):
    return int(f * SCALE)


# ---- Keras layers, float32
def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x, dtype=np.float32))).astype(np.float32)


def batchnorm(x, bn):
    eps = np.float32(bn.get("eps", 1e-3))
    inv = (bn["gamma"] / np.sqrt(bn["var"] + eps)).astype(np.float32)
    return (x * inv + (bn["beta"] - bn["mean"] * inv)).astype(np.float32)


def dense(x, d, activation):
    z = (x @ d["kernel"] + d["bias"]).astype(np.float32)
    return _sigmoid(z) if activation == "sigmoid" else z


def lstm(xs, w, return_sequences):
    """xs (T, in) float32 -> (T, units) or (units,)"""
    units = w["recurrent_kernel"].shape[0]
    h = np.zeros(units, np.float32)
    c = np.zeros(units, np.float32)
    seq = []
    for t in range(xs.shape[0]):
        z = (xs[t] @ w["kernel"] + h @ w["recurrent_kernel"] + w["bias"]).astype(np.float32)
        zi, zf, zc, zo = z[:units], z[units:2 * units], z[2 * units:3 * units], z[3 * units:]
        c = (_sigmoid(zf) * c + _sigmoid(zi) * np.tanh(zc)).astype(np.float32)
        h = (_sigmoid(zo) * np.tanh(c)).astype(np.float32)
        seq.append(h)
    return np.stack(seq) if return_sequences else h


def lstm_net(x, m):
    """load_custom_model((10, 18), [100, 100], [50, 4]) -- model_structure.py:69-81 -- on one (10, 18) window -> (4,)"""
    v = batchnorm(np.asarray(x, np.float32), m["bn0"])
    v = batchnorm(lstm(v, m["lstm0"], True), m["bn1"])
    v = batchnorm(lstm(v, m["lstm1"], False), m["bn2"])
    v = batchnorm(dense(v, m["dense0"], "sigmoid"), m["bn3"])
    return dense(v, m["dense1"], "linear")


def dense_net(x, m):
    """a stored Sequential of BatchNormalization / Dense / Dropout layers (kstar_nn, bpw_nn: models.load_model, :125,:143)"""
    v = np.asarray(x, np.float32)
    for kind, p in m["layers"]:
        if kind == "bn":
            v = batchnorm(v, p)
        elif kind == "dense":
            v = dense(v, p, p["activation"])
    return v


def ensemble(net, x, models, nmodels, ystd, ymean):
    return np.mean([net(x, m).astype(np.float32) * np.asarray(ystd) + np.asarray(ymean) for m in models[:nmodels]], axis=0)


class KSTARSolver:
    """kstar_solver.py:120-428 without the plotting state; ``weights`` = {"lstm": [...], "nn": [...], "bpw": [...]} lists of
    per-network parameter dicts (safediffcon_amd.kstar.load_weights / tests/golden/kstar_weights.npz)"""

    def __init__(self, weights, n_model_box=1):
        self.w = weights
        self.nmodels = n_model_box                      # reset_model_number(): the LSTM and bpw ensembles use the first n
        self.x = np.zeros([SEQ_LEN, 18])
        self.inputs = {p: f2i(v) for p, v in zip(INPUT_PARAMS, INPUT_INIT)}
        self.out = {k: 0.0 for k in OUTPUT_ORDER}

    def control(self, action):
        for i, idx in enumerate(ACTION_TO_INPUT):
            a = np.clip(np.float64(action[i]), LOW_ACTION[i], HIGH_ACTION[i])
            self.inputs[INPUT_PARAMS[idx]] = f2i(a)

    def _inp(self, name):
        return i2f(self.inputs[name])

    def _window_row(self):
        """columns 4..16 of the LSTM input, kstar_solver.py:218-233 / :241-258"""
        names = ["Ip", "Bt", "GW", "Elon", "UpTri", "LoTri", "InMid", "OutMid", "Pnb1a", "Pnb1b", "Pnb1c", "Pec2", "InMid"]
        row = [self._inp(n) for n in names]
        row[11] += self._inp("Pec3")
        row[12] = 1.0 if row[12] > 1.265 + 1.e-4 else 0.0
        return row

    def predict_0d(self, steady):
        if steady:
            names = ["Ip", "Bt", "Pnb1a", "Pnb1b", "Pnb1c", "Pec2", "Pec3", "Zec2", "Zec3", "InMid", "OutMid", "Elon", "UpTri", "LoTri",
                     "InMid", "GW"]
            x = np.zeros(17)
            x[:16] = [self._inp(n) for n in names]
            x[9], x[10] = 0.5 * (x[9] + x[10]), 0.5 * (x[10] - x[9])
            x[14] = 1.0 if x[14] > 1.265 + 1.e-4 else 0.0
            x[16] = YEAR_IN
            y = ensemble(dense_net, x, self.w["nn"], 1, NN_YSTD, NN_YMEAN)        # kstar_nn(n_models=1), :140
            self.x[:, :4] = y
            self.x[:, 4:17] = self._window_row()
            self.x[:, 17] = YEAR_IN
        else:
            self.x[:-1, 4:] = self.x[1:, 4:]
            self.x[-1, 4:17] = self._window_row()
            y = ensemble(lstm_net, self.x, self.w["lstm"], self.nmodels, LSTM_YSTD, LSTM_YMEAN)
            self.x[:-1, :4] = self.x[1:, :4]
            self.x[-1, :4] = y
        self.out["bn"], self.out["q95"], self.out["q0"], self.out["li"] = (float(v) for v in y)
        # (beta_p, W_mhd), kstar_solver.py:270-292
        x = np.array([self.out["bn"]] + [self._inp(n) for n in ["Ip", "Bt", "InMid", "OutMid", "Elon", "UpTri", "LoTri"]])
        x[3], x[4] = 0.5 * (x[3] + x[4]), 0.5 * (x[4] - x[3])
        y = ensemble(dense_net, x, self.w["bpw"], self.nmodels, BPW_YSTD, BPW_YMEAN)
        self.out["bp"], self.out["wmhd"] = float(y[0]), float(y[1])
        # H factors, :325-350
        ip, bt, fgw = self._inp("Ip"), self._inp("Bt"), self._inp("GW")
        ptot = max(self._inp("Pnb1a") + self._inp("Pnb1b") + self._inp("Pnb1c") + self._inp("Pec2") + self._inp("Pec3"), 1.e-1)
        rin, rout, k = self._inp("InMid"), self._inp("OutMid"), self._inp("Elon")
        rgeo, amin = 0.5 * (rin + rout), 0.5 * (rout - rin)
        ne = fgw * 10 * (ip / (np.pi * amin ** 2))
        m = 2.0
        tau89 = 0.038 * ip ** 0.85 * bt ** 0.2 * ne ** 0.1 * ptot ** -0.5 * rgeo ** 1.5 * k ** 0.5 * (amin / rgeo) ** 0.3 * m ** 0.5
        tau98 = 0.0562 * ip ** 0.93 * bt ** 0.15 * ne ** 0.41 * ptot ** -0.69 * rgeo ** 1.97 * k ** 0.78 * (amin / rgeo) ** 0.58 * m ** 0.19
        self.out["h89"] = 1.e-6 * self.out["wmhd"] / ptot / tau89
        self.out["h98"] = 1.e-6 * self.out["wmhd"] / ptot / tau98

    def simulate(self, actions):
        """actions (>= 121, 9) -> (122, 8), kstar_solver.py:389-428"""
        rows = []
        self.predict_0d(True)
        rows.append([self.out[k] for k in OUTPUT_ORDER])
        for idx in range(1 + 12 * 10):
            self.control(actions[idx])
            self.predict_0d(False)
            rows.append([self.out[k] for k in OUTPUT_ORDER])
        return np.array(rows)


def control_trajectories(diffused, nt_total, weights, n_model_box=1):
    """tokamak/utils/metrics.py:60-85 on a numpy (B, C >= 12, T) array -> (B, 3, nt_total) of (beta_p, q95, l_i)"""
    diffused = np.asarray(diffused)
    actions = np.transpose(diffused[:, 3:, :nt_total - 1], (0, 2, 1))
    out = np.zeros((diffused.shape[0], 3, nt_total), dtype=diffused.dtype)
    for b in range(diffused.shape[0]):
        rows = KSTARSolver(weights, n_model_box).simulate(actions[b])
        out[b] = rows[:, [1, 4, 6]].T
    return out


def calculate_safety_score(x):
    """metrics.py:144-151: min over time of q95 (channel 1)"""
    return np.asarray(x)[:, 1, :].min(axis=-1)


def reported_safe_metric(q95, threshold):
    """metrics.py:125-142"""
    scores = np.asarray(q95).min(axis=1)
    ratio = threshold / scores
    safe, unsafe = scores >= threshold, scores < threshold
    inside = (ratio * safe).sum() / max(safe.sum(), 1)
    outside = (ratio * unsafe).sum() / max(unsafe.sum(), 1)
    return float(inside + outside)


# ---- the reference's trained controller in closed loop (what generated its dataset): kstar_data_generator_random_target.py
LOW_TARGET, HIGH_TARGET = [0.8, 4.0, 0.80], [2.1, 7.0, 1.05]             # :69-70
TARGET_INIT = [1.45, 5.5, 0.925]                                         # :101
RAND_TARGET_MINS, RAND_TARGET_MAXS = [1.06, 4.6, 0.85], [1.84, 6.4, 1.00]   # :104-105
LOOKBACK = 3


def rl_policy(policy, observation):
    """SB2_model.predict (common/model_structure.py:191-204; relu hidden layers, tanh head, bavg = 0): 39 -> 9 actuators"""
    low = np.array((LOW_ACTION + LOW_TARGET) * LOOKBACK + LOW_TARGET)
    high = np.array((HIGH_ACTION + HIGH_TARGET) * LOOKBACK + HIGH_TARGET)
    y = 2 * (observation - low) / (high - low) - 1
    for i in range(len(policy["layers"])):
        y = np.maximum(0, y @ policy[f"fc{i}_kernel"] + policy[f"fc{i}_bias"])
    y = np.tanh(y @ policy["dense_kernel"] + policy["dense_bias"])
    return 0.5 * (np.array(HIGH_ACTION) - np.array(LOW_ACTION)) * (y + 1) + np.array(LOW_ACTION)


def closed_loop(weights, policy, seed=0, n_model_box=1, solver_cls=None):
    """random_target_simulation (kstar_data_generator_random_target.py:433-520) without the bookkeeping: the controller sees the
    last three (action, beta_p, q95, l_i) rows and the target, the target changes every 30 steps
    -> (actions (121, 9), rows (122, 8), targets (121, 3))"""
    rng = np.random.default_rng(seed)

    def new_target():
        return [i2f(f2i(rng.uniform(lo, hi))) for lo, hi in zip(RAND_TARGET_MINS, RAND_TARGET_MAXS)]

    s = (solver_cls or KSTARSolver)(weights, n_model_box)
    s.predict_0d(True)
    rows = [[s.out[k] for k in OUTPUT_ORDER]]
    hist = [list(LOW_ACTION) + list(TARGET_INIT)] * LOOKBACK
    target = new_target()
    actions, targets = [], []
    for step in range(1 + 12 * 10):
        a = rl_policy(policy, np.array(sum(hist, []) + target))
        s.control(a)
        s.predict_0d(False)
        hist = hist[1:] + [list(a) + [s.out["bp"], s.out["q95"], s.out["li"]]]
        actions.append(a)
        targets.append(list(target))
        rows.append([s.out[k] for k in OUTPUT_ORDER])
        if step % 30 == 29:
            target = new_target()
    return np.array(actions), np.array(rows), np.array(targets)
