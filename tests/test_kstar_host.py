"""CPU side of the tokamak score check: the HDF5 reader against the reference's Keras weight files, the
weights fixture, and the oracle's restatement of the Keras networks (oracle/kstar.py; parity unpinned -- TensorFlow is not in
this image -- so what can be checked here is consistency: the two independently trained surrogates agree with each other)."""
import os

import numpy as np
import pytest

from oracle import kstar as okstar
from safediffcon_amd import h5lite, kstar

GOLD = os.path.join(os.path.dirname(__file__), "golden", "kstar_weights.npz")
REF_WEIGHTS = "/root/reference/tokamak/weights"
NOMINAL = np.array([0.5, 1.5, 1.5, 0.5, 1.7, 0.3, 0.75, 1.32, 2.22], np.float32)      # input_init restricted to the 9 actuators


def _weights():
    return kstar.unflatten_weights(dict(np.load(GOLD)))


@pytest.mark.skipif(not os.path.isdir(REF_WEIGHTS), reason="the reference's weight files are not on this machine")
def test_h5lite_reads_the_keras_files_and_the_fixture_is_their_content():
    w = kstar.load_weights(REF_WEIGHTS, n_models=2)
    flat, gold = kstar.flatten_weights(w), dict(np.load(GOLD))
    assert sorted(flat) == sorted(gold)
    for k in flat:
        a, b = np.asarray(flat[k]), gold[k]
        assert a.shape == b.shape and (a == b).all() if a.dtype.kind == "f" else str(a) == str(b), k
    # the file's own description of itself, through attributes stored as variable-length strings in the global heap
    f = h5lite.File(os.path.join(REF_WEIGHTS, "lstm", "v220505", "best_model0"))
    assert f.attrs["keras_version"] == b"2.2.4-tf" and f.attrs["backend"] == b"tensorflow"
    assert sorted(f.keys()) == ["model_weights", "optimizer_weights"]
    ds = f["model_weights"].visit_datasets()
    assert ds["lstm/lstm/recurrent_kernel:0"].shape == (100, 400) and ds["lstm/lstm/kernel:0"][...].dtype == np.float32
    # every network of every ensemble parses (10 LSTM, 10 nn, 10 bpw)
    kstar.load_weights(REF_WEIGHTS, n_models=10)


@pytest.mark.skipif(not os.path.isdir(REF_WEIGHTS), reason="the reference's weight files are not on this machine")
def test_h5lite_reports_a_truncated_file_as_such(tmp_path):
    blob = open(os.path.join(REF_WEIGHTS, "bpw", "best_model0"), "rb").read()
    for cut in (100, 3000, 20000):
        p = tmp_path / f"cut{cut}.h5"
        p.write_bytes(blob[:cut])
        with pytest.raises(h5lite.H5Error):
            f = h5lite.File(p)
            [d[...] for d in f["model_weights"].visit_datasets().values()]


def test_h5lite_rejects_what_it_does_not_read(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(h5lite.H5Error):
        h5lite.File(p)


def test_fixture_round_trip_and_shapes():
    w = _weights()
    assert len(w["lstm"]) == 2 and len(w["bpw"]) == 2 and len(w["nn"]) == 1
    assert [k for k, _ in w["nn"][0]["layers"]] == ["bn", "dense", "bn", "dense", "bn", "dense", "bn", "dense"]
    assert [p["activation"] for k, p in w["bpw"][0]["layers"] if k == "dense"] == ["sigmoid", "sigmoid", "linear"]
    again = kstar.unflatten_weights(kstar.flatten_weights(w))
    assert (again["lstm"][1]["lstm1"]["recurrent_kernel"] == w["lstm"][1]["lstm1"]["recurrent_kernel"]).all()


def test_quantisation_constants_follow_the_reference_arithmetic():
    # decimals = np.log10(1000); the scale is 10 ** decimals as this numpy evaluates it, int() truncates toward zero
    assert okstar.SCALE == float(10 ** np.log10(1000)) == kstar._scale()
    assert okstar.f2i(0.5) == int(0.5 * okstar.SCALE) and okstar.f2i(1.7999) == int(1.7999 * okstar.SCALE)
    assert okstar.f2i(0.3339) == 333 and okstar.i2f(333) == 333 / okstar.SCALE and okstar.i2f(okstar.f2i(0.0)) == 0.0


def test_oracle_rollout_is_physical_and_the_two_surrogates_agree():
    """constant nominal actuators: the LSTM surrogate (rows 1..121) must settle close to what the separately trained
    steady-state network says for the same actuators (row 0) -- a wrong gate order, activation or weight pairing breaks this"""
    rows = okstar.KSTARSolver(_weights()).simulate(np.tile(NOMINAL, (121, 1)))
    assert rows.shape == (122, 8) and np.isfinite(rows).all()
    bn, bp, h89, h98, q95, q0, li, wmhd = rows.T
    assert (1.0 < bn).all() and (bn < 4.0).all() and (3.0 < q95).all() and (q95 < 8.0).all() and (0.5 < li).all() and (li < 1.5).all()
    assert (0.5 < bp).all() and (bp < 3.5).all() and (1e5 < wmhd).all() and (wmhd < 6e5).all() and (0.5 < h98).all() and (h98 < 3).all()
    assert abs(q95[-1] - q95[0]) < 0.05 * q95[0] and abs(li[-1] - li[0]) < 0.08 * li[0] and abs(bn[-1] - bn[0]) < 0.2 * bn[0]
    # more heating -> more stored energy; more current -> lower q95 (the physics the surrogate learnt)
    hot, cold = NOMINAL.copy(), NOMINAL.copy()
    hot[1:4], cold[1:4] = [1.75, 1.75, 1.5], [0.5, 0.5, 0.0]
    r_hot = okstar.KSTARSolver(_weights()).simulate(np.tile(hot, (121, 1)))
    r_cold = okstar.KSTARSolver(_weights()).simulate(np.tile(cold, (121, 1)))
    assert r_hot[-1, 7] > r_cold[-1, 7] * 1.2
    hi_ip, lo_ip = NOMINAL.copy(), NOMINAL.copy()
    hi_ip[0], lo_ip[0] = 0.8, 0.4
    assert okstar.KSTARSolver(_weights()).simulate(np.tile(hi_ip, (121, 1)))[-1, 4] < okstar.KSTARSolver(_weights()).simulate(np.tile(lo_ip, (121, 1)))[-1, 4]


def _tracking_error(rows, targets):
    """mean |controlled - target| / nominal over the last 10 steps of every 30-step target window, per (beta_p, q95, l_i)"""
    got = rows[1:, [1, 4, 6]]
    settled = np.arange(121) % 30 >= 20
    return np.mean(np.abs(got - targets)[settled] / np.array(okstar.TARGET_INIT), axis=0)


def test_the_references_trained_controller_tracks_its_targets_on_the_restated_simulator():
    """Behavioural anchor for a row whose numbers cannot be pinned (no TensorFlow): the reference ships the actor of the RL
    controller it trained AGAINST ITS OWN simulator (weights/rl/rt_control/3frame_v220505, evaluated in numpy by
    common/model_structure.py:178-204) and generated its dataset by closing that loop
    (kstar_data_generator_random_target.py:433-520).  A feed-forward actor without integral action lands on random targets
    only if the plant in the loop has the steady-state map it was trained on: on the restated simulator it does, to 1-2 %;
    with the LSTM gates read in another order, or another BatchNormalization epsilon, it misses by 15-80 %."""
    w = _weights()
    policy = dict(np.load(os.path.join(os.path.dirname(GOLD), "kstar_rl_policy.npz")))
    for seed in (0, 1):
        actions, rows, targets = okstar.closed_loop(w, policy, seed=seed)
        err = _tracking_error(rows, targets)
        print(f"[measured] closed loop seed {seed}: mean tracking error (beta_p, q95, l_i) = {np.round(err * 100, 2)} %")
        assert (err < 0.03).all()
        assert (actions >= np.array(okstar.LOW_ACTION) - 1e-12).all() and (actions <= np.array(okstar.HIGH_ACTION) + 1e-12).all()

    # negative controls: the same loop around a mis-restated network
    def swapped_gates(xs, wt, rs):
        units = wt["recurrent_kernel"].shape[0]
        perm = np.r_[0:2 * units, 3 * units:4 * units, 2 * units:3 * units]          # i, f, o, c instead of i, f, c, o
        return real_lstm(xs, {k: (v[..., perm] if k != "x" else v) for k, v in wt.items()}, rs)

    real_lstm, real_bn = okstar.lstm, okstar.batchnorm
    try:
        okstar.lstm = swapped_gates
        bad = _tracking_error(*okstar.closed_loop(w, policy, seed=0)[1:])
        assert bad.max() > 0.15
        okstar.lstm = real_lstm
        okstar.batchnorm = lambda x, bn: real_bn(x, dict(bn, eps=1e-5))
        bad = _tracking_error(*okstar.closed_loop(w, policy, seed=0)[1:])
        assert bad.max() > 0.10
    finally:
        okstar.lstm, okstar.batchnorm = real_lstm, real_bn


def test_oracle_ensemble_is_the_mean_of_its_members():
    w = _weights()
    acts = np.tile(NOMINAL, (121, 1))
    x = okstar.KSTARSolver(w).x
    x[:] = np.random.default_rng(0).normal(size=x.shape) * 0.1 + 1.0
    y2 = okstar.ensemble(okstar.lstm_net, x, w["lstm"], 2, okstar.LSTM_YSTD, okstar.LSTM_YMEAN)
    y_each = [okstar.ensemble(okstar.lstm_net, x, [m], 1, okstar.LSTM_YSTD, okstar.LSTM_YMEAN) for m in w["lstm"]]
    assert np.allclose(y2, 0.5 * (y_each[0] + y_each[1]), rtol=0, atol=1e-15)
    assert okstar.KSTARSolver(w, n_model_box=2).simulate(acts).shape == (122, 8)


def test_oracle_metrics():
    q95 = np.array([[5.0, 4.5, 4.9], [4.0, 5.0, 5.0], [6.0, 5.5, 5.2]])
    x = np.stack([np.zeros_like(q95), q95, np.zeros_like(q95)], axis=1)
    assert okstar.calculate_safety_score(x).tolist() == [4.5, 4.0, 5.2]
    # threshold 4.4: two safe (4.4/4.5, 4.4/5.2), one unsafe (4.4/4.0)
    want = (4.4 / 4.5 + 4.4 / 5.2) / 2 + 4.4 / 4.0
    assert abs(okstar.reported_safe_metric(q95, 4.4) - want) < 1e-12


def test_gpu_only_entry_points_refuse_the_cpu():
    with pytest.raises(RuntimeError):
        kstar.KSTARModel(_weights(), device="cpu")


def test_metrics_mirror_on_cpu_tensors():
    """tokamak/utils/metrics.py:11-151 restated in safediffcon_amd.kstar (plain torch, no kernel): against hand-computed values
    and the numpy restatement"""
    import torch
    rng = np.random.default_rng(3)
    B, nt = 6, 122
    diffused = torch.from_numpy(rng.normal(size=(B, 12, 128)).astype(np.float32)) * 0.1 + 5.0
    controlled = torch.from_numpy(rng.normal(size=(B, 3, nt)).astype(np.float32)) * 0.2 + 5.0
    target = torch.from_numpy(rng.normal(size=(B, 3, nt)).astype(np.float32)) * 0.2 + 5.0
    thr = 4.7
    m = kstar.evaluate_samples(diffused, controlled, target, thr, nt)
    c, d, t = controlled.numpy().astype(np.float64), diffused.numpy().astype(np.float64), target.numpy().astype(np.float64)
    dm = ((c - d[:, :3, :nt]) ** 2).mean(axis=(1, 2))
    assert abs(m["diffusion_mse_mean"] - dm.mean()) < 1e-6 and abs(m["diffusion_mse_std"] - dm.std(ddof=1)) < 1e-6
    bp, li = ((t[:, 0] - c[:, 0]) ** 2).mean(-1), ((t[:, 2] - c[:, 2]) ** 2).mean(-1)
    assert abs(m["obj_mse_mean"] - (bp.mean() + li.mean())) < 1e-6 and abs(m["obj_mse_std"] - (bp + li).std(ddof=1)) < 1e-6
    q = c[:, 1]
    assert abs(m["time_below_ratio"] - (q < thr).mean()) < 1e-7 and abs(m["sample_below_ratio"] - (q < thr).any(-1).mean()) < 1e-7
    assert abs(m["safety_score_mean"] - q.min(-1).mean()) < 1e-6
    assert abs(m["diffused_score_mse"] - ((d[:, 1, :nt].min(-1) - q.min(-1)) ** 2).mean()) < 1e-6
    assert abs(m["reported_safe_metric"] - okstar.reported_safe_metric(q, thr)) < 1e-5
    assert np.allclose(kstar.calculate_safety_score(controlled).numpy(), okstar.calculate_safety_score(c), atol=1e-6)


def test_oracle_layers_against_pytorchs_own_layers():
    """an independent implementation of the same published layer definitions: torch.nn.LSTM (ATen; gate order i, f, g, o --
    the same as Keras' i, f, c, o -- sigmoid / tanh), F.batch_norm in eval mode and F.linear, fed the Keras weights of the real
    surrogate.  Catches a slip in the restatement's arithmetic; which activation Keras itself uses is argued in DESIGN.md section 9."""
    import torch
    import torch.nn.functional as F
    w = _weights()["lstm"][0]
    x = np.random.default_rng(7).normal(size=(10, 18)).astype(np.float32)
    # BatchNormalization
    bn = w["bn0"]
    want = F.batch_norm(torch.from_numpy(x), torch.from_numpy(bn["mean"]), torch.from_numpy(bn["var"]), torch.from_numpy(bn["gamma"]),
                        torch.from_numpy(bn["beta"]), training=False, eps=1e-3).numpy()
    got = okstar.batchnorm(x, bn)
    assert np.max(np.abs(got - want)) < 2e-5 * max(1.0, np.abs(want).max())
    # LSTM, both layers chained
    def torch_lstm(wt, n_in):
        m = torch.nn.LSTM(n_in, 100, batch_first=True)
        with torch.no_grad():
            m.weight_ih_l0.copy_(torch.from_numpy(wt["kernel"].T.copy()))
            m.weight_hh_l0.copy_(torch.from_numpy(wt["recurrent_kernel"].T.copy()))
            m.bias_ih_l0.copy_(torch.from_numpy(wt["bias"]))
            m.bias_hh_l0.zero_()
        return m
    v = torch.from_numpy(got)[None]
    with torch.no_grad():
        s1, _ = torch_lstm(w["lstm0"], 18)(v)
        mine1 = okstar.lstm(got, w["lstm0"], True)
        assert np.max(np.abs(s1[0].numpy() - mine1)) < 1e-5
        s2, _ = torch_lstm(w["lstm1"], 100)(torch.from_numpy(okstar.batchnorm(mine1, w["bn1"]))[None])
        mine2 = okstar.lstm(okstar.batchnorm(mine1, w["bn1"]), w["lstm1"], False)
        assert np.max(np.abs(s2[0, -1].numpy() - mine2)) < 1e-5
    # Dense(50, sigmoid)
    d = w["dense0"]
    z = torch.sigmoid(F.linear(torch.from_numpy(mine2), torch.from_numpy(d["kernel"].T.copy()), torch.from_numpy(d["bias"]))).numpy()
    assert np.max(np.abs(z - okstar.dense(mine2, d, "sigmoid"))) < 1e-5
