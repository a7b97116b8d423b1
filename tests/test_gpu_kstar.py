"""Tokamak score check on the GPU: sdc_kstar_rollout / safediffcon_amd.kstar against the CPU restatement of
KSTARSolver.simulate (oracle/kstar.py), on the reference's real surrogate weights (tests/golden/kstar_weights.npz).

Tolerance: the networks run in fp32 on both sides but sum in different orders (numpy's BLAS vs a thread per gate column), and
the LSTM feeds its outputs back 121 times, so rows are compared to 2e-4 relative; the fp64 parts (action quantisation, output
de-normalisation) are exact, which the quantisation test checks separately."""
import os

import numpy as np
import pytest
import torch

from oracle import kstar as okstar
from safediffcon_amd import kstar

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "kstar_weights.npz")
DEV = "cuda:0"
COLS = np.array([2.3, 1.8, 2.2, 1.8, 5.0, 1.9, 0.9, 3.0e5])       # typical magnitude of each output column


def _weights():
    return kstar.unflatten_weights(dict(np.load(GOLD)))


def _actions(B, seed, T=121):
    """smooth random actuator traces around the operating ranges, some samples pushed past the clip bounds"""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(okstar.LOW_ACTION), np.array(okstar.HIGH_ACTION)
    base = rng.uniform(lo, hi, size=(B, 1, 9))
    walk = np.cumsum(rng.normal(size=(B, T, 9)), axis=1) * (hi - lo) * 0.02
    a = base + walk
    a[::3] += (hi - lo) * rng.uniform(-0.6, 0.6, size=(len(a[::3]), 1, 9))       # out of range -> clipped
    return a.astype(np.float32)


def _rel(got, want):
    return float(np.max(np.abs(got - want) / COLS))


@pytest.mark.parametrize("nbox", [1, 2])
def test_rollout_equals_the_oracle(nbox):
    w = _weights()
    acts = _actions(6, 11)
    model = kstar.KSTARModel(w, DEV, n_model_box=nbox)
    got = model.rollout(torch.from_numpy(acts).to(DEV)).cpu().numpy()
    assert got.shape == (6, 122, 8) and got.dtype == np.float64
    worst = 0.0
    for b in range(6):
        want = okstar.KSTARSolver(w, n_model_box=nbox).simulate(acts[b])
        worst = max(worst, _rel(got[b], want))
    print(f"[measured] KSTAR rollout n_model_box={nbox}: max column-relative error vs the oracle {worst:.2e}")
    assert worst < 2e-4


def test_rollout_of_the_controllers_own_action_sequences():
    """the action traces the reference's trained controller produces in closed loop (what its dataset holds, and what a sampled
    control sequence looks like): replayed through sdc_kstar_rollout they give the rows the closed loop saw"""
    w = _weights()
    policy = dict(np.load(os.path.join(os.path.dirname(GOLD), "kstar_rl_policy.npz")))
    runs = [okstar.closed_loop(w, policy, seed=s) for s in (0, 1, 2)]
    acts = torch.from_numpy(np.stack([r[0] for r in runs]).astype(np.float32)).to(DEV)
    got = kstar.KSTARModel(w, DEV).rollout(acts).cpu().numpy()
    worst = 0.0
    for b, (a64, rows, _) in enumerate(runs):
        # the loop ran on float64 actions; replay the float32 ones the tensor holds through the oracle for the comparison
        want = okstar.KSTARSolver(w).simulate(acts[b].cpu().numpy())
        worst = max(worst, _rel(got[b], want))
        assert _rel(want, rows) < 5e-3                    # float32 actions move a few quantisation cells at most
    print(f"[measured] KSTAR rollout of closed-loop action traces: {worst:.2e}")
    assert worst < 2e-4


def test_first_row_is_the_steady_state_network_and_constant_inputs_stay_exact():
    """row 0 comes from kstar_nn on the initial inputs alone: identical for every sample; the fp64 side of every row
    (quantised inputs -> H-factor formula) must agree with the Python arithmetic to fp64 rounding"""
    w = _weights()
    acts = _actions(3, 5)
    got = kstar.KSTARModel(w, DEV).rollout(torch.from_numpy(acts).to(DEV)).cpu().numpy()
    want0 = okstar.KSTARSolver(w).simulate(acts[0])[0]
    assert (got[:, 0] == got[0, 0]).all()
    assert _rel(got[0, 0], want0) < 1e-5                     # year_in = 2021 through a float32 BatchNormalization: x*inv + off cancels 3 digits
    # h89 / wmhd depends only on the quantised inputs: (1e-6 / ptot / tau89) in fp64
    want = okstar.KSTARSolver(w).simulate(acts[1])
    ratio_got, ratio_want = got[1, :, 2] / got[1, :, 7], want[:, 2] / want[:, 7]
    assert np.max(np.abs(ratio_got / ratio_want - 1.0)) < 1e-12


def test_action_quantisation_clip_and_truncation():
    """control(): np.clip to the actuator range, then int(a * scale) toward zero (kstar_solver.py:107-113,352-378): actions that
    differ below the 1e-3 grid give the same trajectory, actions outside the range act like the bound"""
    w = _weights()
    model = kstar.KSTARModel(w, DEV)
    base = np.tile(np.array([0.5, 1.5, 1.5, 0.5, 1.7, 0.3, 0.75, 1.32, 2.22], np.float32), (121, 1))
    a = np.stack([base, base, base, base])
    a[1] += 0.0004                       # same grid cell after truncation
    a[2, :, 0] = 5.0                     # above the bound -> 0.8
    a[3, :, 0] = 0.8
    got = model.rollout(torch.from_numpy(a).to(DEV)).cpu().numpy()
    cell = np.floor(np.float64(np.float32(1.5) + np.float32(0.0004)) * okstar.SCALE) == np.floor(1.5 * okstar.SCALE)
    assert cell and (got[0] == got[1]).all()
    assert (got[2] == got[3]).all() and not (got[0] == got[2]).all()


def test_all_kernel_shapes_and_ragged_batches_agree():
    """B <= 256 runs one trajectory per workgroup, B <= 512 two, larger batches four: the same trajectories must come out
    bit-identical whatever the batch they ride in, with batches that are not multiples of the group size"""
    w = _weights()
    model = kstar.KSTARModel(w, DEV)
    acts = torch.from_numpy(_actions(515, 3)).to(DEV)
    big = model.rollout(acts)                      # four per workgroup, 3 trajectories in the last one
    mid = model.rollout(acts[:301])                # two per workgroup, 1 in the last one
    small = model.rollout(acts[:7])                # one per workgroup
    assert torch.equal(big[:301], mid) and torch.equal(big[:7], small)
    assert torch.equal(big[512:], model.rollout(acts[512:515]))
    want = okstar.KSTARSolver(w).simulate(acts[514].cpu().numpy())
    assert _rel(big[514].cpu().numpy(), want) < 2e-4


def test_control_trajectories_reads_the_sample_tensor_in_place():
    """metrics.py:60-85 on a (B, 12, 128) sample: channels 3.. are the actuators, read through strides (no permute copy)"""
    w = _weights()
    B, nt = 5, 122
    acts = _actions(B, 21)
    diffused = torch.zeros(B, 12, 128, device=DEV)
    diffused[:, 3:, :121] = torch.from_numpy(acts).to(DEV).permute(0, 2, 1)
    diffused[:, :3] = torch.randn(B, 3, 128, device=DEV)
    got = kstar.control_trajectories(diffused, nt, seed=0, weights=kstar.KSTARModel(w, DEV))
    want = okstar.control_trajectories(diffused.cpu().numpy(), nt, w)
    assert got.shape == (B, 3, nt) and got.dtype == diffused.dtype and got.device == diffused.device
    err = np.max(np.abs(got.cpu().numpy() - want) / np.array([1.8, 5.0, 0.9])[None, :, None])
    print(f"[measured] control_trajectories vs the oracle: {err:.2e}")
    assert err < 2e-4
    s_got, s_want = kstar.calculate_safety_score(got).cpu().numpy(), okstar.calculate_safety_score(want)
    assert np.max(np.abs(s_got - s_want)) < 1e-3
    m = kstar.calculate_safety_metrics(got[:, 1], 4.8, diffused[:, 1, :nt])
    assert abs(m["reported_safe_metric"] - okstar.reported_safe_metric(got[:, 1].cpu().numpy().astype(np.float64), 4.8)) < 1e-5
    ev = kstar.evaluate_samples(diffused, got, diffused[:, :3, :nt] * 0 + got, 4.8, nt)
    assert ev["beta_p_mse_mean"] == 0.0 and ev["obj_mse_mean"] == 0.0 and set(m) <= set(ev)


def test_error_behaviour_follows_the_python():
    w = _weights()
    model = kstar.KSTARModel(w, DEV)
    ok = torch.from_numpy(_actions(2, 1)).to(DEV)
    with pytest.raises(IndexError):                       # actions[idx] past the end, kstar_solver.py:403-424
        model.rollout(ok[:, :100])
    bad = ok.clone()
    bad[1, 50, 3] = float("nan")
    with pytest.raises(ValueError):                       # int(nan) in f2i
        model.rollout(bad)
    with pytest.raises(TypeError):
        model.rollout(ok.double())
    with pytest.raises(RuntimeError):                     # the (3, 122) rows do not fit a (3, nt_total) slot for any other nt_total
        kstar.control_trajectories(torch.zeros(2, 12, 128, device=DEV), 123, 0, weights=model)
    solver = kstar.KSTARSolver(0, weights=model)
    rows = solver.simulate(ok[0].cpu().numpy())
    assert rows.shape == (122, 8) and isinstance(rows, np.ndarray)
