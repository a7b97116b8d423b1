"""The CPU oracle (oracle/) against fixtures produced by the REAL reference
(oracle/make_goldens.py).  CPU-only; these pin the checker the GPU tests use."""
import numpy as np
import pytest
import torch

from oracle import nets, samplers, schedules
from oracle.detweights import det_noise, det_params

TOL = dict(rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("kind", ["cosine", "linear", "sigmoid"])
def test_schedule_tables_bit_exact(golden, kind):
    g = golden("schedule_" + kind)
    tabs = schedules.make_tables(kind, 1000)
    for k in g.keys():
        assert torch.equal(tabs[k], g[k]), k


def _check_stages(g, got, tol):
    for k in g.keys():
        if k.startswith("stage:") and k[6:] in got:
            torch.testing.assert_close(got[k[6:]], g[k], **tol)


def test_unet_burgers_eps(golden):
    g = golden("burgers_unet")
    P = det_params(g.spec(), 100)
    eps = nets.unet_burgers(P, g["x"], g["t"], dim=int(g.scalar("dim")))
    torch.testing.assert_close(eps, g["eps"], **TOL)
    temb = nets.time_mlp(nets._View(P, "time_mlp."), g["t"], int(g.scalar("dim")))
    torch.testing.assert_close(temb, g["temb"], **TOL)


def test_unet_tokamak_eps(golden):
    g = golden("tokamak_unet")
    P = det_params(g.spec(), 200)
    eps = nets.unet_tokamak(P, g["x"], g["t"], dim=int(g.scalar("dim")))
    torch.testing.assert_close(eps, g["eps"], **TOL)


def test_unet_smoke_eps(golden):
    g = golden("smoke_unet")
    P = det_params(g.spec(), 300)
    eps = nets.unet_smoke(P, g["x"], g["t"], dim=int(g.scalar("dim")))
    torch.testing.assert_close(eps, g["eps"], **TOL)


def test_relpos_bias(golden):
    g = golden("smoke_relpos")
    assert torch.equal(nets.rel_pos_bias(g["weight"], 32), g["bias32"])
    assert torch.equal(nets.rel_pos_bias(g["weight"], 8), g["bias8"])


def test_smoke_full_spec_is_236_keys(golden):
    assert len(golden("smoke_fullspec").spec()) == 236


@pytest.mark.parametrize("tag,ums", [("mean", True), ("amax", False)])
def test_burgers_guidance(golden, tag, ums):
    g = golden("burgers_guidance_" + tag)
    Q, w, ub = g.scalar("Q"), g.scalar("w_score"), g.scalar("u_bound")
    grad = samplers.burgers_guidance(Q, w, ub, ums)(g["x"])
    torch.testing.assert_close(grad, g["grad"], rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(samplers.burgers_J(g["x"], Q, w, ub, ums), g["J"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(samplers.burgers_weight(g["x"], Q, w, ub, ums), g["weight"], rtol=1e-5, atol=1e-30)


@pytest.mark.parametrize("tag", ["safe", "mixed"])
def test_tokamak_guidance(golden, tag):
    g = golden("tokamak_guidance_" + tag)
    a = [g.scalar(k) for k in ("Q", "thr", "w_obj", "w_safe")]
    fn = samplers.tokamak_guidance(g["target"], 122, *a, g.scalar("scaler"))
    torch.testing.assert_close(fn(g["x"]), g["grad"], rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(samplers.tokamak_J(g["x"], g["target"], 122, *a), g["J"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(samplers.tokamak_weight(g["x"], g["target"], 122, *a, g.scalar("scaler")),
                               g["weight"], rtol=1e-6, atol=0)


def test_smoke_guidance_and_conformal(golden):
    g = golden("smoke_guidance")
    Q, ws, sb, r = (g.scalar(k) for k in ("Q", "w_safe", "safe_bound", "ratio"))
    torch.testing.assert_close(samplers.smoke_guidance(Q, ws, sb)(g["x"]), g["grad"], rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(samplers.smoke_J(g["x"], Q, ws, sb), g["J"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(samplers.smoke_weight(g["x"], Q, ws, sb, r), g["weight"], rtol=1e-5, atol=0)
    for k in ("plain", "inf", "zero"):
        assert torch.equal(samplers.normalize_weights(g["nw_in_" + k], smoke=True), g["nw_out_" + k]), k
    for a in (0.04, 0.01, 0.5):
        assert samplers.quantile_smoke(g["q_scores"], a) == g[f"q_{a}"]


def test_burgers_conformal(golden):
    g = golden("burgers_conformal")
    for k in ("plain", "inf", "zero"):
        assert torch.equal(samplers.normalize_weights(g["nw_in_" + k]), g["nw_out_" + k]), k
    for a in (0.98, 0.9, 0.5, 0.05):
        assert samplers.quantile_lucid(g["q_scores"], a) == g[f"q_{a}"]


def _eps_fn(net, P, dim):
    return lambda x, t: net(P, x, t, dim=dim)


def _spec(golden, name):
    return golden(name).spec()


def test_burgers_trajectories(golden):
    spec = _spec(golden, "burgers_unet")
    g = golden("burgers_traj_guided")
    P = det_params(spec, int(g.scalar("weight_seed")))
    tabs = schedules.make_tables("cosine", int(g.scalar("T")))
    noise = det_noise((2, 3, 16, 128), int(g.scalar("noise_seed")))
    eps = _eps_fn(nets.unet_burgers, P, int(g.scalar("dim")))
    nablaJ = samplers.burgers_guidance(g.scalar("Q"), g.scalar("w_score"), g.scalar("u_bound"))
    out = samplers.sample_burgers(eps, tabs, 2, noise, u_init=g["u0"], u_final=g["uT"], nablaJ=nablaJ,
                                  J_scheduler=lambda t: 1.0, guidance_u0=True, enable_grad=False)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    # guidance must actually have mattered in this fixture
    out0 = samplers.sample_burgers(eps, tabs, 2, noise, u_init=g["u0"], u_final=g["uT"], nablaJ=None,
                                   guidance_u0=True, enable_grad=False)
    assert (out0 - g["out"]).abs().max() > 1e-3

    g = golden("burgers_traj_calib")
    out = samplers.sample_burgers(eps, tabs, 2, noise, u_init=g["u0"], u_final=g["uT"], nablaJ=None,
                                  guidance_u0=False, w_groundtruth=g["w_gt"], enable_grad=False)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    assert int(g.scalar("draws")) == 1 + 2 * 7


def test_tokamak_trajectories(golden):
    spec = _spec(golden, "tokamak_unet")
    g = golden("tokamak_traj_guided")
    P = det_params(spec, int(g.scalar("weight_seed")))
    tabs = schedules.make_tables("cosine", int(g.scalar("T")))
    noise = det_noise((2, 12, 128), int(g.scalar("noise_seed")))
    eps = _eps_fn(nets.unet_tokamak, P, int(g.scalar("dim")))
    nablaJ = samplers.tokamak_guidance(g["target"], 122, g.scalar("Q"), g.scalar("thr"), g.scalar("w_obj"),
                                       g.scalar("w_safe"), g.scalar("scaler"))
    out = samplers.sample_tokamak(eps, tabs, 2, noise, u_init=g["u0"], u_final=g["uT"], nablaJ=nablaJ,
                                  J_scheduler=lambda t: 1.0, guidance_u0=True, enable_grad=False)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    g = golden("tokamak_traj_calib")
    out = samplers.sample_tokamak(eps, tabs, 2, noise, u_init=g["u0"], u_final=g["uT"], nablaJ=None,
                                  guidance_u0=False, enable_grad=False)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    # the reference's DDPM + w_groundtruth path raises IndexError; so does the oracle
    assert str(golden("tokamak_wgt_bug")["raised"]) == "IndexError"
    with pytest.raises(IndexError):
        samplers.sample_tokamak(eps, tabs, 2, noise, u_init=g["u0"], u_final=g["uT"],
                                w_groundtruth=torch.zeros(2, 9, 128), guidance_u0=False)


def test_smoke_trajectories(golden):
    spec = _spec(golden, "smoke_unet")
    g = golden("smoke_traj_guided")
    P = det_params(spec, int(g.scalar("weight_seed")))
    tabs = schedules.make_tables("sigmoid", int(g.scalar("T")))
    noise = det_noise((2, 8, 7, 16, 16), int(g.scalar("noise_seed")))
    eps = _eps_fn(nets.unet_smoke, P, int(g.scalar("dim")))
    design = samplers.smoke_guidance(g.scalar("Q"), g.scalar("w_safe"), g.scalar("safe_bound"))
    out = samplers.sample_smoke(eps, tabs, 2, noise, init=g["init"], design_fn=design, ratio=g.scalar("ratio"),
                                shape=(8, 7, 16, 16))
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    g = golden("smoke_traj_calib")
    out = samplers.sample_smoke(eps, tabs, 2, noise, init=g["init"], control=g["control"], shape=(8, 7, 16, 16))
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ DDIM
def test_ddim_trajectories(golden):
    g = golden("burgers_ddim_guided")
    P = det_params(_spec(golden, "burgers_unet"), int(g.scalar("weight_seed")))
    T, S, eta = int(g.scalar("T")), int(g.scalar("S")), g.scalar("eta")
    tabs = schedules.make_tables("cosine", T)
    noise = det_noise((2, 3, 16, 128), int(g.scalar("noise_seed")))
    eps = _eps_fn(nets.unet_burgers, P, int(g.scalar("dim")))
    out = samplers.ddim_burgers(eps, tabs, 2, noise, S=S, eta=eta, u_init=g["u0"], u_final=g["uT"],
                                nablaJ=samplers.burgers_guidance(g.scalar("Q"), g.scalar("w_score"), g.scalar("u_bound")))
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    assert int(g.scalar("draws")) == S                       # x_T + one draw per non-final step
    g = golden("burgers_ddim_calib")
    out = samplers.ddim_burgers(eps, tabs, 2, noise, S=S, eta=eta, u_init=g["u0"], u_final=g["uT"], guidance_u0=False,
                                w_groundtruth=g["w_gt"])
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)

    g = golden("tokamak_ddim_guided")
    P = det_params(_spec(golden, "tokamak_unet"), int(g.scalar("weight_seed")))
    noise = det_noise((2, 12, 128), int(g.scalar("noise_seed")))
    eps = _eps_fn(nets.unet_tokamak, P, int(g.scalar("dim")))
    nablaJ = samplers.tokamak_guidance(g["target"], 122, g.scalar("Q"), g.scalar("thr"), g.scalar("w_obj"),
                                       g.scalar("w_safe"), g.scalar("scaler"))
    out = samplers.ddim_tokamak(eps, tabs, 2, noise, S=S, eta=eta, u_init=g["u0"], u_final=g["uT"], nablaJ=nablaJ)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    g = golden("tokamak_ddim_calib")
    out = samplers.ddim_tokamak(eps, tabs, 2, noise, S=S, eta=eta, u_init=g["u0"], u_final=g["uT"], guidance_u0=False,
                                w_groundtruth=g["w_gt"])
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)

    g = golden("smoke_ddim_guided")
    P = det_params(_spec(golden, "smoke_unet"), int(g.scalar("weight_seed")))
    tabs = schedules.make_tables("sigmoid", T)
    noise = det_noise((2, 8, 7, 16, 16), int(g.scalar("noise_seed")))
    eps = _eps_fn(nets.unet_smoke, P, int(g.scalar("dim")))
    out = samplers.ddim_smoke(eps, tabs, 2, noise, S=S, eta=eta, init=g["init"], ratio=g.scalar("ratio"),
                              design_fn=samplers.smoke_guidance(g.scalar("Q"), g.scalar("w_safe"), g.scalar("safe_bound")),
                              shape=(8, 7, 16, 16))
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    g = golden("smoke_ddim_calib")
    out = samplers.ddim_smoke(eps, tabs, 2, noise, S=S, eta=eta, init=g["init"], control=g["control"], shape=(8, 7, 16, 16))
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)


def test_burgers_rollout_oracle_vs_reference_solver(golden):
    """evaluation rollout: oracle restatement == reference burgers_numeric_solve_free, bit for bit"""
    from oracle import solvers
    g = golden("burgers_rollout")
    traj = solvers.burgers_rollout(g["u0"], g["f"])
    assert traj.shape == (3, 11, 128)
    assert torch.equal(traj, g["traj"])


# ------------------------------------------------------------------ full schedule: T = 1000 through the real coefficient tables
@pytest.mark.parametrize("tree", ["burgers", "tokamak", "smoke"])
def test_full_schedule_trajectory(golden, tree):
    """One 1000-step guided DDPM trajectory per tree produced by the REAL reference (oracle/make_goldens.py gen_long): the
    oracle walks the same 1000-entry tables (posterior_log_variance clamp at t = 0, sqrt_recipm1 at t = 999)."""
    g = golden(tree + "_traj_long")
    T, dim = int(g.scalar("T")), int(g.scalar("dim"))
    assert T == 1000 and int(g.scalar("draws")) == T       # x_T + one draw per step with t > 0
    P = det_params(_spec(golden, tree + "_unet"), int(g.scalar("weight_seed")))
    if tree == "burgers":
        noise = det_noise((2, 3, 16, 128), int(g.scalar("noise_seed")))
        out = samplers.sample_burgers(_eps_fn(nets.unet_burgers, P, dim), schedules.make_tables("cosine", T), 2, noise,
                                      u_init=g["u0"], u_final=g["uT"], guidance_u0=True, enable_grad=False,
                                      nablaJ=samplers.burgers_guidance(g.scalar("Q"), g.scalar("w_score"), g.scalar("u_bound")),
                                      J_scheduler=lambda t: 1.0)
    elif tree == "tokamak":
        noise = det_noise((2, 12, 128), int(g.scalar("noise_seed")))
        nablaJ = samplers.tokamak_guidance(g["target"], 122, g.scalar("Q"), g.scalar("thr"), g.scalar("w_obj"),
                                           g.scalar("w_safe"), g.scalar("scaler"))
        out = samplers.sample_tokamak(_eps_fn(nets.unet_tokamak, P, dim), schedules.make_tables("cosine", T), 2, noise,
                                      u_init=g["u0"], u_final=g["uT"], nablaJ=nablaJ, J_scheduler=lambda t: 1.0,
                                      guidance_u0=True, enable_grad=False)
    else:
        noise = det_noise((2, 8, 7, 16, 16), int(g.scalar("noise_seed")))
        out = samplers.sample_smoke(_eps_fn(nets.unet_smoke, P, dim), schedules.make_tables("sigmoid", T), 2, noise,
                                    init=g["init"], ratio=g.scalar("ratio"), shape=(8, 7, 16, 16),
                                    design_fn=samplers.smoke_guidance(g.scalar("Q"), g.scalar("w_safe"), g.scalar("safe_bound")))
    err = (out - g["out"]).abs().max().item()
    print(f"{tree}: oracle vs reference after {T} steps: max|diff| {err:.3e} (|out| <= {g['out'].abs().max().item():.2f})")
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=2e-5)
