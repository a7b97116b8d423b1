"""bench_extras.py -- everything `bench.py` reports BESIDE its headline line, written to the side file the headline names
(`extra_file`): the calibration-mode sampler, the other BASELINE configs and shipped widths, the per-rank batches of an 8-way
shard, the PyTorch-ROCm eager strawman, the fine-tuning step, the score checks, the opt-in full runs.  Imported by bench.py
only (never by the package); the legs that time or check against `oracle/` are baselines / checkers outside bench.py's timed
region, like its `cpu_baseline` leg."""
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T_DDPM = 1000
# key -> (workload, U-Net dim (0 = the BASELINE width), batch, why it is there)
EXTRA_WORKLOADS = {
    "c2": ("c2", 0, 256, "BASELINE configs[1]"),
    "c3": ("c3", 0, 128, "BASELINE configs[2]"),
    "c4": ("c4", 0, 64, "BASELINE configs[3]"),
    "c2_turbo": ("c2", 128, 256, "configs[1] with the only shipped-checkpoint net, Unet2D dim 128 (1D/configs/inference_config.py:125-134)"),
    "c3_turbo": ("c3", 128, 128, "configs[2] with Unet1D dim 128 (tokamak/configs/inference_config.py:118-141 'turbo')"),
    "c3_small": ("c3", 64, 128, "configs[2] with Unet1D dim 64 (tokamak/configs/inference_config.py:76, the default)"),
    "c2_shard8": ("c2", 0, 32, "per-rank batch of configs[1] sharded over 8 GPUs (SURVEY 8e)"),
    "c3_shard8": ("c3", 0, 16, "per-rank batch of configs[2] sharded over 8 GPUs (SURVEY 8e)"),
    # the same with net.split_small_grids = True: Cin of the 3-tap convs split over workgroups where the grid leaves CUs idle
    "c2_shard8_split": ("c2", 0, 32, "c2_shard8 with the small-batch plan (net.split_small_grids)", True),
    "c3_shard8_split": ("c3", 0, 16, "c3_shard8 with the small-batch plan (net.split_small_grids)", True),
    # (the narrow shipped nets at the full batch gain nothing from it: c3_small 2.15 -> 2.06 ms/step, c3_turbo 3.63 -> 3.61; their steps
    # are so light that the box's clock management decides more -- one and the same plan has read 2.06 and 4.86 ms/step)
}
ALL_EXTRA_WORKLOADS = "c2,c3,c2_turbo,c3_turbo,c3_small,c2_shard8,c3_shard8,c2_shard8_split,c3_shard8_split"


def strawman(step_seconds, name, batch, steps, dim, dev):
    """the PyTorch-ROCm eager time of the same restatement on this MI355X (MIOpen / rocBLAS / aten kernels), one guided
    denoising step -- what a hipified port of the reference would run"""
    s = step_seconds(name, batch, steps, dim, dev)
    return dict(what="oracle's functional U-Net + autograd guidance + posterior update executed by PyTorch-ROCm eager on the same GPU",
                batch=batch, steps=steps, ms_per_step=round(s * 1e3, 2), ms_per_trajectory_step=round(s * 1e3 / batch, 3),
                value=round(batch / (T_DDPM * s), 4), unit="trajectories/s")


def kstar_score_check(batch, dev):
    """the tokamak score check that follows a C3 sampling pass (BASELINE config 3; tokamak/utils/metrics.py:60-85): the batch's
    control sequences through the KSTAR surrogate, sdc_kstar_rollout against the CPU restatement timed on two trajectories"""
    import numpy as np
    import torch
    from oracle import kstar as okstar                 # cpu_baseline leg only
    from safediffcon_amd import kstar
    w = kstar.unflatten_weights(dict(np.load(os.path.join(ROOT, "tests", "golden", "kstar_weights.npz"))))
    model = kstar.KSTARModel(w, dev)
    lo, hi = torch.tensor(kstar.LOW_ACTION), torch.tensor(kstar.HIGH_ACTION)
    g = torch.Generator().manual_seed(0)
    acts = (lo + (hi - lo) * torch.rand(batch, kstar.N_STEPS, 9, generator=g)).float().to(dev)
    rows = model.rollout(acts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        rows = model.rollout(acts)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    t0 = time.perf_counter()
    want = [okstar.KSTARSolver(w).simulate(acts[i].cpu().numpy()) for i in range(2)]
    cpu_ms = (time.perf_counter() - t0) / 2 * 1e3
    err = max(float(np.max(np.abs(rows[i].cpu().numpy() - want[i]) / np.abs(want[i]).max(axis=0))) for i in range(2))
    return {"what": "control_trajectories: 122-row KSTAR surrogate rollout of every sampled control sequence (one launch)",
            "batch": batch, "hip_ms": round(ms, 2), "trajectories_per_s": round(batch / ms * 1e3, 1),
            "cpu_restatement_ms_per_trajectory": round(cpu_ms, 1), "max_rel_err_vs_cpu_restatement": float(f"{err:.2e}"),
            "parity": "unpinned: the reference's simulator needs TensorFlow (DESIGN.md section 9)"}


def smoke_score_check(batch, dev, cpu_steps=8):
    """the smoke score check that follows a C4 sampling pass (2d/inference_2d.py:407-456 multi_evaluate ->
    2d/dataset/apps/evaluate_solver.py:209-350): every sampled control sequence through the 255-step fluid rollout,
    sdc_smoke_rollout (one launch for the batch) beside the CPU restatement timed on a few steps of one sample"""
    import numpy as np
    import torch
    from oracle import smoke_solver as osolver         # cpu_baseline leg only
    from safediffcon_amd import smoke_solver as ss
    g = torch.Generator().manual_seed(3)
    pred = torch.randn(batch, 32, 7, 64, 64, generator=g) * 0.8
    data = torch.rand(batch, 32, 7, 64, 64, generator=g)
    data[:, 0, 0, 40:, :] = 0
    sim = ss.init_sim_128()
    pd, dd = pred.to(dev), data.to(dev)
    out = ss.solver_out(sim, pd.clone(), dd)             # warm-up (LDS opt-in, label upload)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ss.solver_out(sim, pd.clone(), dd)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    # CPU restatement (bit-identical to the reference solver, tests/test_smoke_solver_oracle.py): `cpu_steps` projections
    # of sample 0, extrapolated to the 255 of a rollout; and the error of the HIP fields against it at the first recorded frame
    T = 8 * (cpu_steps // 8 + 1)
    p0, d0 = pred[0].numpy().copy(), data[0].numpy()
    p0[:, 3:5, 8:56, 8:56] = 0
    t0 = time.perf_counter()
    want = osolver.solver(osolver.init_velocity(), d0[0, 0], p0[:T // 8, 3], p0[:T // 8, 4], T)
    cpu_s_per_step = (time.perf_counter() - t0) / (T - 1)
    got = out[0].cpu().numpy()
    err_v = float(np.abs(got[1, 1] - want[2][1][..., 0]).max() / max(np.abs(want[2][1]).max(), 1e-30))
    err_d = float(np.abs(got[1, 0] - want[0][1]).max())
    return {"what": "multi_evaluate's solver: 255 steps x (500-iteration float64 CG pressure projection + 3 semi-Lagrangian "
                    "advections + bucket book-keeping) for every sampled control sequence, one launch, one workgroup per sample",
            "batch": batch, "hip_ms_per_batch": round(ms, 1), "trajectories_per_s": round(batch / ms * 1e3, 1),
            "us_per_cg_iteration": round(ms * 1e3 / 255 / 500, 2),
            "cpu_restatement_s_per_trajectory": round(cpu_s_per_step * 255, 1), "cpu_sample": f"{T - 1} steps of one trajectory, 1 thread, x 255 / {T - 1}",
            "reference_runs": "one Python process per trajectory (2d/inference_2d.py:422-447)",
            "max_rel_err_velocity_frame1_vs_cpu_restatement": float(f"{err_v:.2e}"), "max_abs_err_density_frame1": float(f"{err_d:.2e}"),
            "parity": "pinned: oracle bit-identical to fixtures from the reference solver (tests/golden/smoke_solver_*.npz)"}


def finetune_step(name, batch, dim, dev, steps=3, eager=True):
    """One fine-tuning step (loss = mean(w_b p_losses_b); loss.backward(), 2d/inference_2d.py:267-279) through the
    drop-in net's differentiable HIP path, beside the same step of the oracle's functional net under PyTorch-ROCm autograd."""
    import torch
    import safediffcon_amd as sdc
    from oracle import nets as onets
    from oracle.detweights import det_tensor
    torch.manual_seed(0)
    if name == "c2":
        net = sdc.Unet2D(dim=dim or 64, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1).to(dev)
        gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=T_DDPM, temporal=True, use_conv2d=True,
                                          is_condition_u0=True, is_condition_uT=True, condition_idx=10).to(dev)
        shape, fwd, kw = (3, 16, 128), onets.unet_burgers, dict(dim=net.dim)
    elif name == "c3":
        net = sdc.Unet1D(dim=dim or 256, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1).to(dev)
        gd = sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=T_DDPM).to(dev)
        shape, fwd, kw = (12, 128), onets.unet_tokamak, dict(dim=net.dim)
    else:
        net = sdc.Unet3D_with_Conv3D(dim=dim or 64, dim_mults=(1, 2, 4), channels=7).to(dev)
        gd = sdc.GaussianDiffusionSmoke(net, image_size=64, frames=32, timesteps=T_DDPM, loss_type="l2").to(dev)
        shape, fwd, kw = (32, 7, 64, 64), onets.unet_smoke, dict(dim=net.dim, dim_mults=(1, 2, 4))
    state = det_tensor((batch, *shape), 9, 0.3).to(dev)
    w = torch.ones(batch, device=dev)
    t = torch.randint(0, T_DDPM, (batch,), generator=torch.Generator().manual_seed(3)).to(dev)
    noise = det_tensor((batch, *shape), 10).to(dev)

    def hip_step():
        net.zero_grad(set_to_none=True)
        loss = (w * gd.p_losses(state, t, noise=noise, mean=False)).mean()
        loss.backward()
        return loss

    P = {k: v.detach().clone().requires_grad_() for k, v in net.state_dict().items()}

    def eager_step():
        for v in P.values():
            v.grad = None
        x = gd.q_sample(state, t, noise)
        eps = fwd(P, x, t, **kw)
        loss = (w * ((eps - noise) ** 2).flatten(1).mean(1)).mean()
        loss.backward()
        return loss

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    ms_hip = timeit(hip_step)
    # the same step captured once in a hipGraph and replayed (sdc.GraphedLossStep): no host work per launch
    ms_graph = None
    try:
        gstep = sdc.GraphedLossStep(gd, state, weight=w, t=t, noise=noise)
        ms_graph = timeit(lambda: gstep())
        del gstep
    except Exception as e:                           # noqa: BLE001  (report, do not fail the line)
        ms_graph = f"capture failed: {str(e)[:120]}"
    out = dict(what="loss = mean(w_b p_losses_b(state)); loss.backward()  (U-Net forward + backward, all parameter gradients)",
               batch=batch, hip_ms=round(ms_hip, 2), hip_ms_per_sample=round(ms_hip / batch, 2),
               hip_graph_ms=(round(ms_graph, 2) if isinstance(ms_graph, float) else ms_graph),
               backward="every node on libsdc_hip.so kernels in both directions (conv data gradient on the forward Winograd kernels, "
                        "sdc_conv_wgrad, sdc_gn_silu_bwd, sdc_chan_norm_bwd, sdc_attn_bwd, sdc_linattn_bwd, sdc_act_bwd)")
    if not eager:
        return out
    try:
        ms_eager = timeit(eager_step)
        out.update(torch_rocm_autograd_ms=round(ms_eager, 2), speedup=round(ms_eager / ms_hip, 2))
    except RuntimeError as e:
        out["torch_rocm_autograd_error"] = str(e)[:160]
    del P
    torch.cuda.empty_cache()
    return out


def cpu_c1_full(cpu_model):
    """BASELINE configs[0] ("C1") in full on the host cores: Unet2D dim 64, B=16, unguided 1000-step p_sample_loop through the
    CPU oracle ("C1 timed in full").  Minutes of CPU time: run with --cpu-c1-full, not part of the default line."""
    import torch
    from oracle import nets as onets, samplers as osam, schedules as osched
    from oracle.detweights import det_params, det_tensor
    import safediffcon_amd as sdc
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    net = sdc.Unet2D(dim=64, channels=3, resnet_block_groups=1)
    P = det_params([(k, tuple(v.shape)) for k, v in net.state_dict().items()], 0)
    tabs = osched.make_tables("cosine", T_DDPM)
    B = 16
    u0, uT = det_tensor((B, 128), 2, 0.1), det_tensor((B, 128), 3, 0.1)
    g = torch.Generator().manual_seed(2)
    noise = lambda i: torch.randn(B, 3, 16, 128, generator=g)      # noqa: E731
    t0 = time.perf_counter()
    with torch.no_grad():
        out = osam.sample_burgers(lambda a, b: onets.unet_burgers(P, a, b, dim=64), tabs, B, noise, u_init=u0, u_final=uT,
                                  nablaJ=None, enable_grad=False)
    el = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return dict(workload="C1: 1D Burgers Unet2D dim=64, B=16, unguided 1000-step DDPM, CPU oracle (port) in full", seconds=round(el, 2),
                value=round(B / el, 5), unit="trajectories/s", ms_per_step=round(el * 1e3 / T_DDPM, 2), cores=torch.get_num_threads(),
                cpu_model=cpu_model(), logical_cpus_visible=os.cpu_count())


def run_extras(ctx, a, W, wl, B, prec, dev, step_ms):
    """the N = 1 extras of one bench run -> dict (bench.py writes it to the side file).  `ctx` carries bench.py's own harness
    (workload(), timed(), build_roofline(), the oracle step timer of the cpu_baseline leg) so that every extra workload is
    timed exactly like the headline.  Runs on bench.py's side stream under no_grad."""
    import torch
    from safediffcon_amd import conformal
    extra, phases = {}, {}

    def phase(name, t0):
        torch.cuda.synchronize()
        phases[name] = round(time.perf_counter() - t0, 2)

    # calibration pass: the unguided calibration-mode sampler at the reference's calibration batch size, a bounded
    # number of steps (a full pass is cal_batches x 1000 steps), then score -> all-gather -> quantile on its output
    t_ph = time.perf_counter()
    Bc = W["cal_B"]
    Sc = W["calib"](Bc)
    Sc.init()
    dtc = ctx.timed(Sc, 2, a.cal_steps)
    pred = Sc.x.clone()
    Sc.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kind = {"c2": "burgers", "c3": "tokamak", "c4": "smoke"}[wl]
    gpar = {"c2": [500.0, 0.64, 0.0, 10.0], "c3": [0.0, 1.0, 0.01, 4.98, 0.0], "c4": [0.9, 0.1, 0.0, 100.0]}[wl]
    kw = dict(target=torch.ones(Bc, 3, 122, device=dev)) if wl == "c3" else {}
    s_, w_ = conformal.scores_and_weights(kind, pred, pred.flip(0), gpar, **kw)
    Qc = float(conformal.weighted_quantile(s_, w_, W["conformal"]["alpha"], smoke=(wl == "c4"))[0].item())
    tq = time.perf_counter() - t0
    msc = dtc / a.cal_steps * 1e3
    extra["calibration"] = {
        "what": f"calibration-mode sampler (unguided; conditions + ground-truth channels imposed"
                f"{'; two noise draws per step' if wl == 'c2' else ''}) at the reference's calibration batch {Bc}, "
                f"{a.cal_steps} timed steps; then score kernel -> all-gather -> normalise/sort/rank on its output",
        "batch": Bc, "batches_per_pass": W["cal_batches"], "ms_per_step": round(msc, 3),
        "projected_seconds_per_full_pass": round(W["cal_batches"] * T_DDPM * msc / 1e3, 1),
        "score_allgather_quantile_ms": round(tq * 1e3, 3), "Q_on_partial_trajectories": round(Qc, 6)}
    del pred
    phase("calibration", t_ph)
    # the other single-GPU BASELINE configs; with --all-extras also the widths the reference ships besides them (Unet2D dim 128
    # "turbo", 1D/configs/inference_config.py:125-134; Unet1D dim 128 / 64, tokamak/configs/inference_config.py:118-141, :76) and
    # the per-rank batches of an 8-way shard of the 1-D configs ("weight re-reads dominate -- report it"), a few steps
    # each through the same harness
    for key in [w for w in a.extra_workloads.split(",") if w and w != wl]:
        if key not in EXTRA_WORKLOADS:
            raise SystemExit(f"--extra-workloads: unknown entry {key!r} (known: {sorted(EXTRA_WORKLOADS)})")
        t_ph = time.perf_counter()
        other, dim2, B2, why = EXTRA_WORKLOADS[key][:4]
        W2 = ctx.workload(other, dim2, B2, dev, ctx.rank, ctx.world, prec, cal_steps=(a.cal_steps if key in ("c2", "c3", "c4") else 0),
                          split_small=len(EXTRA_WORKLOADS[key]) > 4 and EXTRA_WORKLOADS[key][4])
        S2 = W2["prep"]()
        S2.init()
        dt2 = ctx.timed(S2, 3, a.extra_steps)
        ok2 = bool(torch.isfinite(S2.x).all().item())
        # the dominant kernel of this workload against its roofline, like the headline's (PMC traffic: profiles/r*_pmc_traffic.json)
        r2 = ctx.build_roofline(S2, dt2 / a.extra_steps * 1e3, other)
        S2.close()
        ms2 = dt2 / a.extra_steps * 1e3
        wbytes = 4 * sum(p_.numel() for p_ in W2["gd"].model.parameters())
        extra[key] = {"workload": W2["desc"], "why": why, "batch": B2, "steps": a.extra_steps,
                      "ms_per_step": round(ms2, 4), "ms_per_trajectory_step": round(ms2 / B2, 5),
                      "value": round(B2 / (T_DDPM * dt2 / a.extra_steps), 4), "unit": "trajectories/s", "finite": ok2,
                      # reading every weight once per step at the 8 TB/s HBM peak, as a share of the measured step: what a
                      # weight-bandwidth-bound step would show as ~1 (the packed Winograd taps are 16/9 - 64/27 x larger)
                      "weights_mb": round(wbytes / 1e6, 1),
                      "weight_read_share_of_step_at_hbm_peak": round(wbytes / 8e12 / (ms2 * 1e-3), 4),
                      "roofline": {k: r2[k] for k in r2 if k not in ("stages", "all_kernels", "notes", "top_kernels")},
                      "stages": r2["stages"], "all_kernels": r2["all_kernels"]}
        del W2, S2
        torch.cuda.empty_cache()
        phase(key, t_ph)
    if a.other_precisions:
        def other_prec(mode):
            W["gd"].model.precision = mode
            S2 = W["prep"]()
            S2.init()
            dt_ = ctx.timed(S2, a.warmup, a.steps)
            ok_ = bool(torch.isfinite(S2.x).all().item())
            S2.close()
            W["gd"].model.precision = prec
            return {"value": round(B / (T_DDPM * dt_ / a.steps), 4), "unit": "trajectories/s",
                    "ms_per_step": round(dt_ / a.steps * 1e3, 4), "finite": ok_}
        extra["fp32_direct"] = other_prec(0)
        extra["fp32_wino2d"] = other_prec(3)
    if not a.no_strawman:
        t_ph = time.perf_counter()
        sb = a.strawman_batch or {"c2": 256, "c3": 128, "c4": 8}[wl]
        try:
            st = strawman(ctx.step_seconds, wl, sb, 2, a.dim, dev)
            st["hip_ms_per_trajectory_step"] = round(step_ms / B, 3)
            st["hip_speedup_per_trajectory"] = round((st["ms_per_step"] / sb) / (step_ms / B), 2)
            extra["strawman"] = st
        except RuntimeError as e:                    # e.g. out of memory in the eager net: report, do not fail the line
            extra["strawman"] = {"error": str(e)[:200]}
        torch.cuda.empty_cache()
        phase("strawman", t_ph)
    if not a.no_finetune:
        t_ph = time.perf_counter()
        with torch.enable_grad():
            extra["finetune_step"] = finetune_step(wl, a.finetune_batch or {"c2": 64, "c3": 64, "c4": 4}[wl], a.dim, dev,
                                                   eager=a.all_extras)
            if wl == "c4" and a.all_extras:
                # the 1-D nets' steps (C3 <= 25 ms, C2 <= 22 ms at B = 64, the replayed-hipGraph form
                # `hip_graph_ms` being what a fine-tuning loop runs)
                for other in ("c2", "c3"):
                    torch.cuda.empty_cache()
                    extra[f"finetune_step_{other}"] = finetune_step(other, 64, 0, dev)
        phase("finetune", t_ph)
    if wl == "c3":
        t_ph = time.perf_counter()
        extra["kstar_score_check"] = kstar_score_check(B, dev)
        phase("kstar_score_check", t_ph)
    if wl == "c4" and a.all_extras:
        t_ph = time.perf_counter()
        extra["smoke_score_check"] = smoke_score_check(B, dev)
        phase("smoke_score_check", t_ph)
    extra["phases_s"] = phases
    return extra


def full_sample(W, wl, B):
    """--full-sample: ONE complete 1000-step sample() call (validates value = B / (1000 x step time)); for the smoke workload
    also the pipeline's next call on exactly these trajectories (2d/inference_2d.py:389-456): the score check"""
    import torch
    S3 = W["prep"]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S3.init()
    for _ in range(S3.n_main):
        S3.step()
    S3.final()
    torch.cuda.synchronize()
    full_s = time.perf_counter() - t0
    sampled = S3.x.clone()
    S3.close()
    out = {"seconds_for_one_1000_step_sample": round(full_s, 3), "trajectories_per_s": round(B / full_s, 4),
           "note": "includes x_T draw, conditioning, graph capture and the final eager step"}
    if wl == "c4":
        import numpy as np
        from safediffcon_amd import smoke_solver as ss
        data = torch.zeros_like(sampled)
        data[:, 0, 0] = sampled[:, 0, 0]              # the simulator starts from the imposed frame-0 density
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ss.multi_evaluate(sampled, data, float(W["conformal"]["Q"]), 0.1)
        torch.cuda.synchronize()
        ev_s = time.perf_counter() - t0
        J, safe = res[0], res[1]
        out["score_check_of_the_sampled_batch"] = {
            "seconds": round(ev_s, 3), "finite_objective": int(np.isfinite(J).sum()), "batch": int(B),
            "mean_J_target": float(np.nanmean(J)), "mean_safe_target": float(np.nanmean(safe)),
            "sample_plus_score_check_trajectories_per_s": round(B / (full_s + ev_s), 4),
            "note": "random-init weights: the sampled controls are noise-like, the numbers only show the chain runs end to end"}
    return out


def full_calibration(W, wl, dev):
    """--full-calibration: one complete calibration pass end to end: cal_batches x (1000-step calibration-mode sample) ->
    scores -> quantile"""
    import sys
    import torch
    from safediffcon_amd import conformal
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ss, ws = [], []
    for ib in range(W["cal_batches"]):
        Sc = W["calib"](W["cal_B"])
        Sc.init()
        for i_ in range(Sc.n_main):
            Sc.step()
            if i_ % 250 == 249:                   # (a progress line every ~25 s: long runs must not look hung)
                torch.cuda.synchronize()
                print(f"[full-calibration] batch {ib + 1}/{W['cal_batches']} step {i_ + 1}/{Sc.n_main} "
                      f"{time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
        Sc.final()
        kind = {"c2": "burgers", "c3": "tokamak", "c4": "smoke"}[wl]
        gpar = {"c2": [500.0, 0.64, 0.0, 10.0], "c3": [0.0, 1.0, 0.01, 4.98, 0.0], "c4": [0.9, 0.1, 0.0, 100.0]}[wl]
        kw = dict(target=torch.ones(W["cal_B"], 3, 122, device=dev)) if wl == "c3" else {}
        s_, w_ = conformal.scores_and_weights(kind, Sc.x, Sc.x.flip(0), gpar, **kw)
        ss.append(s_), ws.append(w_)
        Sc.close()
    Qf = float(conformal.weighted_quantile(torch.cat(ss), torch.cat(ws), W["conformal"]["alpha"], smoke=(wl == "c4"))[0].item())
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"n": W["cal_B"] * W["cal_batches"], "seconds": round(el, 2), "Q": round(Qf, 6),
            "note": "complete calibration pass: sampling (1000 steps per batch) + score + quantile"}
