#!/usr/bin/env python3
"""Generate tests/golden/smoke_solver_*.npz by running the REAL reference rollout
(/root/reference/2d/dataset/apps/evaluate_solver.py `solver`, with its vendored PhiFlow 1.x) on seeded inputs.
Build-container only; the fixtures are data (inputs + the arrays the reference returned).

The reference targets Python < 3.10 / numpy < 1.23; oracle/phi_compat.py restores the two removed behaviours it
relies on (no arithmetic is touched) and stubs `imageio`, which the module imports for GIF output only.

    python oracle/make_smoke_solver_fixture.py            # all cases (~2 min)

Cases
  short_a   32 steps (4 control frames), smooth controls, amplitude 1            everything in float64
  short_nan 32 steps, zero smoke: both records are 0/0 = NaN like the reference's
  short_128 16 steps at nx = 128 (space_interval 1), 2 control frames
  full_a    256 steps x 32 control frames at 64 x 64 (the pipeline's shapes), amplitude 1.5
  full_b    same shapes, amplitude 8 + noise, smoke already inside a bucket and the hazard area at t = 0
Full cases store the density fields as float32 (their exact type in the reference), the velocity of every frame as
float32 and three frames as float64, the two records as float64.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.environ.get("SDC_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")
REF_APPS = "/root/reference/2d/dataset/apps"
sys.path.insert(0, ROOT)

F64_FRAMES = (1, 16, 31)


def controls(nt, nx, amp, noise, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:nx, 0:nx] / nx
    c1 = np.stack([amp * np.sin(2 * np.pi * (xx * 2 + 0.1 * f)) * np.cos(2 * np.pi * yy) for f in range(nt)])
    c2 = np.stack([amp * np.cos(2 * np.pi * (yy * 1.5 - 0.07 * f)) * (0.5 + xx) for f in range(nt)])
    c1 = c1 + noise * amp * rng.standard_normal(c1.shape)
    c2 = c2 + noise * amp * rng.standard_normal(c2.shape)
    c1, c2 = c1.astype(np.float32), c2.astype(np.float32)
    lo, hi = nx // 8, nx - nx // 8                      # multi_evaluate zeroes [8:56] at nx = 64 (inference_2d.py:417)
    c1[:, lo:hi, lo:hi] = 0
    c2[:, lo:hi, lo:hi] = 0
    return c1, c2


def density(nx, kind):
    d = np.zeros((nx, nx), np.float32)
    s = nx / 64
    b = lambda y0, y1, x0, x1: (slice(int(y0 * s), int(y1 * s)), slice(int(x0 * s), int(x1 * s)))
    if kind == "blob":
        d[b(9, 17, 24, 40)] = 1.0
        d[b(11, 15, 28, 36)] = 1.5
    elif kind == "wide":
        rng = np.random.default_rng(7)
        d[b(9, 30, 10, 54)] = rng.uniform(0.2, 1.0, d[b(9, 30, 10, 54)].shape).astype(np.float32)
        d[b(57, 62, 28, 36)] = 0.7          # inside the target bucket at t = 0
        d[b(21, 30, 23, 27)] = 0.9          # inside the hazard area at t = 0
    elif kind == "none":
        pass
    return d


CASES = {
    "short_a": dict(T=32, nt=4, nx=64, amp=1.0, noise=0.0, seed=0, dens="blob"),
    "short_nan": dict(T=32, nt=4, nx=64, amp=1.0, noise=0.0, seed=0, dens="none"),
    "short_128": dict(T=16, nt=2, nx=128, amp=2.0, noise=0.05, seed=3, dens="wide"),
    "full_a": dict(T=256, nt=32, nx=64, amp=1.5, noise=0.0, seed=1, dens="blob"),
    "full_b": dict(T=256, nt=32, nx=64, amp=8.0, noise=0.1, seed=2, dens="wide"),
}


def main(which):
    from oracle import phi_compat
    phi_compat.install(REF_APPS)
    import matplotlib
    matplotlib.use("Agg")
    import evaluate_solver as es                       # the reference module, loaded through the compat importer

    sim = es.init_sim_128()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "smoke_solver_domain.npz"),
                        fluid_mask=sim._fluid_mask[0, ..., 0], active_mask=sim._active_mask[0, ..., 0],
                        velocity_mask=sim._velocity_mask.staggered[0],
                        init_velocity=es.init_velocity_()[0],
                        bucket_concat=es.get_bucket_mask()[1], bucket_keep=es.get_bucket_mask()[2],
                        bucket_each=np.stack(es.get_bucket_mask()[0]),
                        safe_concat=es.get_bucket_mask_safe()[1], safe_keep=es.get_bucket_mask_safe()[2],
                        safe_each=np.stack(es.get_bucket_mask_safe()[0]))
    for name, c in CASES.items():
        if which and name not in which:
            continue
        c1, c2 = controls(c["nt"], c["nx"], c["amp"], c["noise"], c["seed"])
        d0 = density(c["nx"], c["dens"])
        out = es.solver(sim, es.init_velocity_(), d0, c1, c2, per_timelength=c["T"])
        dens, zdens, vel, oc1, oc2, rec, rec_s = out
        assert np.array_equal(dens, dens.astype(np.float32)) and np.array_equal(zdens, zdens.astype(np.float32))
        arrs = dict(c1=c1, c2=c2, init_density=d0, per_timelength=np.int64(c["T"]),
                    out_c1=oc1, out_c2=oc2, smoke_out_record=rec[:, 0, 0], smoke_out_safe_record=rec_s[:, 0, 0],
                    record_is_tiled=np.bool_(np.array_equal(rec, np.broadcast_to(rec[:, :1, :1], rec.shape), equal_nan=True)
                                             and np.array_equal(rec_s, np.broadcast_to(rec_s[:, :1, :1], rec_s.shape), equal_nan=True)))
        if c["T"] == 256:
            arrs.update(densitys=dens.astype(np.float32), zero_densitys=zdens.astype(np.float32),
                        velocitys_f32=vel.astype(np.float32), velocitys_f64=vel[list(F64_FRAMES)],
                        f64_frames=np.array(F64_FRAMES))
        else:
            arrs.update(densitys=dens, zero_densitys=zdens, velocitys=vel)
        path = os.path.join(OUT, f"smoke_solver_{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)  records[-1] = {rec[-1, 0, 0]:.6f} / {rec_s[-1, 0, 0]:.6f}",
              flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
