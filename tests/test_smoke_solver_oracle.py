"""The oracle's restatement of the smoke evaluation rollout (oracle/smoke_solver.py) against fixtures produced by the
REAL reference (2d/dataset/apps/evaluate_solver.py `solver` + vendored PhiFlow; oracle/make_smoke_solver_fixture.py).
The restatement follows the reference operation for operation, so the comparison is exact (array_equal)."""
import os

import numpy as np
import pytest

from oracle import smoke_solver as ss

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(G, f"smoke_solver_{name}.npz"))


def test_domain_masks_equal_the_reference_simulation():
    d = _load("domain")
    dom = ss.domain()
    assert np.array_equal(d["fluid_mask"], dom["fluid"]) and np.array_equal(d["active_mask"], dom["fluid"])
    assert np.array_equal(d["velocity_mask"], dom["vmask"])
    assert np.array_equal(d["init_velocity"], ss.init_velocity())
    each, concat, keep = ss.bucket_masks()
    assert np.array_equal(d["bucket_each"], np.stack(each)) and np.array_equal(d["bucket_concat"], concat)
    assert np.array_equal(d["bucket_keep"], keep)
    each, concat, keep = ss.bucket_masks_safe()
    assert np.array_equal(d["safe_each"], np.stack(each)) and np.array_equal(d["safe_concat"], concat)
    assert np.array_equal(d["safe_keep"], keep)


@pytest.mark.parametrize("name", ["short_a", "short_nan", "short_128"])
def test_short_rollouts_bit_exact(name):
    f = _load(name)
    assert bool(f["record_is_tiled"])
    out, iters = ss.solver(ss.init_velocity(), f["init_density"], f["c1"], f["c2"], int(f["per_timelength"]), return_iters=True)
    assert set(iters) == {ss.CG_MAX_ITER}            # the reference's CG never reaches 1e-8 in its 500 iterations
    dens, zdens, vel, oc1, oc2, rec, rec_s = out
    assert np.array_equal(dens, f["densitys"]) and np.array_equal(zdens, f["zero_densitys"])
    assert np.array_equal(vel, f["velocitys"])
    assert np.array_equal(oc1, f["out_c1"]) and np.array_equal(oc2, f["out_c2"])
    assert np.array_equal(rec[:, 0, 0], f["smoke_out_record"], equal_nan=True)
    assert np.array_equal(rec_s[:, 0, 0], f["smoke_out_safe_record"], equal_nan=True)
    if name == "short_nan":
        assert np.isnan(rec).all() and np.isnan(rec_s).all()


def test_full_length_rollout_bit_exact():
    """256 steps x 32 control frames at 64 x 64: the shapes InferencePipeline.multi_evaluate uses (~35 s of numpy)."""
    f = _load("full_b")
    dens, zdens, vel, oc1, oc2, rec, rec_s = ss.solver(ss.init_velocity(), f["init_density"], f["c1"], f["c2"], 256)
    assert np.array_equal(dens.astype(np.float32), f["densitys"]) and np.array_equal(dens, dens.astype(np.float32))
    assert np.array_equal(zdens.astype(np.float32), f["zero_densitys"])
    assert np.array_equal(vel[list(f["f64_frames"])], f["velocitys_f64"])
    assert np.array_equal(vel.astype(np.float32), f["velocitys_f32"])
    assert np.array_equal(rec[:, 0, 0], f["smoke_out_record"]) and np.array_equal(rec_s[:, 0, 0], f["smoke_out_safe_record"])
    assert rec[-1, 0, 0] > 0.2 and rec_s[-1, 0, 0] > 0.3          # smoke does reach the target bucket and the hazard area


def test_multi_evaluate_fields_assembles_the_seven_channels():
    f = _load("short_a")
    B, nt = 1, 4
    pred = np.zeros((B, nt, 7, 64, 64), np.float32)
    data = np.zeros_like(pred)
    pred[0, :, 3], pred[0, :, 4] = f["c1"], f["c2"]
    pred[0, :, 3:5, 20:30, 20:30] = 5.0                 # inside [8:56]: must be ignored (indirect control)
    data[0, 0, 0] = f["init_density"]
    out = ss.multi_evaluate_fields(pred, data, per_timelength=32)
    assert np.array_equal(out[0, :, 0], f["densitys"]) and np.array_equal(out[0, :, 1], f["velocitys"][..., 0])
    assert np.array_equal(out[0, :, 2], f["velocitys"][..., 1]) and np.array_equal(out[0, :, 3], f["out_c1"])
    assert np.array_equal(out[0, :, 5, 0, 0], f["smoke_out_record"], equal_nan=True)
