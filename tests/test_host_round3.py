"""CPU-side checks added in round 3: what the N-GPU bench line is quoted on (C5 naming), the fail-fast of `bench.py --gpus N`
without N devices, the kernel-source stamp of the PMC records."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_n8_line_is_c5_with_batch_512_and_200_calibration_samples():
    import bench
    c = bench.shard_config("c4", 64, 8)
    assert c["tag"] == "C5" and c["global_batch"] == 512 and c["n_cal"] == 200 and c["n_cal_per_rank"] == 25 and c["alpha"] == 0.04
    c = bench.shard_config("c4", 64, 1)
    assert c["tag"] == "C4" and c["global_batch"] == 64 and c["n_cal"] == 200
    for n in (2, 4):
        c = bench.shard_config("c4", 64, n)
        assert c["tag"] == "C4" and c["global_batch"] == 64 * n and c["n_cal"] == 200
    assert bench.shard_config("c2", 256, 4)["n_cal"] == 1000


def test_bench_gpus_n_fails_fast_without_n_devices():
    """no GPU in the build container: `bench.py --gpus 2` must stop in the parent with a clear message (not a rendezvous
    timeout or a HIP error from a rank)"""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "SDC_FORCE_DEVICE", "SDC_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


def test_pmc_records_carry_the_kernel_source_hash():
    from safediffcon_amd.build import source_hash
    h = source_hash()
    assert len(h) == 16 and h == source_hash()
    for fn in os.listdir(os.path.join(ROOT, "profiles")):
        if fn.startswith("r3_pmc_") and fn.endswith(".json"):
            with open(os.path.join(ROOT, "profiles", fn)) as fh:
                assert "kernel_source_hash" in json.load(fh), fn


def test_wg2_lane_offsets_include_the_k_row():
    """a park lane of the F(2x2,3x3) kernel adds up to 3 channel strides (W = 128) to its 32-bit offset; the host
    guard must count them (sdc_conv_describe launches nothing)."""
    import ctypes as C
    from safediffcon_amd import _lib
    from safediffcon_amd._lib import SdcConvDesc
    lib = _lib.get_lib()

    def pick(cs):
        d = SdcConvDesc()
        d.B, d.Cin0, d.Cin1, d.Cout = 4096, 64, 0, 64
        d.iD, d.iH, d.iW = d.oD, d.oH, d.oW = 1, 16, 128
        d.kD, d.kH, d.kW = 1, 3, 3
        d.sD = d.sH = d.sW = d.uD = d.uH = d.uW = 1
        d.pD, d.pH, d.pW = 0, 1, 1
        d.up_mode, d.precision = 0, 3
        d.x0s[:] = (2048, cs, 2048, 128, 1)              # channel-outermost input: the channel stride is the big one
        d.ys[:] = (64 * 2048, 2048, 2048, 128, 1)
        buf = C.create_string_buffer(128)
        assert lib.sdc_conv_describe(C.byref(d), buf, 128, None) == 0
        return buf.value.decode()
    span = 4095 * 2048 + 15 * 128 + 127
    ok = ((1 << 30) - 1 - span) // 3 // 4 * 4
    assert span + 3 * ok < (1 << 30) <= span + 3 * (ok + 4)
    assert pick(ok).startswith("conv_wg2") and pick(ok).endswith("<128>")      # (round 6: conv_wg2s_kernel<128> where it applies)
    assert not pick(ok + 4).startswith("conv_wg2")


def test_removed_precision_1_is_refused():
    import ctypes as C
    import pytest
    from safediffcon_amd import _lib
    from safediffcon_amd._lib import SdcConvDesc
    from safediffcon_amd.engine import Plan
    d = SdcConvDesc()
    d.B, d.Cin0, d.Cout = 1, 64, 64
    d.iD, d.iH, d.iW = d.oD, d.oH, d.oW = 1, 16, 128
    d.kD, d.kH, d.kW = 1, 3, 3
    d.sD = d.sH = d.sW = d.uD = d.uH = d.uW = 1
    d.pD, d.pH, d.pW = 0, 1, 1
    d.precision = 1
    d.x0s[:] = (64 * 2048, 2048, 2048, 128, 1)
    d.ys[:] = (64 * 2048, 2048, 2048, 128, 1)
    buf = C.create_string_buffer(128)
    assert _lib.get_lib().sdc_conv_describe(C.byref(d), buf, 128, None) != 0
    assert "precision" in _lib.last_error()


def test_bench_launches_itself_world8_gloo_prints_the_c5_line():
    """`python bench.py --gpus 8` (BASELINE configs[4]): eight ranks through torch.distributed.run (gloo, no GPU touched), 25
    calibration scores per rank, rank 0 relays one line tagged C5 with global batch 512 and the un-sharded Q"""
    import json
    import subprocess
    import sys
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDC_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--selftest-launcher"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dist_world_size"] == 8 and out["config"] == "C5"
    assert out["global_batch"] == 512 and out["calibration_per_rank"] == 25
    from oracle import samplers as osam
    g = torch.Generator().manual_seed(7)
    scores, weights = torch.rand(200, generator=g), torch.rand(200, generator=g) * 3
    want = osam.quantile_smoke(osam.normalize_weights(weights, smoke=True) * scores, 0.04)
    assert abs(out["conformal_Q"] - float(want)) < 1e-6
