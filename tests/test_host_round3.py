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
