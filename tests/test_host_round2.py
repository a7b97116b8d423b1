"""CPU-side checks added in round 2: the oracle against the production-width reference fixtures, the 32-bit-offset guard
of the conv dispatch, rank-mixed Philox keys, and the self-launching bench (gloo, world 2, no GPU)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

from oracle import nets as onets
from oracle.detweights import det_params, det_tensor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = dict(rtol=5e-5, atol=5e-6)


def test_oracle_matches_wide_reference_fixtures(golden):
    """the CPU oracle at the production widths against eps produced by the REAL reference (oracle/make_goldens.py *_wide)"""
    g = golden("burgers_unet_wide")
    P = det_params(g.spec(), int(g.scalar("weight_seed")))
    eps = onets.unet_burgers(P, det_tensor((2, 3, 16, 128), int(g.scalar("x_seed"))), g["t"], dim=64)
    torch.testing.assert_close(eps, g["eps"], **TOL)
    g = golden("tokamak_unet_wide")
    P = det_params(g.spec(), int(g.scalar("weight_seed")))
    eps = onets.unet_tokamak(P, det_tensor((2, 12, 128), int(g.scalar("x_seed"))), g["t"], dim=256)
    torch.testing.assert_close(eps, g["eps"], **TOL)
    g = golden("smoke_unet_wide")
    P = det_params(g.spec(), int(g.scalar("weight_seed")))
    eps = onets.unet_smoke(P, det_tensor((1, 32, 7, 32, 32), int(g.scalar("x_seed"))), g["t"], dim=64, dim_mults=(1, 2, 4))
    torch.testing.assert_close(eps, g["eps"], **TOL)


def test_conv_offsets_beyond_32_bits_leave_the_fast_kernels():
    """The gather kernels address with 32-bit byte offsets from a scalar base; the host must route any operand whose
    per-lane (batch + spatial) offsets reach 2^30 elements to the generic 64-bit kernel (no GPU work: sdc_conv_describe launches nothing)."""
    from safediffcon_amd import _lib
    from safediffcon_amd._lib import SdcConvDesc
    lib = _lib.get_lib()

    def pick(B):
        d = SdcConvDesc()
        d.B, d.Cin0, d.Cin1, d.Cout = B, 64, 0, 64
        d.iD, d.iH, d.iW = d.oD, d.oH, d.oW = 32, 64, 64
        d.kD = d.kH = d.kW = 3
        d.sD = d.sH = d.sW = d.pD = d.pH = d.pW = d.uD = d.uH = d.uW = 1
        d.up_mode, d.precision = 0, 2
        S = 32 * 64 * 64
        d.x0s[:] = (64 * S, S, 64 * 64, 64, 1)
        d.ys[:] = (64 * S, S, 64 * 64, 64, 1)
        buf = C.create_string_buffer(128)
        assert lib.sdc_conv_describe(C.byref(d), buf, 128, None) == 0
        return buf.value.decode()
    assert pick(64).startswith("conv_wg")             # C4 level 0 at B=64: 2^29 elements, 2^31 bytes -- inside the 32-bit form
    assert pick(128).startswith("conv_wg")            # per-lane part (batch + spatial; the channel part is a 64-bit scalar base) < 2^30
    assert pick(129).endswith(",false>")              # per-lane offsets reach 2^30 elements: generic kernel with 64-bit addressing


def test_philox_key_is_mixed_with_the_rank():
    from safediffcon_amd.diffusion import _mix_rank
    assert _mix_rank(12345, 0) == 12345
    keys = {_mix_rank(12345, r) for r in range(8)}
    assert len(keys) == 8 and all(0 <= k < 2 ** 62 for k in keys)


def test_bench_launches_itself_world2_gloo():
    """`python bench.py --gpus 2` from a plain shell: the parent starts two ranks through torch.distributed.run and relays
    rank 0's one JSON line (CPU rehearsal of the RCCL path: gloo, no GPU touched)."""
    env = dict(os.environ, SDC_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist_world_size"] == 2 and out["backend"] == "gloo"
    # the same Q as the un-sharded arithmetic
    from oracle import samplers as osam
    g = torch.Generator().manual_seed(7)
    scores, weights = torch.rand(200, generator=g), torch.rand(200, generator=g) * 3
    want = osam.quantile_smoke(osam.normalize_weights(weights, smoke=True) * scores, 0.04)
    assert abs(out["conformal_Q"] - float(want)) < 1e-6
    # a mismatching WORLD_SIZE is refused with a message, not an assertion
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
