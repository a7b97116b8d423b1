"""CPU-side checks: the C-ABI library loads and exports every symbol include/sdc.h declares,
the drop-in modules carry the reference's state_dict layout, host-side tables are bit-exact."""
import os
import re

import pytest
import torch

import safediffcon_amd as sdc
from safediffcon_amd import _lib, unet
from safediffcon_amd.diffusion import schedule_tables, BurgersGuidance, TokamakGuidance, SmokeGuidance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sdc.h")).read()
    declared = set(re.findall(r"^(?:int|size_t)\s+(sdc_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 24
    lib = _lib.get_lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sdc.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == declared
    assert lib.sdc_version() == 1


def test_error_path_no_gpu_needed():
    lib = _lib.get_lib()
    rc = lib.sdc_act(0, 0, 10, 0, 0)           # null pointers are rejected before any launch
    assert rc == -4
    assert "null" in _lib.last_error()
    with pytest.raises(_lib.SdcError):
        _lib.check(rc, "sdc_act")


@pytest.mark.parametrize("kind", ["cosine", "linear", "sigmoid"])
def test_schedule_tables_bit_exact(golden, kind):
    g = golden("schedule_" + kind)
    tabs = schedule_tables(kind, 1000)
    for k in g.keys():
        assert torch.equal(tabs[k], g[k]), k


def test_state_dict_layout_matches_reference(golden):
    cases = [("burgers_unet", lambda: sdc.Unet2D(dim=8, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1)),
             ("tokamak_unet", lambda: sdc.Unet1D(dim=8, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1)),
             ("smoke_unet", lambda: sdc.Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2, 4), channels=7)),
             ("smoke_fullspec", lambda: sdc.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=7))]
    for name, ctor in cases:
        ref = golden(name).spec()
        net = ctor()
        mine = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        assert mine == ref, name
    assert sum(p.numel() for p in net.parameters()) == 23066903      # inference_2d.py:24 prints this count


def test_cpu_tensors_are_refused():
    net = sdc.Unet1D(dim=8, channels=12, resnet_block_groups=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 12, 128), torch.zeros(1, dtype=torch.long))
    gd = sdc.GaussianDiffusionTokamak(net, seq_length=128, timesteps=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gd.sample(batch_size=1, u_init=torch.zeros(1, 3), u_final=torch.zeros(1, 2, 122))


def test_relpos_table(golden):
    g = golden("smoke_relpos")
    assert torch.equal(unet.rel_pos_bias_table(g["weight"], 32), g["bias32"])


def test_guidance_specs_match_reference_functions(golden):
    g = golden("burgers_guidance_mean")
    s = BurgersGuidance(g.scalar("Q"), g.scalar("w_score"), g.scalar("u_bound"), True)
    torch.testing.assert_close(s(g["x"]), g["grad"], rtol=1e-6, atol=1e-9)
    g = golden("burgers_guidance_amax")
    s = BurgersGuidance(g.scalar("Q"), g.scalar("w_score"), g.scalar("u_bound"), False)
    torch.testing.assert_close(s(g["x"]), g["grad"], rtol=1e-6, atol=1e-9)
    g = golden("tokamak_guidance_mixed")
    s = TokamakGuidance(g["target"], 122, g.scalar("w_obj"), g.scalar("w_safe"), g.scalar("scaler"), g.scalar("Q"), g.scalar("thr"))
    torch.testing.assert_close(s(g["x"]), g["grad"], rtol=1e-6, atol=1e-9)
    g = golden("smoke_guidance")
    s = SmokeGuidance(g.scalar("Q"), g.scalar("w_safe"), g.scalar("safe_bound"))
    torch.testing.assert_close(s(g["x"]), g["grad"], rtol=1e-6, atol=1e-9)


def test_pack_batch_plan_is_host_only_and_consistent():
    """sdc_pack_batch_plan (no GPU call): sections as sdc_pack_conv_weight_floats reports them, a block prefix without gaps, the
    largest LDS tile, and the argument errors of the single-weight entry point"""
    import ctypes as C
    from safediffcon_amd._lib import SdcPackItem
    lib = _lib.get_lib()
    shapes = [(64, 7, 1, 1, 3, 2, 0), (64, 7, 1, 1, 3, 5, 1), (24, 40, 1, 3, 3, 3, 0), (16, 24, 3, 3, 3, 4, 1), (130, 66, 1, 1, 1, 0, 0),
              (8, 12, 1, 7, 7, 0, 0), (2048, 2048, 1, 1, 3, 2, 1)]
    items = (SdcPackItem * len(shapes))()
    for it, (co, ci, kd, kh, kw, prec, flip) in zip(items, shapes):
        it.w, it.out = 0x1000, 0x2000                      # never dereferenced by the plan
        it.Cout, it.Cin, it.kD, it.kH, it.kW, it.precision, it.flip = co, ci, kd, kh, kw, prec, flip
    nb, lds = C.c_int(0), C.c_int(0)
    assert lib.sdc_pack_batch_plan(items, len(shapes), C.byref(nb), C.byref(lds)) == 0
    blocks = 0
    for it, (co, ci, kd, kh, kw, prec, _) in zip(items, shapes):
        assert sum(it.n) == lib.sdc_pack_conv_weight_floats(co, ci, kd, kh, kw, prec)
        assert it.n[0] == co * ci * kd * kh * kw
        assert it.block0 == blocks and it.grid_x == -(-co // (1 << it.co_sh)) and it.grid_y == -(-ci // (1 << it.ci_sh))
        taps = kd * kh * kw
        assert ((taps << it.ci_sh) + 1 << it.co_sh) * 4 <= lds.value <= 64 * 1024
        assert all((c * it.tap_magic) >> 24 == c // taps for c in range(0, taps << it.co_sh, 7))
        blocks += it.grid_x * it.grid_y
    assert nb.value == blocks
    items[2].precision = 1
    assert lib.sdc_pack_batch_plan(items, len(shapes), C.byref(nb), C.byref(lds)) != 0 and "item 2" in _lib.last_error()
    assert lib.sdc_pack_batch_plan(items, 0, C.byref(nb), C.byref(lds)) != 0
    assert lib.sdc_pack_batch_run(None, 1, 1, 1024, None) != 0
