"""-m "not gpu": the host-side weight repacking (engine.Plan.conv_weight) checked against torch CPU convolutions --
the sub-pixel forms of ConvTranspose3d (1,4,4)/(1,2,2) and Upsample(x2)+Conv2d 3x3, and the Winograd F(2,3) taps.
Only the packing arithmetic runs here (no kernel is launched); the kernels that consume these layouts are covered by
the -m gpu tests."""
import pytest
import torch
import torch.nn.functional as F

from safediffcon_amd.engine import Plan


@pytest.fixture(scope="module")
def plan():
    try:
        return Plan("cpu", precision=2)
    except Exception as e:          # the library must load even without a GPU (tests/test_abi_and_host.py checks that)
        pytest.skip(f"libsdc_hip.so not loadable here: {e}")


def _unpack(wp, taps_h, taps_w, cin, cout):
    """[K = (kh, kw, ci)][Cout] -> (Cout, Cin, kh, kw)"""
    return wp.reshape(taps_h, taps_w, cin, cout).permute(3, 2, 0, 1).contiguous()


def test_subpixel_transposed_conv_weights(plan):
    g = torch.Generator().manual_seed(0)
    cin, cout, H, W = 6, 5, 7, 9
    x = torch.randn(2, cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(cin, cout, 1, 4, 4, generator=g, dtype=torch.float64)
    ref = F.conv_transpose2d(x, w[:, :, 0], stride=2, padding=1)
    out = torch.zeros_like(ref)
    for ph in (0, 1):
        for pw in (0, 1):
            wp = plan.conv_weight(w.float(), ("convT_sub", ph, pw)).double()
            k = _unpack(wp, 2, 2, cin, cout)
            # pad (1 - ph) rows before / ph rows after (the kernel's one-sided margin), same along W
            xp = F.pad(x, (1 - pw, pw, 1 - ph, ph))
            out[:, :, ph::2, pw::2] = F.conv2d(xp, k)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)


def test_subpixel_upsample_conv_weights(plan):
    g = torch.Generator().manual_seed(1)
    cin, cout, H, W = 5, 4, 6, 8
    x = torch.randn(2, cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    out = torch.zeros_like(ref)
    for ph in (0, 1):
        for pw in (0, 1):
            wp = plan.conv_weight(w.float(), ("up2_sub", ph, pw)).double()
            k = _unpack(wp, 2, 2, cin, cout)
            xp = F.pad(x, (1 - pw, pw, 1 - ph, ph))
            out[:, :, ph::2, pw::2] = F.conv2d(xp, k)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)


def test_winograd_taps(plan):
    """[Wp | Wg]: y(2j), y(2j+1) from the four transformed taps == the direct 3-tap correlation, for every (kh, ci, co)."""
    g = torch.Generator().manual_seed(2)
    cin, cout = 3, 4
    w = torch.randn(cout, cin, 1, 3, 3, generator=g)
    buf = plan.conv_weight(w)
    nw = 9 * cin * cout
    assert buf.numel() == nw + nw // 3 * 4
    wg = buf[nw:].double().reshape(1, 3, 4, cin, cout)            # (kd, kh, xi, ci, co)
    d = torch.randn(4, generator=g, dtype=torch.float64)           # four inputs under one output pair
    for kh in range(3):
        for ci in range(cin):
            for co in range(cout):
                t = wg[0, kh, :, ci, co]
                m = torch.stack([(d[0] - d[2]) * t[0], (d[1] + d[2]) * t[1], (d[2] - d[1]) * t[2], (d[1] - d[3]) * t[3]])
                y0, y1 = m[0] + m[1] + m[2], m[1] - m[2] - m[3]
                gk = w[co, ci, 0, kh].double()
                torch.testing.assert_close(y0, (d[0:3] * gk).sum(), rtol=1e-5, atol=1e-6)
                torch.testing.assert_close(y1, (d[1:4] * gk).sum(), rtol=1e-5, atol=1e-6)
    # 1x1 weights carry no Winograd taps
    assert plan.conv_weight(torch.randn(4, 3, 1, 1, 1)).numel() == 12


def test_winograd_2d_taps():
    """precision 3: [Wp | Wg (along W) | Wg2 (over H and W)]: the 2x2 output tile from the 16 transformed products equals
    the direct 3x3 correlation of the 4x4 patch, for every (kd, ci, co)."""
    p3 = Plan("cpu", precision=3)
    g = torch.Generator().manual_seed(3)
    cin, cout, kD = 3, 4, 3
    w = torch.randn(cout, cin, kD, 3, 3, generator=g)
    buf = p3.conv_weight(w)
    nw = kD * 9 * cin * cout
    assert buf.numel() == nw + nw // 3 * 4 + nw // 9 * 16
    u = buf[nw + nw // 3 * 4:].double().reshape(kD, cin, cout, 4, 4).permute(0, 3, 4, 1, 2)      # stored (kd, ci, co, j, xi)
    d = torch.randn(4, 4, generator=g, dtype=torch.float64)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    V = Bt @ d @ Bt.t()
    for kd in range(kD):
        for ci in range(cin):
            for co in range(cout):
                Y = At @ (u[kd, :, :, ci, co] * V) @ At.t()
                gk = w[co, ci, kd].double()
                want = torch.stack([torch.stack([(d[a:a + 3, b:b + 3] * gk).sum() for b in range(2)]) for a in range(2)])
                torch.testing.assert_close(Y, want, rtol=1e-5, atol=1e-6)
    # taps that are not 3x3 over (H, W) keep the precision-2 layout; a Conv1d k3 (kD = kH = 1) appends its six F(4,3) taps:
    # [Wp | Wg (F(2,3)) | Wg43], and the four outputs of a quad from the six products equal the direct correlation
    w1 = torch.randn(4, 3, 1, 1, 3, generator=g)
    assert p3.conv_weight(w1).numel() == 36 + 48
    buf1 = Plan("cpu", precision=5).conv_weight(w1)                      # precision 5 (opt-in): as 4 + the F(4,3) taps of 1-D convs
    assert buf1.numel() == 36 + 48 + 72
    u43 = buf1[36 + 48:].double().reshape(6, 3, 4)                       # (xi, ci, co)
    Bt43 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                         [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
    At43 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    d6 = torch.randn(6, generator=g, dtype=torch.float64)
    for ci in range(3):
        for co in range(4):
            y = At43 @ (u43[:, ci, co] * (Bt43 @ d6))
            want = torch.stack([(d6[i:i + 3] * w1[co, ci, 0, 0].double()).sum() for i in range(4)])
            torch.testing.assert_close(y, want, rtol=1e-5, atol=1e-6)
    w2 = torch.randn(4, 3, 1, 2, 3, generator=g)                         # kH = 2: no F(4,3) section
    assert Plan("cpu", precision=5).conv_weight(w2).numel() == 72 + 96


def test_winograd_3d_taps():
    """precision 4: [Wp | Wg | Wg2 | Wg3 (over D, H and W)]: the 2x2x2 output block from the 64 transformed products equals
    the direct 3x3x3 correlation of the 4x4x4 patch, for every (ci, co); 3x3 taps keep the precision-3 layout."""
    p4 = Plan("cpu", precision=4)
    g = torch.Generator().manual_seed(4)
    cin, cout = 3, 2
    w = torch.randn(cout, cin, 3, 3, 3, generator=g)
    buf = p4.conv_weight(w)
    nw = 27 * cin * cout
    assert buf.numel() == nw + nw // 3 * 4 + nw // 9 * 16 + nw // 27 * 64
    u3 = buf[nw + nw // 3 * 4 + nw // 9 * 16:].double().reshape(4, cin, cout, 4, 4)     # stored (jd, ci, co, j, xi)
    d = torch.randn(4, 4, 4, generator=g, dtype=torch.float64)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    V = torch.einsum("ad,bh,cw,dhw->abc", Bt, Bt, Bt, d)
    for ci in range(cin):
        for co in range(cout):
            Y = torch.einsum("pa,qb,rc,abc->pqr", At, At, At, u3[:, ci, co] * V)
            gk = w[co, ci].double()
            want = torch.stack([torch.stack([torch.stack([(d[a:a + 3, b:b + 3, c:c + 3] * gk).sum() for c in range(2)])
                                             for b in range(2)]) for a in range(2)])
            torch.testing.assert_close(Y, want, rtol=1e-5, atol=1e-6)
    # the depth components the kernel walks: planes s[-1] - s[1], s[0] + s[1], s[1] - s[0], s[0] - s[2] of the patch
    torch.testing.assert_close(torch.einsum("ad,dhw->ahw", Bt, d), torch.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]))
    w2 = torch.randn(4, 3, 1, 3, 3, generator=g)
    assert p4.conv_weight(w2).numel() == 108 + 144 + 192
