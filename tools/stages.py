"""Stage accounting shared by bench.py and tools/stage_report.py: for one recorded libsdc_hip.so call return the stage
it belongs to, its algorithmic FLOP and bytes (SURVEY 8d definitions: read input + write output + weights, fp32) and
the FLOP it actually issues on the matrix cores (Winograd forms issue fewer than the direct form they stand for)."""
import ctypes as C

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
PEAK_HBM_GBS = 8000.0             # HBM3E datasheet peak (6.3 TB/s is what a float4 copy reaches)


def conv_kernel_of(lib, d):
    """(kernel template instance, share of the direct-form MACs issued as MFMA work) -- asks the library itself"""
    buf = C.create_string_buffer(160)
    f = C.c_double(1.0)
    rc = lib.sdc_conv_describe(C.byref(d), buf, 160, C.byref(f))
    if rc:
        return "conv(?)", 1.0
    return buf.value.decode(), f.value


def classify(lib, fn, a):
    """-> dict(stage, kernel, flops, issued, bytes)"""
    if fn is lib.sdc_conv or fn is lib.sdc_conv_gn or fn is lib.sdc_conv_splitk:
        d = a[0]._obj
        P = d.B * d.oD * d.oH * d.oW
        cin, taps = d.Cin0 + d.Cin1, d.kD * d.kH * d.kW
        nin = d.B * cin * d.iD * d.iH * d.iW
        split = fn is lib.sdc_conv_splitk             # (x0, x1, wp, bias, y, work, bytes): no residual; + the partial copies
        by = 4.0 * (nin + P * d.Cout * (2 if (not split and a[5]) else 1) + taps * cin * d.Cout)
        kern, share = conv_kernel_of(lib, d)
        kind = f"conv {d.kD}x{d.kH}x{d.kW}" + (" (up/transposed)" if d.uH > 1 else "") + (" s2" if d.sH > 1 else "")
        if "conv_wg3" in kern:
            kind += " [Winograd F(2x2x2,3x3x3) over (D,H,W)]"
        elif "conv_wg2" in kern:
            kind += " [Winograd F(2x2,3x3) over (H,W)]"
        elif "conv_f43" in kern:
            kind += " [Winograd F(4,3) along W]"
        elif "conv_wg" in kern:
            kind += " [Winograd F(2,3) along W]"
        if fn is lib.sdc_conv_gn:
            kind += " + GroupNorm statistics in the epilogue"
        if split:
            kind += " [Cin split over workgroups + sum]"
            kern += " splitk"
        fl = 2.0 * P * d.Cout * cin * taps
        return dict(stage=kind, kernel=kern, flops=fl, issued=fl * share, bytes=by)
    if fn is lib.sdc_gn_finalize:
        return dict(stage="groupnorm stats (finalize of the conv-epilogue sums)", kernel="gn_finalize", flops=0.0, issued=0.0, bytes=0.0)
    if fn is lib.sdc_gn_stats:
        Bb, Cc, S = a[2], a[3], a[5]
        return dict(stage="groupnorm stats", kernel="gn_stats", flops=3.0 * Bb * Cc * S, issued=0.0, bytes=4.0 * Bb * Cc * S)
    if fn is lib.sdc_gn_apply:
        Bb, Cc, S = a[11], a[12], a[14]
        return dict(stage="groupnorm apply+SiLU(+res)", kernel="gn_apply", flops=8.0 * Bb * Cc * S, issued=0.0,
                    bytes=4.0 * Bb * Cc * S * (3 if a[9] else 2))
    if fn is lib.sdc_gn_fused:
        Bb, Cc, S = a[10], a[11], a[13]
        return dict(stage="groupnorm stats+apply+SiLU(+res), one launch", kernel="gn_fused", flops=11.0 * Bb * Cc * S, issued=0.0,
                    bytes=4.0 * Bb * Cc * S * (3 if a[8] else 2))
    if fn is lib.sdc_gn_pointwise_out:
        Bb, Cc, co, S = a[8], a[9], a[11], a[12]
        return dict(stage="final ResnetBlock: GroupNorm apply+SiLU+res inside the 1x1 output conv", kernel="gn_pw_out",
                    flops=Bb * S * Cc * (8.0 + 2.0 * co), issued=0.0, bytes=4.0 * Bb * S * ((2 if a[4] else 1) * Cc + co))
    if fn is lib.sdc_chan_norm:
        Bb, Cc, S = a[4], a[5], a[6]
        return dict(stage="channel LN/RMS(+res)", kernel="chan_norm", flops=8.0 * Bb * Cc * S, issued=0.0,
                    bytes=4.0 * Bb * Cc * S * (3 if a[2] else 2))
    if fn is lib.sdc_linattn:
        outer, inner, heads, nn = a[3], a[4], a[5], a[6]
        seqs = outer * inner * heads
        fl = seqs * nn * (2 * 2 * 32 * 32 + 10 * 32)
        return dict(stage="linear attention core", kernel="la_ctx/la_out", flops=fl, issued=seqs * nn * 4.0 * 32 * 32,
                    bytes=4.0 * seqs * nn * 32 * 4)
    if fn is lib.sdc_linattn_block or fn is lib.sdc_linattn_block_gn:
        gnf = fn is lib.sdc_linattn_block_gn
        outer, inner, Cc, nn = (a[13], a[14], a[15], a[16]) if gnf else (a[8], a[9], a[10], a[11])
        toks = outer * inner * nn
        if gnf:
            # + the producing ResnetBlock's GroupNorm apply + SiLU + residual add, done on pass 1's tile loads: x_raw and the
            # residual read, y written (h parks in y between the passes: not algorithmic traffic)
            mm = toks * (2.0 * Cc * 384 + 4 * 2 * 2 * 32 * 32 + 2.0 * 128 * Cc)
            return dict(stage="fused LinearAttention block + the ResnetBlock's GroupNorm apply / SiLU / residual on load",
                        kernel="la_blk_* (gn)", flops=mm + toks * 24.0 * Cc, issued=mm, bytes=12.0 * toks * Cc)
        # reference-equivalent work: qkv 1x1 (C -> 384), attention core (4 heads x two 32x32 products), out 1x1 (128 -> C),
        # two channel norms; bytes: x read once, y written once (what the fused block is priced against)
        mm = toks * (2.0 * Cc * 384 + 4 * 2 * 2 * 32 * 32 + 2.0 * 128 * Cc)
        return dict(stage="fused LinearAttention block (norm+qkv+core+out+norm+res)", kernel="la_blk_*", flops=mm + toks * 16.0 * Cc,
                    issued=mm, bytes=8.0 * toks * Cc)
    if fn is lib.sdc_tattn_block:
        outer, inner, Cc, ntok = a[7], a[8], a[9], a[10]
        toks = outer * inner * ntok
        mm = toks * (2.0 * Cc * 384 + 4 * 2 * 2 * ntok * 32 + 2.0 * 128 * Cc)
        return dict(stage=f"fused temporal-attention block, width {Cc} (norm+qkv+rotary/bias attention+out+res)", kernel="ta_block_kernel",
                    flops=mm + toks * 8.0 * Cc, issued=mm, bytes=8.0 * toks * Cc)
    if fn is lib.sdc_attn:
        outer, inner, heads, nt = a[4], a[5], a[6], a[7]
        seqs = outer * inner * heads
        st = a[11]
        fl = seqs * 4.0 * nt * nt * 32
        return dict(stage=("temporal attention core" if st != 1 else "softmax attention core"),
                    kernel=("tattn_kernel" if st != 1 else "attn_kernel"), flops=fl, issued=fl, bytes=4.0 * seqs * nt * 32 * 4)
    return dict(stage=fn.__name__, kernel=fn.__name__, flops=0.0, issued=0.0, bytes=0.0)


def time_plan(plan, lib, stream, reps=2):
    """HIP-event time of every recorded call of a plan (events on the launch stream), grouped twice:
    by stage and by kernel template instance.  -> (stages, kernels): name -> dict(launches, ms, flops, issued, bytes)"""
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.sdc_event_create(C.byref(e0))
    lib.sdc_event_create(C.byref(e1))
    stages, kernels = {}, {}
    for fn, args in plan.calls:
        w = classify(lib, fn, args)
        fn(*args, stream)                                    # warm
        lib.sdc_event_record(e0, stream)
        for _ in range(reps):
            fn(*args, stream)
        lib.sdc_event_record(e1, stream)
        ms = C.c_float()
        lib.sdc_event_elapsed_ms(e0, e1, C.byref(ms))
        for table, key in ((stages, w["stage"]), (kernels, w["kernel"])):
            g = table.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, issued=0.0, bytes=0.0))
            g["launches"] += 1
            g["ms"] += ms.value / reps
            for k in ("flops", "issued", "bytes"):
                g[k] += w[k]
    lib.sdc_event_destroy(e0)
    lib.sdc_event_destroy(e1)
    return stages, kernels


def markdown(title, stages):
    total_ms = sum(v["ms"] for v in stages.values())
    out = [f"## {title}: {total_ms:.2f} ms (sum of stage times, HIP events on the launch stream, MI355X)\n",
           "| stage | launches | ms | share | GFLOP (algorithmic) | TFLOP/s | MFMA-issued TFLOP/s | issued % of fp32-MFMA peak (157.3) | MB (algorithmic) | GB/s | % of HBM peak (8 TB/s) |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms"]):
        s = v["ms"] * 1e-3
        tf = v["flops"] / s / 1e12 if s else 0
        ti = v["issued"] / s / 1e12 if s else 0
        gbs = v["bytes"] / s / 1e9 if s else 0
        out.append(f"| {k} | {v['launches']} | {v['ms']:.3f} | {100 * v['ms'] / total_ms:.1f}% | {v['flops'] / 1e9:.1f} | {tf:.1f} | {ti:.1f} | "
                   f"{100 * ti / PEAK_F32_MFMA_TFLOPS:.1f}% | {v['bytes'] / 1e6:.0f} | {gbs:.0f} | {100 * gbs / PEAK_HBM_GBS:.1f}% |")
    return "\n".join(out)
