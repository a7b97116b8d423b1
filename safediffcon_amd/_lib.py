"""ctypes binding of libsdc_hip.so (the C ABI in include/sdc.h).

This is the stub a reference maintainer would add.  There is
no CPU fallback: if the shared library is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdc_hip.so")

SDC_MODEL_BURGERS, SDC_MODEL_TOKAMAK, SDC_MODEL_SMOKE = 0, 1, 2

_f32p = C.c_void_p       # device pointers travel as integers
_i32p = C.c_void_p
_stream = C.c_void_p
_i64 = C.c_int64


class SdcConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "B", "Cin0", "Cin1", "Cout", "iD", "iH", "iW", "oD", "oH", "oW", "kD", "kH", "kW", "sD", "sH", "sW",
        "pD", "pH", "pW", "uD", "uH", "uW", "up_mode", "precision")] + [
        ("x0s", C.c_int64 * 5), ("x1s", C.c_int64 * 5), ("ys", C.c_int64 * 5), ("rs", C.c_int64 * 5)]


class SdcStepDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "model", "B", "d0", "d1", "d2", "d3", "guide", "clip", "impose", "cond_idx", "pad_zero", "use_max",
        "has_wgt", "skip_draws", "ddim")] + [("_pad", C.c_int32), ("seed", C.c_uint64)]


class SdcWgradDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "B", "M", "N", "oD", "oH", "oW", "iD", "iH", "iW", "kD", "kH", "kW", "sD", "sH", "sW", "pD", "pH", "pW",
        "uD", "uH", "uW", "_pad")] + [("gs", C.c_int64 * 5), ("xs", C.c_int64 * 5)]


class SdcPackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p)] + [(n, C.c_int32) for n in (
        "Cout", "Cin", "kD", "kH", "kW", "precision", "flip", "co_sh", "ci_sh", "grid_x", "grid_y", "block0")] + [
        ("tap_magic", C.c_uint32), ("_pad", C.c_int32), ("n", C.c_int64 * 5)]


class SdcSpan(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("nwords", C.c_int64)]


class SdcKstarMlp(C.Structure):
    _fields_ = [("nlayers", C.c_int32), ("width", C.c_int32 * 7), ("act", C.c_int32 * 6), ("params", C.c_void_p), ("stride", C.c_int64)]


class SdcKstarModel(C.Structure):
    _fields_ = [("n_lstm", C.c_int32), ("n_bpw", C.c_int32), ("lstm", C.c_void_p), ("lstm_stride", C.c_int64),
                ("head", SdcKstarMlp), ("steady", SdcKstarMlp), ("bpw", SdcKstarMlp),
                ("lstm_ystd", C.c_double * 4), ("lstm_ymean", C.c_double * 4), ("nn_ystd", C.c_double * 4), ("nn_ymean", C.c_double * 4),
                ("bpw_ystd", C.c_double * 2), ("bpw_ymean", C.c_double * 2), ("scale", C.c_double), ("inputs0", C.c_double * 15),
                ("low_action", C.c_double * 9), ("high_action", C.c_double * 9), ("year_in", C.c_double)]


# name -> (restype, argtypes); every symbol include/sdc.h declares
SIGNATURES = {
    "sdc_version": (C.c_int, []),
    "sdc_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "sdc_conv": (C.c_int, [C.POINTER(SdcConvDesc), _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _stream]),
    "sdc_conv_describe": (C.c_int, [C.POINTER(SdcConvDesc), C.c_char_p, C.c_size_t, C.POINTER(C.c_double)]),
    "sdc_conv_gnparts": (C.c_int, [C.POINTER(SdcConvDesc), C.c_int]),
    "sdc_conv_gn": (C.c_int, [C.POINTER(SdcConvDesc), _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, _stream]),
    "sdc_gn_finalize": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _i64, C.c_float, _stream]),
    "sdc_gn_stats": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_int, _i64, C.c_float, _stream]),
    "sdc_gn_stats_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sdc_gn_apply": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _i32p, _i64, _i64, _i64, _f32p, _f32p,
                               C.c_int, C.c_int, C.c_int, _i64, _stream]),
    "sdc_gn_fused_ok": (C.c_int, [C.c_int, C.c_int, C.c_int, _i64]),
    "sdc_gn_fused": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i32p, _i64, _i64, _i64, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _i64,
                               C.c_float, _stream]),
    "sdc_gn_pointwise_out": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _i64,
                                       _i64, _i64, _i64, _i64, _stream]),
    "sdc_chan_norm": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, _i64, C.c_int, C.c_float, _stream]),
    "sdc_linattn": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _i64, _i64, _i64, _i64, _i64, _i64,
                              _i64, _stream]),
    "sdc_linattn_block_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, _i64]),
    "sdc_linattn_block": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _i64,
                                    _i64, _i64, _i64, C.c_int, C.c_int, C.c_float, _stream]),
    "sdc_linattn_block_gn": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                       C.c_int, C.c_int, C.c_int, _i64, _i64, _i64, _i64, C.c_int, C.c_int, C.c_float, _stream]),
    "sdc_tattn_block": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  _i64, _i64, _i64, C.c_float, _stream]),
    "sdc_attn": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _i64, _i64, _i64, _i64,
                           _i64, _i64, _i64, _i64, _stream]),
    "sdc_act": (C.c_int, [_f32p, _f32p, _i64, C.c_int, _stream]),
    "sdc_guide_reduce": (C.c_int, [C.POINTER(SdcStepDesc), _f32p, _f32p, _f32p, _i32p, _f32p, _f32p, _stream]),
    "sdc_step_update": (C.c_int, [C.POINTER(SdcStepDesc), _f32p, _f32p, _f32p, _f32p, _i32p, _i32p, _f32p, _i64,
                                  _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _stream]),
    "sdc_impose": (C.c_int, [C.POINTER(SdcStepDesc), _f32p, _f32p, _f32p, _f32p, _stream]),
    "sdc_randn": (C.c_int, [_f32p, _i64, C.c_uint64, _i32p, _stream]),
    "sdc_advance": (C.c_int, [_i32p, C.c_int, _i32p, C.c_int, _stream]),
    "sdc_advance_table": (C.c_int, [_i32p, _i32p, _i32p, _i32p, C.c_int, _stream]),
    "sdc_conformal_score": (C.c_int, [C.POINTER(SdcStepDesc), _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _stream]),
    "sdc_burgers_rollout": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_float, _stream]),
    "sdc_conv_wgrad_bytes": (C.c_size_t, [C.POINTER(SdcWgradDesc)]),
    "sdc_conv_wgrad": (C.c_int, [C.POINTER(SdcWgradDesc), _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_size_t, _stream]),
    "sdc_gn_silu_bwd_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, _i64]),
    "sdc_gn_silu_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _f32p, _f32p, _f32p, _f32p, _f32p,
                                  C.c_int, C.c_int, C.c_int, _i64, _stream]),
    "sdc_chan_norm_bwd_parts": (C.c_size_t, [C.c_int, _i64]),
    "sdc_chan_norm_bwd_bytes": (C.c_size_t, [C.c_int, C.c_int, _i64]),
    "sdc_chan_norm_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, _i64, C.c_int, C.c_float, _stream]),
    "sdc_pack_conv_weight_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sdc_pack_conv_weight": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "sdc_pack_batch_plan": (C.c_int, [C.POINTER(SdcPackItem), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdc_pack_batch_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _stream]),
    "sdc_attn_bwd_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sdc_attn_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _stream]),
    "sdc_linattn_bwd_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, _i64]),
    "sdc_linattn_bwd": (C.c_int, [_f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                  _stream]),
    "sdc_act_bwd": (C.c_int, [_f32p, _f32p, _f32p, _i64, C.c_int, _stream]),
    "sdc_sumpool2": (C.c_int, [_f32p, _f32p, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "sdc_kstar_lstm_floats": (C.c_size_t, []),
    "sdc_kstar_rollout": (C.c_int, [C.POINTER(SdcKstarModel), _f32p, _i64, _i64, _i64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _stream]),
    "sdc_smoke_rollout_workspace_bytes": (C.c_size_t, [C.c_int]),
    "sdc_smoke_rollout": (C.c_int, [_f32p, _f32p, _i64, _i64, _f32p, _i64, _f32p, _i64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _stream]),
    "sdc_conv_splitk_bytes": (C.c_size_t, [C.POINTER(SdcConvDesc)]),
    "sdc_conv_splitk": (C.c_int, [C.POINTER(SdcConvDesc), _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_size_t, _stream]),
    "sdc_linear": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _i64, _i64, _stream]),
    "sdc_linear_dgrad": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _i64, _i64, _stream]),
    "sdc_linear_wgrad": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, _i64, _i64, _stream]),
    "sdc_checksum_spans": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _stream]),
    "sdc_graph_begin": (C.c_int, [_stream]),
    "sdc_graph_end": (C.c_int, [_stream, C.POINTER(C.c_void_p)]),
    "sdc_graph_launch": (C.c_int, [C.c_void_p, _stream]),
    "sdc_graph_destroy": (C.c_int, [C.c_void_p]),
    "sdc_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "sdc_event_record": (C.c_int, [C.c_void_p, _stream]),
    "sdc_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "sdc_event_destroy": (C.c_int, [C.c_void_p]),
}

_lib = None


class SdcError(RuntimeError):
    pass


def use_library(path):
    """Kernel experiments only (tools/, bench.py's A/B runs): load another BUILD of libsdc_hip.so -- the experiments library of
    `python -m safediffcon_amd.build --experiments`, or a previous commit's -- instead of the in-tree one.  Must be called before
    the first get_lib(); nothing in the package calls it and no environment variable reaches the loader."""
    global LIB_PATH
    if _lib is not None:
        raise SdcError("use_library() after the library has been loaded")
    LIB_PATH = os.path.abspath(path)


def get_lib():
    """Load libsdc_hip.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdcError(
                f"{LIB_PATH} not found: build it with `python -m safediffcon_amd.build` "
                "(hipcc --offload-arch=gfx950). safediffcon_amd has no CPU/eager fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    get_lib().sdc_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, what=""):
    if rc != 0:
        raise SdcError(f"{what or 'libsdc_hip'} failed (code {rc}): {last_error()}")
