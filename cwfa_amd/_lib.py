"""ctypes binding of libcwfa_hip.so (C ABI: include/cwfa_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved, importing an op raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcwfa_hip.so")

CLAMP = {"NONE": 0, "ATAN": 1, "TANH": 2, "SIGMOID": 3}
ACT = {None: 0, "none": 0, "elu": 1, "prelu": 2, "gelu": 3, "relu": 4}
CHAIN_MAX = 8

c_f32p = C.c_void_p
c_i64p = C.c_void_p
c_f64p = C.c_void_p


class AffineStage(C.Structure):
    _fields_ = [("s_raw", c_f32p), ("t", c_f32p), ("s_bs", C.c_int64), ("t_bs", C.c_int64), ("clamp_kind", C.c_int),
                ("clamp", C.c_float), ("pre_scale", C.c_float), ("t_neg_div_sqrt2", C.c_int), ("perm", c_i64p),
                ("perm_axis", C.c_int), ("gin", C.c_int)]


class Chain(C.Structure):
    _fields_ = [("n_stages", C.c_int), ("stage", AffineStage * CHAIN_MAX), ("src_c", C.c_void_p), ("src_h", C.c_void_p)]


class ChainGrads(C.Structure):
    _fields_ = [("ds", c_f32p * CHAIN_MAX), ("dt", c_f32p * CHAIN_MAX), ("ds_bs", C.c_int64 * CHAIN_MAX),
                ("dt_bs", C.c_int64 * CHAIN_MAX)]


class ConvOpts(C.Structure):
    _fields_ = [("bias", c_f32p), ("act", C.c_int), ("prelu_alpha", c_f32p), ("residual", c_f32p), ("res_bs", C.c_int64),
                ("act2", C.c_int), ("in_scale", c_f32p), ("in_shift", c_f32p), ("in_affine_bs", C.c_int), ("in_add", c_f32p),
                ("in_add_bs", C.c_int64), ("upshuffle2", C.c_int), ("in_blocked8", C.c_int), ("out_blocked8", C.c_int),
                ("in_cat", c_f32p), ("in_cat_bs", C.c_int64), ("in_cat_from", C.c_int), ("in_cat_c1", C.c_int), ("out_stats", c_f64p), ("prelu_per_channel", C.c_int)]


class Couple(C.Structure):
    _fields_ = [("x", c_f32p), ("y", c_f32p), ("x_bs", C.c_int64), ("y_bs", C.c_int64), ("n", C.c_int), ("clamp_kind", C.c_int),
                ("clamp", C.c_float), ("pre_scale", C.c_float), ("rev", C.c_int), ("logdet", c_f64p), ("in_blocked8", C.c_int)]


# name -> (restype, argtypes); must list EVERY function declared in include/cwfa_hip.h (tests/test_boundary.py checks)
i, i64, f, d, p = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p
SIGNATURES = {
    "cwfa_version": (i, []),
    "cwfa_last_error": (C.c_char_p, []),
    "cwfa_set_option": (i, [C.c_char_p, i]),
    "cwfa_haar1d_fwd_f32": (i, [p, p, p, i, i, i64, i64, i64, i64, p]),
    "cwfa_haar1d_inv_f32": (i, [p, p, p, i, i, i64, i64, i64, i64, p]),
    "cwfa_haar2d_fwd_f32": (i, [p, p, i, i, i, i, i, f, p]),
    "cwfa_haar2d_inv_f32": (i, [p, p, i, i, i, i, i, f, p]),
    "cwfa_haar3d_fwd_f32": (i, [p, p, i, i, i, i, i, f, i64, p]),
    "cwfa_haar3d_inv_f32": (i, [p, p, i, i, i, i, i, f, i64, p]),
    "cwfa_gather_f32": (i, [p, p, p, i, i, i, i, i, i64, i64, p]),
    "cwfa_affine_f32": (i, [p, p, C.POINTER(AffineStage), i, i, i, i, i, i64, i64, p, p, p]),
    "cwfa_channel_affine_f32": (i, [p, p, p, p, i, p, p, i, i, i64, i64, i64, p]),
    "cwfa_chain_inv_f32": (i, [p, p, p, C.POINTER(Chain), i, i, i, i, i64, i64, i64, p, p]),
    "cwfa_chain_fwd_f32": (i, [p, p, p, C.POINTER(Chain), p, i, i, i, i, i64, i64, i64, p, p, p]),
    "cwfa_conv2d_packed_floats": (i64, [i, i, i]),
    "cwfa_conv2d_pack_f32": (i, [p, p, i, i, i, i, p]),
    "cwfa_conv2d_f32": (i, [p, p, p, i, i, i, i, i, i, i64, i64, C.POINTER(ConvOpts), p]),
    "cwfa_subnet_pack1x1_f32": (i, [p, p, p]),
    "cwfa_subnet_layer_tape_f32": (i, [p, p, p, p, p, p, p, i, i, i, i64, i64, i64, p]),
    "cwfa_subnet_layer_f32": (i, [p, p, p, p, p, p, i, i, i, i64, i64, p]),
    "cwfa_conv3d_1k1_f32": (i, [p, p, p, p, p, p, p, i, i, i, i, i, p]),
    "cwfa_conv3d_1k1_split_f32": (i, [p, p, p, p, p, p, p, i, i, i, i, i, p]),
    "cwfa_channel_stats_f32": (i, [p, p, i, i, i64, i64, p]),
    "cwfa_bn_fold_f32": (i, [p, d, p, p, p, p, f, p, i, p, p, i, p]),
    "cwfa_maxpool_f32": (i, [p, p, p, p, p, i, i, i, i, i, i, p]),
    "cwfa_sample_stats_f32": (i, [p, p, i, i64, p]),
    "cwfa_layernorm_apply_f32": (i, [p, p, p, p, f, p, i, i64, p]),
    "cwfa_attention_combine_f32": (i, [p, p, p, p, p, p, p, p, i, i, i64, p]),
    "cwfa_scale_channels_f32": (i, [p, p, p, i, i, i64, p]),
    "cwfa_axpby_f32": (i, [p, p, f, f, p, i64, p]),
    "cwfa_bn_running_update_f32": (i, [p, C.c_double, f, p, p, p, i, p]),
    "cwfa_bn_finish_f32": (i, [p, d, p, p, p, f, i, p, p, f, p, p, f, f, i, p, p, i, i, p]),
    "cwfa_chain_bwd_f32": (i, [p, p, C.POINTER(Chain), C.POINTER(ChainGrads), p, p, i, i, i, i, i64, i64, i64, f, f, i, p, p]),
    "cwfa_affine_bwd_f32": (i, [p, p, C.POINTER(AffineStage), i, i, i, i, i, i64, i64, p, p, p, p, p]),
    "cwfa_chain_inv_bwd_f32": (i, [p, p, C.POINTER(Chain), C.POINTER(ChainGrads), i, i, i, i, i64, i64, f, i, i, p, p, p, p]),
    "cwfa_conv2d_wgrad_workspace_bytes": (i64, [i, i, i, i, i, i]),
    "cwfa_conv2d_wgrad_f32": (i, [p, p, p, p, p, i, i, i, i, i, i, i64, i64, f, p]),
    "cwfa_elu_bwd_f32": (i, [p, p, p, p, i, i64, i64, i64, i64, i64, p]),
    "cwfa_conv3d_hidden_fwd_f32": (i, [p, p, p, p, i, i, i, i, i, p]),
    "cwfa_conv3d_hidden_bwd_f32": (i, [p, p, p, p, p, p, i, i, i, i, i, p]),
    "cwfa_conv3d_input_bwd_f32": (i, [p, p, p, i, i, i, i, i, p]),
    "cwfa_conv3d_wgrad_workspace_bytes": (i64, [i, i, i, i]),
    "cwfa_conv3d_wgrad_f32": (i, [p, p, p, p, p, i, i, i, i, i, i, i, f, p]),
    "cwfa_prelu_bwd_f32": (i, [p, p, p, p, p, i, i64, i64, i64, i64, p]),
    "cwfa_plane_affine_f32": (i, [p, p, p, i, p, p, i, i, i64, i64, i64, i64, p]),
    "cwfa_bn_bwd_stats_f32": (i, [p, p, p, p, i, i, i64, i64, i64, p]),
    "cwfa_bn_act_bwd_f32": (i, [p, p, p, i, p, p, p, p, p, i, i, i64, i64, i64, i64, p]),
    "cwfa_maxpool2_bwd_f32": (i, [p, p, p, p, i, i, i, i, p]),
    "cwfa_gelu_f32": (i, [p, p, p, i64, i, p]),
    "cwfa_layernorm_bwd_f32": (i, [p, p, p, p, p, p, p, p, p, i, i64, p]),
    "cwfa_attention_bwd_f32": (i, [p, p, p, p, p, p, p, p, p, i, i, i64, p]),
    "cwfa_split_workspace_bytes": (i64, [i, i, i64]),
    "cwfa_split_input_f32": (i, [p, p, i, i, i64, i64, p, p, i64, p, i64, p]),
    "cwfa_conv_split_packed_bytes": (i64, [i, i, i]),
    "cwfa_conv_split_pack_f32": (i, [p, p, i, i, i, i, p]),
    "cwfa_conv_split_f32": (i, [p, p, p, i, i, i, i, i, i, i64, C.POINTER(ConvOpts), p]),
    "cwfa_conv3x3_split_packed_bytes": (i64, [i, i]),
    "cwfa_conv3x3_split_pack_f32": (i, [p, p, i, i, p]),
    "cwfa_conv3x3_split_f32": (i, [p, p, p, i, i, i, i, i, i64, i64, C.POINTER(ConvOpts), p]),
    "cwfa_conv7x7_split_packed_bytes": (i64, [i, i]),
    "cwfa_conv7x7_split_pack_f32": (i, [p, p, i, i, p]),
    "cwfa_conv7x7_split_f32": (i, [p, p, p, i, i, i, i, i, i64, i64, C.POINTER(ConvOpts), p]),
    "cwfa_couple_rows": (i, [i, p]),
    "cwfa_conv3x3_split_couple_f32": (i, [p, p, p, i, i, i, i, i64, C.POINTER(Couple), p]),
    "cwfa_subnet_layer_split_packed_bytes": (i64, []),
    "cwfa_subnet_layer_split_pack_f32": (i, [p, p, p, p]),
    "cwfa_subnet_layer_split_f32": (i, [p, p, p, p, p, i, i, i, i64, i64, i, p]),
    "cwfa_subnet_layer_split_tape_f32": (i, [p, p, p, p, p, p, i, i, i, i64, i64, i64, p]),
    "cwfa_subnet_layer_first_packed_bytes": (i64, []),
    "cwfa_subnet_layer_first_pack_f32": (i, [p, p, p, i, p, p]),
    "cwfa_subnet_layer_first_f32": (i, [p, p, p, p, p, p, i, i, i, i, i64, i64, i64, i, p]),
    "cwfa_extract_views_f32": (i, [p, p, p, i, i, i, i, i, i, f, f, i64, p]),
}
del i, i64, f, d, p

_lib = None


class CwfaHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CwfaHipError(
                f"{LIB_PATH} not found: build the HIP extension first (python -m cwfa_amd.build, or "
                f"__graft_entry__.build()).  cwfa_amd has no CPU / PyTorch fallback.")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be in the process BEFORE our library so that both bind
        # to the SAME HIP runtime (we launch on torch's streams).  Loaded the other way round the system runtime wins
        # and torch then finds no device.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the symbol is missing -> loud
            fn.restype = res
            fn.argtypes = args
        if h.cwfa_version() < 100:
            raise CwfaHipError("libcwfa_hip.so is older than this Python layer")
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().cwfa_last_error()
        raise CwfaHipError(f"{what or 'libcwfa_hip'} failed (code {rc}): {msg.decode() if msg else ''}")
