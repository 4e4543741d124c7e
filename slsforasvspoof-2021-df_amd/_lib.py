"""ctypes binding of libslsmi.so (C ABI in include/slsmi.h).

There is no CPU or PyTorch fallback: if the library is missing, or a call
fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLSMI_LIB") or os.path.join(_HERE, "libslsmi.so")  # SLSMI_LIB: diagnostic builds

HEAD_SLS, HEAD_SAE, HEAD_NONE = 0, 1, 2
FMT_BF16, FMT_F16 = 0, 1
SAE_SPARSE, SAE_RECON, SAE_OFF = 0, 1, 2
VIEW_SAE_CODE = 0
FORMATS = {"bf16": FMT_BF16, "f16": FMT_F16}
MAX_CONV = 8

GEMM_BF16, GEMM_BF16_GELU, GEMM_F32, GEMM_F32_RESID, GEMM_F32_RESID_GELU, GEMM_F32_RELU, GEMM_F32_ATOMIC = range(7)

STATUS = {0: "OK", -1: "INVALID", -2: "HIP", -3: "MISSING", -4: "WORKSPACE", -5: "NAME", -6: "AUDIO_OPEN",
          -7: "AUDIO_FORMAT", -8: "AUDIO_DATA", -9: "AUDIO_RATE"}


class SlsmiError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("n_conv", C.c_int32),
        ("conv_dim", C.c_int32 * MAX_CONV),
        ("conv_kernel", C.c_int32 * MAX_CONV),
        ("conv_stride", C.c_int32 * MAX_CONV),
        ("embed_dim", C.c_int32),
        ("ffn_dim", C.c_int32),
        ("heads", C.c_int32),
        ("layers", C.c_int32),
        ("conv_pos", C.c_int32),
        ("conv_pos_groups", C.c_int32),
        ("ln_eps", C.c_float),
        ("head", C.c_int32),
        ("sae_dict", C.c_int32),
        ("sae_k", C.c_int32),
        ("bn_eps", C.c_float),
        ("operand_format", C.c_int32),
        ("sae_features", C.c_int32),
        ("sae_window", C.c_int32),
    ]


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); every symbol include/slsmi.h declares
_P = C.c_void_p
SIGNATURES = {
    "slsmi_version": (C.c_int, []),
    "slsmi_default_config": (None, [C.POINTER(Config), C.c_int]),
    "slsmi_create": (C.c_int, [C.c_int, C.POINTER(Config), C.POINTER(_P)]),
    "slsmi_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int, _P]),
    "slsmi_finalize": (C.c_int, [_P, _P]),
    "slsmi_reload_weights": (C.c_int, [_P, _P]),
    "slsmi_num_frames": (C.c_int, [_P, C.c_int]),
    "slsmi_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "slsmi_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, _P, _P, _P, _P]),
    "slsmi_forward_varlen": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    "slsmi_enable_training": (C.c_int, [_P]),
    "slsmi_set_low_latency": (C.c_int, [_P, C.c_int]),
    "slsmi_set_ln_fold": (C.c_int, [_P, C.c_int]),
    "slsmi_set_posconv_slab": (C.c_int, [_P, C.c_int]),
    "slsmi_set_conv_precision": (C.c_int, [_P, C.c_uint, C.c_uint]),
    "slsmi_set_range_report": (C.c_int, [_P, _P, C.c_int]),
    "slsmi_gemm_bf16_period": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int64, _P, _P, C.c_int64, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "slsmi_train_set_attention_masks": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_float]),
    "slsmi_debug_mfma_rate": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double), _P]),
    "slsmi_adam_step": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_double, _P]),
    "slsmi_debug_gemm_timing_begin": (C.c_int, [C.c_int]),
    "slsmi_debug_gemm_timing_end": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                              C.POINTER(C.c_double)]),
    "slsmi_posconv": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _P]),
    "slsmi_gemm_tn_grouped": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                      C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_int, C.c_int64, C.c_int64, _P]),
    "slsmi_gemm_tn": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                C.c_int64, C.c_int64, C.c_int64, C.c_float, _P]),
    "slsmi_gemm_bf16_ln": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P]),
    "slsmi_ln_fold_prep": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_float, C.c_int, _P]),
    "slsmi_ln_fold_rows": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P, C.c_float, C.c_int, _P]),
    "slsmi_gemm_bf16_ln_rows": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, _P]),
    "slsmi_ln_fold_consts": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "slsmi_set_grad": (C.c_int, [_P, C.c_char_p, _P]),
    "slsmi_train_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "slsmi_train_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, _P, _P, _P]),
    "slsmi_train_backward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, _P, C.c_int, _P]),
    "slsmi_train_set_masks": (C.c_int, [_P, _P]),
    "slsmi_train_set_layer_grads_event": (C.c_int, [_P, _P]),
    "slsmi_train_set_fused_adam_hyper": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "slsmi_train_set_fused_adam": (C.c_int, [_P, C.c_char_p, _P, _P, _P, C.c_int64]),
    "slsmi_train_drop_fused_copies": (C.c_int, [_P]),
    "slsmi_train_backbone_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, _P]),
    "slsmi_train_backbone_backward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, _P]),
    "slsmi_train_sls_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, _P, _P, _P]),
    "slsmi_train_sls_backward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t, _P, _P, _P]),
    "slsmi_last_error": (C.c_char_p, [_P]),
    "slsmi_destroy": (C.c_int, [_P]),
    "slsmi_gemm_bf16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "slsmi_gemm_bf16_static": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "slsmi_time_gemm_bf16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    "slsmi_conv0_ln_gelu": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_float, _P, C.c_int64, C.c_int,
                                      _P]),
    "slsmi_ln512_bf16": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_float, C.c_int, C.c_int, _P]),
    "slsmi_ln512_f32": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_float, C.c_int, C.c_int, _P]),
    "slsmi_ln512_split3": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_float, C.c_int, _P]),
    "slsmi_ln1024_f32_bf16": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_float, C.c_int, _P]),
    "slsmi_attention": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "slsmi_attention_varlen": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P]),
    "slsmi_audio_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "slsmi_audio_read": (C.c_int, [C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "slsmi_audio_resample": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "slsmi_audio_load_batch": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int64, C.c_int, _P, C.c_int,
                                         C.POINTER(C.c_int)]),
    "slsmi_workspace_view": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "slsmi_gemm_resid_rowdot": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, _P, _P, _P]),
    "slsmi_sls_weights": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "slsmi_sls_sum_pool": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float,
                                     C.c_int, _P]),
    "slsmi_window_topk": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "slsmi_window_topk_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "slsmi_topk_rows": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
}


def load() -> C.CDLL:
    """Load libslsmi.so from the package directory (built by csrc/Makefile or
    __graft_entry__.build()).  Raises SlsmiError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlsmiError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc --offload-arch=gfx950); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, ctx=None, what: str = "") -> None:
    if rc == 0:
        return
    msg = ""
    if ctx is not None:
        m = load().slsmi_last_error(ctx)
        msg = m.decode() if m else ""
    raise SlsmiError(f"{what or 'libslsmi call'} failed: {STATUS.get(rc, rc)} {msg}".strip())


class TrainMasks(C.Structure):
    """slsmi_train_masks of include/slsmi.h."""
    _fields_ = [("input", C.c_void_p), ("encoder", C.c_void_p), ("attn_out", C.c_void_p), ("ffn_act", C.c_void_p),
                ("ffn_out", C.c_void_p), ("layer_keep", C.c_void_p)]
