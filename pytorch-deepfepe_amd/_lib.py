"""ctypes binding of libdfepe_hip.so (the C ABI declared in include/dfepe.h).

The library is the product: there is no CPU fallback.  `lib()` raises if the shared object is
missing, so a GPU box without the HIP extension fails loudly instead of silently running something else.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFEPE_LIB_PATH: another build of the same library (scripts/ab_step.sh times two builds alternately on one box)
LIB_PATH = os.environ.get("DFEPE_LIB_PATH") or os.path.join(_HERE, "libdfepe_hip.so")

OK = 0
W8PT_RAW_MATCHES = 1
W8PT_LOGITS = 2
W8PT_SQRT2 = 4
W8PT_NO_ROWNORM = 8
W8PT_FORCE_110 = 16
W8PT_NO_HARTLEY = 32
W8PT_ROW_PER_PAIR = 64
W8PT16_MAX_N = 128
TAIL_MAX_LAYERS = 16  # dfepe_loss_tail: layers per launch (kTailMaxLayers, csrc/loss_tail_body.h)
EPI_HOMOGENEOUS = 8
CHEIR_FP64_ONLY = 1

_P = c_void_p
_SIGNATURES = {
    "dfepe_version": (c_int, []),
    "dfepe_strerror": (c_char_p, [c_int]),
    "dfepe_save_floats": (c_int, []),
    "dfepe_selftest_rowgroup": (c_int, [_P, _P, _P, _P]),
    "dfepe_w8pt_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_uint, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "dfepe_w8pt_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_uint, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfepe_w8pt_rows_fwd": (c_int, [_P, c_int, c_int, c_uint, _P, _P, _P, _P]),
    "dfepe_floss_fwd": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P, _P, c_int, c_float, _P, _P, _P]),
    "dfepe_floss_bwd": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P, _P, c_int, c_float, _P, c_float, _P, _P, _P, _P]),
    "dfepe_pose_fwd": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfepe_pose_bwd": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, c_float, c_float, c_float, c_float, _P, _P, _P]),
    "dfepe_loss_head": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, _P, _P, _P]),
    "dfepe_loss_tail_workspace_bytes": (c_size_t, [c_int]),
    "dfepe_loss_tail": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P, _P, c_int, c_float, _P, _P, _P, c_float, c_float, c_float, c_float,
                                c_float, c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "dfepe_loss_head_pending": (c_int, [_P, _P]),
    "dfepe_loss_tail_jac": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, _P, _P, c_int, c_float, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P,
                                    _P, _P]),
    "dfepe_loss_tail_bwd": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P]),
    "dfepe_loss_stats": (c_int, [_P, c_int, c_float, _P, c_int, c_float, _P, c_int, c_float, _P, c_int, c_float, c_int, _P, _P, _P, _P]),
    "dfepe_deepf_input": (c_int, [_P, _P, c_int, c_int, c_int, c_float, c_float, _P, c_size_t, c_size_t, c_int, c_size_t, _P, _P, _P]),
    "dfepe_row_dot": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, _P, _P]),
    "dfepe_cheirality": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P]),
    "dfepe_cheirality_workspace_bytes": (c_size_t, [c_int]),
    "dfepe_cheirality_ex": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_uint, _P, _P, _P, _P, _P]),
    "dfepe_w8pt_pose_fwd": (c_int, [_P, _P, c_int, c_int, c_uint, c_float, c_float, c_float, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfepe_metrics_summary_bytes": (c_size_t, []),
    "dfepe_metrics_summary": (c_int, [_P, _P, c_size_t, _P, _P, c_int, _P, _P]),
    "dfepe_epi_metrics": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_float, c_float, _P, _P]),
    "dfepe_epi_residual_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P]),
    "dfepe_epi_residual_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P]),
    "dfepe_geo_misc": (c_int, [c_int, _P, _P, c_int, _P, _P]),
    "dfepe_inorm_lrelu_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P]),
    "dfepe_inorm_lrelu_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    "dfepe_est_points": (c_int, []),
    "dfepe_est_split": (c_int, [_P, c_long, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "dfepe_est_absmax": (c_int, [_P, c_long, _P, _P]),
    "dfepe_est_wprep_workspace_bytes": (c_size_t, [c_int]),
    "dfepe_est_wprep": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfepe_est_colsum": (c_int, [c_int, _P, _P, _P, _P, _P]),
    "dfepe_est_split_f16": (c_int, [_P, c_long, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "dfepe_est_layer_fwd": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, _P, _P, _P, c_float, c_float, _P, c_size_t, _P, c_size_t, _P, _P]),
    "dfepe_est_gemm_nt_f16": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "dfepe_est_gemm_nt": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "dfepe_est_gemm_tn": (c_int, [_P, c_size_t, c_int, _P, c_size_t, c_int, c_int, c_int, _P, _P]),
    "dfepe_est_in_bwd": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, _P, c_float, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "dfepe_est_dgamma_zero": (c_int, [_P, _P, _P, _P, c_size_t, _P, c_size_t, _P, c_int, c_int, _P, _P, c_float, c_int, c_int, c_long, _P, _P,
                                      c_size_t, _P, c_int, c_int, _P]),
    "dfepe_est_dgamma_zero_multi": (c_int, [c_int] + [_P] * 17 + [c_float, c_int, c_long, _P]),
    "dfepe_est_dgrad_in_bwd": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P, c_float, _P, c_size_t, _P, _P, _P]),
    "dfepe_est_norm_fwd": (c_int, [_P, c_int, c_int, c_long, c_int, _P, _P, c_float, c_float, _P, c_size_t, _P, c_size_t, _P, c_int, _P, _P]),
    "dfepe_est_in_bwd_n": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, _P, c_float, c_int, c_long, c_int, _P, c_size_t, _P, _P, c_int, _P, _P]),
    "dfepe_est_saved_bytes": (c_size_t, [c_int, _P, _P, c_long, c_int, c_int, c_int]),
    "dfepe_est_forward_workspace_bytes": (c_size_t, [c_int, _P, _P, c_long, c_int, c_int, c_int]),
    "dfepe_est_backward_workspace_bytes": (c_size_t, [c_int, _P, _P, c_long, c_int, c_int, c_int]),
    "dfepe_est_forward": (c_int, [_P, c_long, c_long, c_long, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, c_int, _P, _P, _P, _P]),
    "dfepe_est_backward": (c_int, [_P, c_long, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_long, c_long, _P]),
    "dfepe_est_head_fwd": (c_int, [_P, c_size_t, c_int, c_int, _P, _P, _P, _P]),
    "dfepe_est_head_dw": (c_int, [_P, c_size_t, c_int, c_int, c_int, _P, _P, _P, _P]),
    "dfepe_est_gemm_nt_f16_splitk": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, _P, _P, c_int, c_int, c_size_t, _P]),
    "dfepe_est_gemm_nt_splitk": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, _P, c_int, c_int, c_size_t, _P]),
    "dfepe_est_gemm_nt_gx": (c_int, [_P, c_size_t, _P, c_size_t, c_int, c_int, c_int, _P, c_int, c_int, c_long, c_long, _P]),
    "dfepe_est_gemm_tn_multi": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "dfepe_est_norm_fwd_r": (c_int, [_P, c_int, c_int, c_size_t, c_int, c_long, c_int, _P, _P, c_float, c_float, _P, c_size_t, _P, c_size_t, _P, _P]),
    "dfepe_est_in_bwd_r": (c_int, [_P, c_int, c_int, c_size_t, _P, _P, _P, c_size_t, _P, _P, _P, c_float, c_int, c_long, c_int, _P, c_size_t, _P, _P, _P]),
    "dfepe_est_prep_bytes": (c_size_t, [c_int, _P, _P]),
    "dfepe_est_prepare": (c_int, [c_int, _P, _P, _P, _P, _P]),
    "dfepe_nn_match_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dfepe_nn_match_two_way": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P]),
    "dfepe_gather_matches": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class DfepeError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP library; raises DfepeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DfepeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python pytorch-deepfepe_amd/build.py` or `__graft_entry__.build()`); there is no CPU fallback"
            )
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, what: str) -> None:
    if code != OK:
        msg = lib().dfepe_strerror(code).decode()
        raise DfepeError(f"{what} failed: {msg} (code {code})")
