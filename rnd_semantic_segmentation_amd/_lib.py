"""ctypes binding of libmi355seg.so (the C-ABI declared in include/mi355seg.h).

The product path has no CPU fallback: `lib()` raises if the shared library is
missing, and every wrapper in kernels.py insists on CUDA (HIP) tensors.
"""
import ctypes
import os

import torch  # noqa: F401  (first: the process must bind torch's bundled HIP runtime, not a second copy)
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355SEG_LIB") or os.path.join(_HERE, "libmi355seg.so")      # override: kernel experiments (tools/)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mi355seg.h")

P, I, F, Z, L = c_void_p, c_int, c_float, c_size_t, c_long

# name -> (restype, argtypes); mirrors include/mi355seg.h one to one (tests/test_cabi.py checks it)
SIGNATURES = {
    "mi_version": (I, []),
    "mi_last_error": (c_char_p, []),
    "mi_pack_weight_fwd": (I, [P, P, I, I, I, P]),
    "mi_pack_weight_dgrad": (I, [P, P, P, I, I, I, P]),
    "mi_pack_weights_multi": (I, [P, P, P, P, P, I, I, P]),
    "mi_conv_gemm": (I, [P, P, P] + [I] * 12 + [P, P, P, P, P, I, I, F, P]),
    "mi_conv_gemm_route": (I, [I] * 10),
    "mi_conv_gemm_pp": (I, [P, P, P] + [I] * 12 + [P, P, P, P, P, I, I, F, I, P]),
    "mi_conv_chain": (I, [P] * 6 + [L, I, I, I] + [P] * 8 + [I, I, P]),
    "mi_conv_wgrad_workspace": (Z, [I] * 6),
    "mi_conv_wgrad_route": (I, [I] * 12),
    "mi_conv_wgrad": (I, [P, P, P] + [I] * 11 + [P, I, I, I, Z, P, Z, P]),
    "mi_conv_wgrad_job_bytes": (Z, []),
    "mi_conv_wgrad_partial": (I, [P, P, P] + [I] * 11 + [P, I, I, I, Z, P, Z, P, P]),
    "mi_conv_wgrad_reduce": (I, [P, I, P]),
    "mi_aspp_pack_fwd": (I, [P, P, I, I, P]),
    "mi_aspp_pack_dgrad": (I, [P, P, I, I, P]),
    "mi_aspp_col2im": (I, [P, P, P, I, I, I, I, P, P]),
    "mi_aspp_im2col": (I, [P, P, I, I, I, I, P, P]),
    "mi_colsum_workspace": (Z, [I, I]),
    "mi_aspp_bias_grad": (I, [P, P, I, I, I, P, Z, P]),
    "mi_upsample_ac_fwd": (I, [P, P] + [I] * 6 + [P]),
    "mi_upsample_ac_bwd": (I, [P, P] + [I] * 6 + [P]),
    "mi_ce_workspace": (Z, [I, I, I]),
    "mi_softmax_ce_fwd": (I, [P, P, P, I, I, I, I, I, P, Z, P]),
    "mi_softmax_ce_bwd": (I, [P, P, P, P, I, I, I, I, I, F, P]),
    "mi_upsample_ce_workspace": (Z, [I] * 6),
    "mi_upsample_ce": (I, [P, P, P, P] + [I] * 7 + [F, P, Z, P]),
    "mi_upsample_ce_ex": (I, [P, P, P, P] + [I] * 7 + [F, I, P, Z, P]),
    "mi_upsample_softmax": (I, [P, P, P] + [I] * 6 + [P]),
    "mi_stem_pool_fwd": (I, [P, P, P, P, P] + [I] * 6 + [P]),
    "mi_stem_pool_bwd": (I, [P, P, P, P] + [I] * 6 + [P]),
    "mi_stem_im2col": (I, [P, P] + [I] * 6 + [P]),
    "mi_bias_grad_bf16": (I, [P, P, I, I, I, P, Z, P]),
    "mi_upsample_softce_workspace": (Z, [I] * 6),
    "mi_upsample_softce": (I, [P, F, F, P, I, I, P, P] + [I] * 6 + [F, P, Z, P]),
    "mi_adam_step": (I, [P, P, P, P, Z, F, F, F, F, I, P]),
    "mi_adam_step_clamped": (I, [P, P, P, P, Z, F, F, F, F, I, F, P]),
    "mi_adam_step_dev": (I, [P, P, P, P, Z, P, P]),
    "mi_sgd_step": (I, [P, P, P, Z, F, F, F, P]),
    "mi_sgd_step_dev": (I, [P, P, P, Z, P, P]),
    "mi_pack_weight_f32": (I, [P, P, I, I, I, P]),
    "mi_conv_f32": (I, [P, P, P] + [I] * 11 + [P, P, P, I, P]),
    "mi_stem_f32": (I, [P, P, P, P, P, I, I, I, P]),
    "mi_maxpool_f32": (I, [P, P, I, I, I, I, P]),
    "mi_comm_unique_id": (I, [P]),
    "mi_comm_init_rank": (I, [P, I, P, I]),
    "mi_comm_destroy": (I, [P]),
    "mi_allreduce_bucket": (I, [P, Z, I, I, P, P]),
    "mi_relu_mask": (I, [P, P, P, Z, I, P]),
    "mi_frozen_bn_fold": (I, [P, P, P, P, P, P, I, P]),
    "mi_structure_loss_workspace": (Z, [I, I, I]),
    "mi_structure_loss": (I, [P, P, I, I, I, P, P, F, P, Z, P]),
    "mi_bn_workspace": (Z, [L, I]),
    "mi_bn_colsum": (I, [P, P, L, I, P, P, Z, P]),
    "mi_bn_apply": (I, [P, P, P, P, P, P, P, I, L, I, P]),
    "mi_bn_colsum2": (I, [P, P, L, I, P, P, P, Z, P]),
    "mi_conv_gemm_stats_workspace": (Z, [L, I]),
    "mi_conv_gemm_stats": (I, [P, P, P] + [I] * 11 + [P, P, P, Z, P, P, P, P, P, F, F, P, P]),
    "mi_bn_finalize": (I, [P, P, P, c_double, P, P, P, P, P, F, F, P, I, P]),
    "mi_bn_bwd_colsums": (I, [P, P, P, P, P, L, I, P, P, P, Z, P]),
    "mi_bn_bwd_apply": (I, [P, P, P, P, P, P, P, F, P, P, L, I, P]),
    "mi_gconv_pack_elems": (Z, [I] * 4),
    "mi_gconv_pack_multi": (I, [P, P, P, P, I, I, P]),
    "mi_gconv_stats_elems": (Z, [I] * 4),
    "mi_gconv": (I, [P, L, P, P, L] + [I] * 16 + [P, P, I, P]),
    "mi_gconv_wgrad_workspace": (Z, [I] * 7),
    "mi_gconv_wgrad": (I, [P, L, P, L, P] + [I] * 16 + [P, Z, P, I, P]),
    "mi_gconv_wgrad_multi_table_bytes": (Z, [I]),
    "mi_gconv_wgrad_multi_workspace": (Z, [P, I]),
    "mi_gconv_wgrad_multi": (I, [P, I, P, Z, P, Z, P]),
    "mi_gbn_finalize": (I, [P, I, I, L, P, P, P, P, F, F, P, P, P, P, P]),
    "mi_gbn_fold": (I, [P, P, P, P, F, P, P, I, P]),
    "mi_gbn_apply": (I, [P, L, P, P, P, L, P, L, I, L, I, I, P]),
    "mi_gbn_apply_multi": (I, [P, L, P, P, P, L, P, L, L, I, I, I, P, P, P, P, P, P, P]),
    "mi_gcolsum_workspace": (Z, [L, I]),
    "mi_gbn_bwd_sums": (I, [P, L, I, P, L, P, L, I, P, P, L, I, P, P, I, P, Z, P]),
    "mi_gbn_bwd_apply": (I, [P, L, I, P, L, P, L, I, P, P, P, P, P, F, P, L, L, I, P]),
    "mi_gbinary": (I, [I, I, P, L, P, L, P, L, L, I, P]),
    "mi_gavgpool": (I, [P, L, P, L] + [I] * 11 + [P]),
    "mi_gmaxpool": (I, [P, L, P, L, P] + [I] * 10 + [P]),
    "mi_gce_workspace": (Z, [L]),
    "mi_gce": (I, [P, L, P, L, I, I, P, P, L, F, P, Z, P]),
    "mi_gresize": (I, [P, L, P, L, I] + [I] * 7 + [F, F, I, P]),
    "mi_gra_fwd": (I, [P, P, L, P, L, L, I, P]),
    "mi_gra_bwd": (I, [P, P, L, P, L, P, L, P, L, I, P]),
    "mi_gdwconv_stats_elems": (Z, [I] * 4),
    "mi_gdwconv": (I, [P, L, P, P, P, L] + [I] * 8 + [P, P]),
    "mi_gdwconv_dgrad": (I, [P, L, P, P, L] + [I] * 8 + [P]),
    "mi_gdwconv_wgrad_workspace": (Z, [I] * 4),
    "mi_gdwconv_wgrad": (I, [P, L, P, L, P, P] + [I] * 9 + [P, Z, P]),
    "mi_gcca_fwd": (I, [P, L, P, L, P, L, P, P, L] + [I] * 5 + [P]),
    "mi_gcca_bwd": (I, [P, L, P, L, P, L, P, P, L, P, P, L, P, L, P, L] + [I] * 5 + [P]),
    "mi_ggate": (I, [P, L, P, L, P, L, P, L, P, L, L, I, P]),
    "mi_gconv_bn_inlaunch_max_pixels": (I, []),
    "mi_gconv_bn": (I, [P, L, P, P, L] + [I] * 15 + [P, P, P, P, P, P, P, F, F, P, P]),
    "mi_gconv_f32": (I, [P, L, P, P, P, P, P, L, I, P, L] + [I] * 15 + [P]),
    "mi_gpool_f32": (I, [P, L, P, L] + [I] * 10 + [P]),
    "mi_gdwconv_f32": (I, [P, L, P, P, P, P, I, P, L] + [I] * 8 + [P]),
    "mi_gcca_f32": (I, [P, L, P, L, P, L, P, L] + [I] * 5 + [P]),
    "mi_gpoint_f32": (I, [I, P, L, P, L, P, P, I, P, L, L, I, P]),
}

_lib = None


class MiError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly: no fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiError(
                "libmi355seg.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The MI355X path has no CPU fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise MiError("%s failed (%d): %s" % (what or "libmi355seg call", rc, lib().mi_last_error().decode()))
