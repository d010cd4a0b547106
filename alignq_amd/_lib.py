"""ctypes binding of libalignq_hip.so (include/alignq.h).  The product path has NO fallback: if the HIP
library is missing or a tensor is not a CUDA fp32 tensor, the call raises."""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("ALIGNQ_SO") or os.path.join(_HERE, "lib", "libalignq_hip.so")   # ALIGNQ_SO: A/B builds of tools/

FORMULA_ADMM, FORMULA_CDF = 0, 1
MAX_BATCH = 128            # rows the FUSED site kernels hold on chip (ALIGNQ_MAX_BATCH)
MAX_CORR_BATCH = 1024      # rows alignq_corr_fwd / _bwd take (ALIGNQ_MAX_CORR_BATCH: blocked Gram above 128)
ABI_VERSION = 23
EINVAL, EUNSUPPORTED = -1, -2          # include/alignq.h: ALIGNQ_EINVAL, ALIGNQ_EUNSUPPORTED

_c = ctypes


class Mailboxes(threading.local):
    """One-slot hand-overs from an autograd Function's forward to the wrapper that called it (autograd hides ctx from callers): a
    convolution's batch-norm partial statistics, a quantiser's integer bins.  Per THREAD (as class attributes two models
    or threads interleaving forwards could pick up each other's tensors); a slot is written inside `forward` and emptied by the
    wrapper right behind the `apply` call, in the same thread."""
    conv3x3 = transition = gemm = site_bins = bnq_bins = None


MB = Mailboxes()


_vp, _i, _i64, _f, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t


class SiteBnArgs(ctypes.Structure):
    """include/alignq.h: alignq_site_bn_args (one site of alignq_site_partials_bn_twin; the arguments of alignq_site_partials_bn)"""
    _fields_ = [("z", _vp), ("bn_part", _vp), ("bn_gamma", _vp), ("bn_beta", _vp), ("running_mean", _vp), ("running_var", _vp),
                ("num_batches_tracked", _vp), ("momentum", _f), ("bn_eps", _f), ("ab", _vp), ("save", _vp), ("C", _i), ("HW", _i),
                ("B", _i), ("F", _i64), ("k", _i), ("act_range", _f), ("eps", _f), ("relu", _i), ("residual", _vp), ("nhwc", _i),
                ("conv_parts", _i), ("xq", _vp), ("bins_out", _vp), ("stats", _vp), ("ws", _vp)]


class SiteBwdBnArgs(ctypes.Structure):
    """include/alignq.h: alignq_site_bwd_bn_args (one site of alignq_site_bwd_apply_bn_twin; alignq_site_bwd_apply_bn's arguments)"""
    _fields_ = [("g", _vp), ("S", _vp), ("z", _vp), ("ab", _vp), ("save", _vp), ("C", _i), ("HW", _i), ("nhwc", _i), ("y_relu", _vp),
                ("y_bins", _vp), ("y_bin_bytes", _i), ("dresidual", _vp), ("stats", _vp), ("B", _i), ("F", _i64), ("act_range", _f),
                ("eps", _f), ("dx", _vp), ("dx_part", _vp)]


# name -> (restype, argtypes)   — mirrors include/alignq.h one to one
SIGNATURES = {
    "alignq_abi_version": (_i, []),
    "alignq_strerror": (_c.c_char_p, [_i]),
    "alignq_uniform_quantize": (_i, [_vp, _vp, _i64, _i, _vp]),
    "alignq_act_quant_fwd": (_i, [_vp, _vp, _vp, _i64, _i, _f, _i, _vp]),
    "alignq_act_quant_bwd": (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
    "alignq_act_quant_relu_fwd": (_i, [_vp, _vp, _i64, _i, _f, _i, _vp]),
    "alignq_act_quant_relu_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _vp]),
    "alignq_bin_bytes": (_i, [_i, _f, _i]),
    "alignq_act_quant_fwd_packed": (_i, [_vp, _vp, _vp, _i64, _i, _f, _i, _i, _vp]),
    "alignq_bins_dequant": (_i, [_vp, _vp, _i64, _i, _f, _i, _i, _vp]),
    "alignq_act_quant_bwd_packed": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _i, _i, _vp]),
    "alignq_weight_ws_bytes": (_sz, [_i64]),
    "alignq_weight_stats": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "alignq_weight_quant_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "alignq_weight_quant_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "alignq_cdf_bwd": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i64, _vp, _vp]),
    "alignq_site_ws_bytes": (_sz, [_i, _i64]),
    "alignq_site_fwd": (_i, [_vp, _i, _i64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "alignq_site_partials": (_i, [_vp, _i, _i64, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "alignq_site_partials_res": (_i, [_vp, _i, _i64, _i, _f, _f, _vp, _i, _vp, _vp, _vp, _vp]),
    "alignq_site_reduce": (_i, [_vp, _i, _i64, _vp, _vp]),
    "alignq_site_reduce_loss": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _f, _f, _vp, _vp]),
    "alignq_site_bwd_ws_bytes": (_sz, [_i]),
    "alignq_site_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _f, _f, _vp, _vp]),
    "alignq_site_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _f, _f, _vp, _vp, _vp]),
    "alignq_site_bwd_fused": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp, _vp, _i, _i64, _f, _f, _vp, _vp, _vp,
                                   _vp, _vp]),
    "alignq_corr_fwd": (_i, [_vp, _i, _i64, _f, _vp, _vp, _vp, _vp]),
    "alignq_corr_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp]),
    "alignq_bnq_ws_bytes": (_sz, [_i, _i]),
    "alignq_bnq_stats": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "alignq_bnq_stats_parts": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "alignq_bnq_fwd_parts": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i,
                                  _vp, _vp]),
    "alignq_bnq_affine": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "alignq_bnq_bwd_dx": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "alignq_site_partials_res_ab": (_i, [_vp, _vp, _i, _i, _i64, _i, _f, _f, _vp, _i, _vp, _vp, _vp, _vp]),
    "alignq_site_bwd_apply_ab": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _f, _f, _vp, _vp]),
    "alignq_site1_groups_fwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _f, _f, _vp, _i, _vp, _vp, _vp, _vp]),
    "alignq_site1_mask_bytes": (_sz, [_i, _i64, _i]),
    "alignq_site1_groups_fwd_m": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "alignq_site1_groups_reduce_loss": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _i, _f, _f, _vp, _vp]),
    "alignq_site1_groups_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _f, _f, _vp, _vp, _vp]),
    "alignq_site1_groups_prep": (_i, [_vp, _vp, _vp, _i, _vp, _f, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "alignq_site1_groups_reduce_loss_multi": (_i, [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _f, _f, _vp, _vp]),
    "alignq_site1_groups_prep_multi": (_i, [_i, _vp, _vp, _vp, _i, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "alignq_site1_cols_bytes": (_sz, [_i64, _i]),
    "alignq_site1_groups_bwd_bn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_site1_groups_bwd_bn_m": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_site_bwd_apply_ab_relu": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i64, _f, _f, _vp, _vp, _vp]),
    "alignq_bnq_mask_bytes": (_sz, [_i64, _i, _i]),
    "alignq_bnq_fwd": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_bnq_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_corr_xy_ws_bytes": (_sz, [_i, _i64]),
    "alignq_corr_xy_fwd": (_i, [_vp, _vp, _i, _i64, _f, _vp, _vp, _vp, _vp]),
    "alignq_corr_xy_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp]),
    "alignq_admm_ws_bytes": (_sz, [_i]),
    "alignq_admm_loss": (_i, [_vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_admm_update": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "alignq_admm_update_ws_bytes": (_sz, [_i, _i]),
    "alignq_admm_update_ws": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp]),
    "alignq_sgd_step": (_i, [_vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _i, _vp]),
    "alignq_sgd_grad_approx": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _f, _vp]),
    "alignq_site_reduce_loss_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp]),
    "alignq_site_reduce_loss_multi_head": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "alignq_site_prep_fused_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "alignq_conv3x3_bn_parts": (_i, [_i, _i, _i, _i]),
    "alignq_conv3x3_nhwc": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "alignq_conv_gen_bn_parts": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "alignq_conv_gen_nhwc_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "alignq_conv_gen_nhwc_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_transition_nhwc_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "alignq_transition_nhwc_bwd": (_i, [_vp] * 8 + [_i] * 6 + [_vp] * 3 + [_vp] * 14 + [_vp]),
    "alignq_conv_stem_bn_parts": (_i, [_i, _i, _i]),
    "alignq_conv_stem_nhwc_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "alignq_conv_stem_nhwc_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_conv_gen_wgrad_ws_bytes": (_sz, [_i, _i, _i]),
    "alignq_conv_gen_nhwc_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_conv3x3_wgrad_ws_bytes": (_sz, [_i]),
    "alignq_conv3x3_nhwc_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "alignq_conv3x3_nhwc_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _i, _i, _vp]),
    "alignq_conv3x3_nhwc_bwd_fill": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "alignq_qconv_supported": (_i, [_i] * 7),
    "alignq_qconv_bn_parts": (_i, [_i] * 8 + [_f]),
    "alignq_qconv_pack_weights": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp]),
    "alignq_qconv_fwd": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_f, _i, _i, _vp, _vp]),
    "alignq_qconv_stem7_bn_parts": (_i, [_i] * 4),
    "alignq_qconv_stem7_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "alignq_qconv_stem7_wgrad_ws_bytes": (_sz, [_i] * 3),
    "alignq_qconv_stem7_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "alignq_qconv_dgrad_ws_bytes": (_sz, [_i] * 7),
    "alignq_qconv_dgrad": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_vp, _vp]),
    "alignq_qconv_wgrad_ws_bytes": (_sz, [_i] * 7),
    "alignq_qconv_wgrad": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_f, _i, _vp, _vp]),
    "alignq_bn_bwd_totals": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "alignq_conv3x3_wgrad_reduce_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "alignq_head_ce_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "alignq_head_ce_bwd_site_prep": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "alignq_head_ce_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "alignq_bucket_copy_multi": (_i, [_i, _vp, _vp, _vp, _i, _vp]),
    "alignq_site_partials_bn_twin": (_i, [_vp, _vp, _vp]),
    "alignq_site_bwd_apply_bn_twin": (_i, [_vp, _vp, _vp]),
    "alignq_dp_counter_bump": (_i, [_vp, _vp]),
    "alignq_dp_flag_publish": (_i, [_vp, _vp, _vp]),
    "alignq_dp_stream_wait_ge": (_i, [_vp, _vp, _c.c_uint32]),
    "alignq_bn_ws_bytes": (_sz, [_i]),
    "alignq_bn_stats": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "alignq_bn_partial_stats": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "alignq_site_partials_bn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _i, _i64, _i, _f, _f, _i,
                                     _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "alignq_site_fill_slots": (_i, [_i, _i64]),
    "alignq_site_partials_bn_fill": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _i, _i64, _i, _f, _f, _i,
                                          _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp]),
    "alignq_site_bn_part_bytes": (_sz, [_i64, _i]),
    "alignq_bn_nhwc_ws_bytes": (_sz, [_i]),
    "alignq_bn_partial_stats_nhwc": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "alignq_site_prep_fused": (_i, [_vp, _vp, _vp, _i, _vp, _f, _vp, _i, _i64, _vp, _vp, _vp, _vp]),
    "alignq_site_bwd_apply_bn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i64, _f, _f, _vp, _vp,
                                      _vp]),
    "alignq_site_bwd_fill_slots": (_i, [_i, _i64]),
    "alignq_site_bwd_apply_bn_fill": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i64, _f, _f, _vp,
                                           _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "alignq_bn_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "alignq_weight_multi_ws_bytes": (_sz, [_i]),
    "alignq_weight_quant_fwd_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "alignq_weight_quant_bwd_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "alignq_sgd_step_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _i, _f, _f, _vp]),
    "alignq_sgd_admm_step_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _i, _f, _f,
                                        _i, _vp, _vp, _vp, _i, _i, _f, _f, _vp]),
}

_lib = None


class AlignQLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises AlignQLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise AlignQLibraryError(
            f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C alignq_amd/csrc` (hipcc --offload-arch=gfx950). alignq_amd has no CPU fallback.")
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export what alignq.h declares
        fn.restype, fn.argtypes = res, args
    if lib.alignq_abi_version() != ABI_VERSION:
        raise AlignQLibraryError("libalignq_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().alignq_strerror(rc).decode()
        raise RuntimeError(f"{what}: {msg} (code {rc})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    """HOST array of device pointers (NULL for None) for the multi-tensor entry points."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def i64_array(values):
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def i32_array(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


def dev_f32(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """Validate (CUDA, fp32) and return a contiguous view/copy.  No silent CPU path."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"alignq_amd: {name} is on {t.device}; the HIP kernels need a CUDA/ROCm tensor "
                           "(there is no CPU fallback in the product path)")
    if t.dtype != torch.float32:
        raise TypeError(f"alignq_amd: {name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def dense_f32(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """Like dev_f32 but also accepts channels-last (NHWC-strided) 4-D tensors without copying: the elementwise and
    per-site kernels only need ONE dense [B, F] row-major image of the storage (any fixed permutation of the feature
    axis leaves statistics, Gram matrices and elementwise results unchanged), so the storage is used as it lies."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"alignq_amd: {name} is on {t.device}; the HIP kernels need a CUDA/ROCm tensor "
                           "(there is no CPU fallback in the product path)")
    if t.dtype != torch.float32:
        raise TypeError(f"alignq_amd: {name} must be float32, got {t.dtype}")
    if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
        return t
    return t.contiguous()


def like_layout(g: torch.Tensor, ref: torch.Tensor, name: str = "grad") -> torch.Tensor:
    """Return g (CUDA fp32) laid out in memory exactly like ref (same strides), copying only if it is not."""
    if not g.is_cuda or g.dtype != torch.float32:
        raise TypeError(f"alignq_amd: {name} must be a CUDA float32 tensor")
    if g.stride() == ref.stride():
        return g
    out = torch.empty_like(ref)
    out.copy_(g)
    return out
