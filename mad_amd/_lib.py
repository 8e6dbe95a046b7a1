"""ctypes binding of libmad_hip.so (the C ABI declared in include/mad_hip.h).

There is no fallback: if the shared library is missing the import of this module raises, and
every operator in ``mad_amd.ops`` raises ``RuntimeError`` on a non-zero status from the library.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAD_LIB_PATH selects another build of the library (the A/B scripts under tools/ set it): the product file is never
# swapped out from under a later run
LIB_PATH = os.environ.get("MAD_LIB_PATH") or os.path.join(_HERE, "libmad_hip.so")

MAD_BF16, MAD_F32 = 0, 1
ALGO_DDPM, ALGO_DDIM = 0, 1
PRED_EPSILON, PRED_SAMPLE, PRED_V = 0, 1, 2
OP_STEP, OP_COMPUTE_EPS, OP_INVERSE = 0, 1, 2
RES_NONE, RES_SAME, RES_UP, RES_DOWN = 0, 1, 2, 3
# bits of the `act` argument of mad_conv3x3_fused / _skip, and the factors of its prescaled form
ACT_SILU, ACT_PRESCALED, W_TILED = 1, 2, 4
PRESCALE_COEF = -1.4426950408889634      # -log2(e): (a, d) as mad_gn_finalize_scaled writes them
PRESCALE_W = -0.6931471805599453         # -ln(2): the weights


class SamplerCoef(Structure):
    """Mirror of ``mad_sampler_coef``."""
    _fields_ = [("algo", c_int32), ("pred_type", c_int32), ("op", c_int32), ("clip", c_int32),
                ("use_clipped", c_int32), ("add_variance", c_int32), ("known_noise", c_int32),
                ("sqrt_a_t", c_float), ("sqrt_b_t", c_float), ("c0", c_float), ("c1", c_float),
                ("std", c_float), ("sqrt_a_prev", c_float), ("sqrt_b_prev", c_float),
                ("dir_coef", c_float), ("clip_range", c_float), ("eps_div", c_float)]


# name -> (restype, argtypes); must list every function include/mad_hip.h declares
SIGNATURES = {
    "mad_version": (c_int, []),
    "mad_last_error": (c_char_p, []),
    "mad_build_id": (c_char_p, []),
    "mad_sampler_update": (c_int, [POINTER(SamplerCoef)] + [c_void_p] * 8 + [c_int] * 5 + [c_void_p]),
    "mad_add_noise": (c_int, [c_void_p] * 4 + [c_float, c_float] + [c_int] * 5 + [c_void_p]),
    "mad_finalize_image": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    "mad_masked_copy": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "mad_keep_mask": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "mad_keep_mask_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "mad_linear_f32": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "mad_pos_embedding": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mad_conv_in": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "mad_conv_in_stats_ok": (c_int, [c_int] * 4),
    "mad_conv_in_stats": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "mad_conv_out": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "mad_conv_out_fused": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5
                           + [c_void_p]),
    "mad_conv2d": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
                   + [c_int] * 7 + [c_void_p]),
    "mad_conv2d_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                 c_void_p] + [c_int] * 7 + [c_void_p]),
    "mad_conv2d_workspace_bytes": (ctypes.c_size_t, [c_int] * 7),
    "mad_conv2d_ws": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                              c_void_p] + [c_int] * 7 + [c_void_p, ctypes.c_size_t, c_void_p]),
    "mad_conv3x3_fused": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mad_conv3x3_fused_skip": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5 +
                               [c_void_p]),
    "mad_conv3x3_img8": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mad_conv_wgrad_workspace": (ctypes.c_size_t, [c_int] * 7),
    "mad_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t] + [c_int] * 9 + [c_void_p]),
    "mad_conv_wgrad_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t] + [c_int] * 10 +
                            [c_void_p]),
    "mad_bias_grad_workspace": (ctypes.c_size_t, [c_int]),
    "mad_bias_grad": (c_int, [c_void_p, c_int, ctypes.c_long, c_int, c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p]),
    "mad_gn_bwd_partial": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, ctypes.c_ulonglong,
                                   c_float, c_int, c_void_p] + [c_int] * 4 + [c_void_p]),
    "mad_gn_apply_dropout": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 5 +
                             [ctypes.c_ulonglong, c_float, c_void_p]),
    "mad_gn_bwd_finalize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "mad_gn_bwd_apply": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, ctypes.c_ulonglong,
                                 c_float, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 4 +
                         [c_void_p]),
    "mad_conv3x3_gn_bwd_sums": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p, c_int, ctypes.c_ulonglong, ctypes.c_float] + [c_int] * 5 + [c_void_p]),
    "mad_gn_affine_grad": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mad_pack_conv_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_long, c_int, c_void_p]),
    "mad_conv_wgrad_edge_workspace": (ctypes.c_size_t, [c_int] * 6),
    "mad_conv_wgrad_edge": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_size_t] +
                            [c_int] * 7 + [c_void_p]),
    "mad_linear_bwd_input_workspace": (ctypes.c_size_t, [c_int] * 3),
    "mad_linear_bwd_input": (c_int, [c_void_p] * 5 + [ctypes.c_size_t] + [c_int] * 5 + [c_void_p]),
    "mad_linear_bwd_weight": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "mad_dropout": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_size_t, ctypes.c_ulonglong, c_float, c_void_p]),
    "mad_mse_loss_grad": (c_int, [c_void_p] * 4 + [c_int, ctypes.c_size_t, c_void_p]),
    "mad_adamw_step": (c_int, [c_void_p] * 5 + [ctypes.c_size_t] + [c_float] * 5 + [c_int, c_float, c_float, c_void_p]),
    "mad_gn_partial": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mad_gn_finalize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "mad_gn_finalize_scaled": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "mad_gn_coef_static_max_floats": (ctypes.c_size_t, []),
    "mad_gn_coef_static": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                   c_int, c_int, c_int, c_float, c_void_p]),
    "mad_gn_apply": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mad_attention": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "mad_attention_lse": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "mad_attention_bwd": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C mad_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). mad_amd has no CPU fallback.")
    # torch ships its own HIP runtime: it must be in the process before this library is, so that the kernels
    # registered here and the streams / allocations torch hands over belong to ONE runtime (loaded the other way
    # round, launches fail with "no ROCm-capable device is detected")
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status: int, what: str) -> None:
    """Turn a non-zero C status into the Python exception convention of the reference."""
    if status != 0:
        msg = lib.mad_last_error()
        raise RuntimeError(f"{what} failed ({status}): {msg.decode() if msg else ''}")
