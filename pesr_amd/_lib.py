"""ctypes binding of libpesr_hip.so (C ABI declared in include/pesr_hip.h).

The product path has no fallback: if the shared object is missing or a symbol is absent this
module raises at import time of the first op, loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PESR_HIP_LIB") or os.path.join(_HERE, "libpesr_hip.so")   # override: experiment builds only

_P = c_void_p  # every device pointer / stream crosses the ABI as void*

# name -> (restype, argtypes); must mirror include/pesr_hip.h (tests/test_abi.py checks the symbol list)
SIGNATURES = {
    "pesr_abi_version": (c_int, []),
    "pesr_pack_conv3x3": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_pack_conv3x3_batched": (c_int, [_P, c_int, _P]),
    "pesr_pack_bias_ps": (c_int, [_P, _P, c_int, _P]),
    "pesr_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                 c_float, c_int, _P, c_size_t, _P]),
    "pesr_conv3x3_dgrad": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P,
                                   c_size_t, _P]),
    "pesr_conv3x3_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, _P,
                                   c_size_t, _P]),
}

SIGNATURES.update({
    "pesr_conv3x3_wgrad_rgb_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_wgrad_rgb": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_size_t, _P]),
    "pesr_conv3x3_wino_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pesr_pack_conv3x3_wino": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_conv3x3_wino": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int,
                                  c_int, _P, c_size_t, _P]),
    "pesr_conv3x3_wino4_score": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pesr_pack_conv3x3_wino4": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_conv3x3_wino4": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int,
                                   c_int, _P, c_size_t, _P]),
    "pesr_conv3x3_bf16_score": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_bf16x3_score": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pesr_pack_conv3x3_bf16x3": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_conv3x3_bf16x3": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int,
                                    c_int, _P]),
    "pesr_pack_conv3x3_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_conv3x3_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_int,
                                  c_int, _P]),
    "pesr_conv3x3_bf16_s2_score": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_bf16_s2": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, _P]),
    "pesr_conv3x3_bf16_s2_dgrad_score": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_bf16_s2_dgrad": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "pesr_conv3x3_wgrad_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_wgrad_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P, c_size_t, _P]),
    "pesr_conv3x3_rgb_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "pesr_conv3x3_rgb_out_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "pesr_conv3x3_rgb_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_conv3x3_rgb_in_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_meanshift_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pesr_meanshift_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "pesr_pixel_shuffle_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_pixel_shuffle_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_relu_mask": (c_int, [_P, _P, _P, _P, c_long, c_float, c_float, _P]),
    "pesr_maxpool2x2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pesr_maxpool2x2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "pesr_bn_workspace_bytes": (c_size_t, [c_long, c_int]),
    "pesr_bn_lrelu_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float,
                                  c_int, _P, c_size_t, _P]),
    "pesr_conv3x3_rgb_bn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_rgb_bn_lrelu_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float,
                                              c_float, c_int, _P, c_size_t, _P]),
    "pesr_bn_lrelu_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P, c_size_t,
                                  _P]),
    "pesr_bn_lrelu_eval_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "pesr_bn_lrelu_eval_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_size_t, _P]),
    "pesr_bn_bwd_bwd_workspace_bytes": (c_size_t, [c_long, c_int]),
    "pesr_bn_bwd_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "pesr_linear_workspace_bytes": (c_size_t, [c_int, c_int, c_long]),
    "pesr_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_long, c_int, c_float, _P, c_size_t, _P]),
    "pesr_linear_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_long, _P, c_size_t, _P]),
    "pesr_linear_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_long, c_int, _P]),
    "pesr_loss_l1_tv_fwd_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, c_size_t, _P]),
    "pesr_mse_fwd_bwd": (c_int, [_P, _P, _P, _P, c_long, c_float, _P, c_size_t, _P]),
    "pesr_crop_augment": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "pesr_psnr_y": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "pesr_adam_step": (c_int, [_P, _P, _P, _P, c_long, c_float, c_float, c_float, c_float, c_int, c_float, _P]),
    "pesr_adam_step_dev": (c_int, [_P, _P, _P, _P, c_long, _P, c_float, c_float, c_float, c_float, _P]),
    "pesr_conv3x3_bn_rows": (c_long, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "pesr_conv3x3_fwd_bn": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "pesr_conv3x3_dgrad_bn": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "pesr_conv3x3_wino4_bn": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "pesr_bn_finalize": (c_int, [_P, c_int, c_int, c_long, c_float, c_float, _P, _P, _P, _P, _P]),
    "pesr_bn_lrelu_bwd_fused": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "pesr_peer_alloc": (c_int, [c_size_t, _P, _P]),
    "pesr_peer_free": (c_int, [_P]),
    "pesr_peer_release": (c_int, [_P, c_size_t]),
    "pesr_peer_export": (c_int, [_P, _P, _P, _P]),
    "pesr_peer_open": (c_int, [_P, _P]),
    "pesr_peer_close": (c_int, [_P]),
    "pesr_peer_allreduce": (c_int, [_P, _P]),
    "pesr_peer_ctx_create": (c_int, [c_int, _P]),
    "pesr_peer_ctx_destroy": (c_int, [_P]),
    "pesr_peer_copy_probe": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "pesr_gan_loss_fwd_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P]),
    "pesr_conv_kxk_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pesr_conv_kxk_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pesr_conv_kxk_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "pesr_spectral_norm_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pesr_spectral_norm_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "pesr_spectral_norm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
})

_lib = None


class PesrHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load (once) and return the library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PesrHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m pesr_amd.build` "
                "(or __graft_entry__.build()). pesr_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE libpesr_hip.so
        # is loaded, so that the library's libamdhip64 dependency binds to that same runtime; loaded first, the library would
        # pull in /opt/rocm's copy and its kernels would then launch into a runtime that never saw torch's device context
        # (hipErrorNoDevice on the first launch).
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:  # pragma: no cover
                raise PesrHipError(f"libpesr_hip.so lacks symbol {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "PESR_EINVAL (unsupported shape/argument)", -2: "PESR_EWORKSPACE (workspace too small)"}.get(
            rc, f"hipError_t {rc}" if rc > 0 else f"code {rc}")
        raise PesrHipError(f"{what} failed: {kind}")
