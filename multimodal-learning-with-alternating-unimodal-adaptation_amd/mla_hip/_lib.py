"""ctypes binding of libmla_hip.so (C ABI declared in include/mla_hip.h).

The library is the product: there is NO fallback.  If it is missing or a call fails, an
exception is raised (`MLAHipError`).  PyTorch is only used for device memory and streams.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmla_hip.so")
LIB_PATH = os.environ.get("MLA_HIP_LIB", LIB_PATH)      # A/B measurements of two builds on one box (scripts/ab_sweep.sh)


class MLAHipError(RuntimeError):
    pass


_P, _I, _F, _Z = c_void_p, c_int, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/mla_hip.h one-to-one (tests/test_abi.py checks this)
PROTOTYPES = {
    "mla_abi_version": (_I, []),
    "mla_last_error": (c_char_p, []),
    "mla_video_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mla_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mla_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mla_conv2d_f32_cfg": (_I, [_I]),
    "mla_conv2d_fwd_partial_elems": (_Z, [_I] * 9),
    "mla_conv2d_fwd": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P]),
    "mla_conv2d_dgrad": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P, _P]),
    "mla_conv2d_wgrad_ws_bytes": (_Z, [_I] * 9),
    "mla_conv2d_wgrad": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _Z, _P]),
    "mla_conv2d_wsplit_bytes": (_Z, [_I] * 4),
    "mla_conv2d_wsplit": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mla_conv2d_wsplit_batch": (_I, [_P, _P, _P, _I, _I, _P]),
    "mla_conv2d_fwd_split": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P]),
    "mla_conv2d_dgrad_split": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P]),
    "mla_conv2d_dgrad_bn_partial_elems": (_Z, [_I, _I, _I, _I]),
    "mla_conv2d_dgrad_bn": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P, _P, _I, _P, _P]),
    "mla_conv2d_dgrad_split_bn": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P, _I, _P, _P]),
    "mla_conv2d_dgrad_classes": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "mla_conv2d_dgrad_split_classes": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P, _I, _P, _I, _I, _P]),
    "mla_conv2d_wgrad_split_ws_bytes": (_Z, [_I] * 9),
    "mla_conv2d_wgrad_split": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _Z, _P]),
    "mla_conv2d_wgrad_tr": (_I, [_I]),
    "mla_conv2d_patch": (_I, [_I]),
    "mla_conv2d_dgrad_merge": (_I, [_I]),
    "mla_conv2d_two_phase": (_I, [_I]),
    "mla_conv2d_bnfold_supported": (_I, [_I] * 9),
    "mla_conv2d_fwd_split_bnin": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P, _P, _P, _P, _P]),
    "mla_conv2d_wgrad_split_bnin": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P, _P, _P, _Z, _P]),
    "mla_conv2d_dgrad_split_bnmask": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _I, _P, _P, _P, _P]),
    "mla_conv2d_split_terms": (_I, [_I]),
    "mla_conv2d_stem_supported": (_I, [_I] * 6),
    "mla_conv2d_stem_waves": (_I, [_I]),
    "mla_conv2d_stem_fwd_partial_elems": (_Z, []),
    "mla_conv2d_stem_fwd_split": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _P, _P]),
    "mla_conv2d_stem_wgrad_split_ws_bytes": (_Z, [_I]),
    "mla_conv2d_stem_wgrad_split": (_I, [_P, _P, _P] + [_I] * 9 + [_P, _Z, _P]),
    "mla_conv2d_split_cfg": (_I, [_I]),
    "mla_bn_partial_scratch_elems": (_Z, [_I]),
    "mla_bn_stats_partial_elems": (_Z, [_I, _I]),
    "mla_bn_stats_partial": (_I, [_P, _I, _I, _P, _P, _P]),
    "mla_bn_finalize": (_I, [_P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P]),
    "mla_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mla_bn_bwd_ws_elems": (_Z, [_I, _I]),
    "mla_bn_bwd": (_I, [_P] * 11 + [_I, _I, _P]),
    "mla_bn_bwd_from_partial": (_I, [_P] * 9 + [_I, _I, _I, _P]),
    "mla_bn_relu_maxpool_fwd": (_I, [_P] * 7 + [_I, _I, _I, _I, _P]),
    "mla_bn_bwd_pooled": (_I, [_P] * 11 + [_I, _I, _I, _I, _P]),
    "mla_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "mla_maxpool3x3s2_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mla_avgpool_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "mla_avgpool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mla_head_ws_elems": (_Z, [_I, _I]),
    "mla_head_ce_fwd_bwd": (_I, [_P] * 10 + [_I, _I, _I, _F, _P]),
    "mla_ce_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "mla_head_bwd": (_I, [_P] * 6 + [_I, _I, _I, _F, _P]),
    "mla_scale_by_device_scalar": (_I, [_P, _P, _Z, _P]),
    "mla_ogm_coeff": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "mla_ogm_chunk_elems": (_I, []),
    "mla_ogm_ws_bytes": (_Z, [_I, _I]),
    "mla_ogm_modulate": (_I, [_P, _P, _P, _I, _I, _P, _I, ctypes.c_uint64, ctypes.c_uint64, _P, _Z, _P]),
    "mla_colsum": (_I, [_P, _P, _I, _I, _F, _P]),
    "mla_gs_ws_elems": (_Z, [_I, _I]),
    "mla_gs_project": (_I, [_P, _P, _P, _I, _I, _F, _P, _P]),
    "mla_sgd_step": (_I, [_P, _P, _P, _Z, _F, _F, _F, _I, _P]),
    "mla_head_logits": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "mla_eval_fuse": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P]),
    "mla_bn_invstd": (_I, [_P, _P, _I, _F, _P]),
    "mla_linear_fwd": (_I, [_P] * 6 + [_I] * 8 + [_P]),
    "mla_linear_dgrad": (_I, [_P] * 6 + [_I] * 8 + [_P]),
    "mla_linear_wgrad_ws_bytes": (_Z, [_I, _I, _I]),
    "mla_linear_wgrad": (_I, [_P, _P, _P] + [_I] * 6 + [_P, _Z, _P]),
    "mla_linear_fwd_split": (_I, [_P] * 6 + [_I] * 8 + [_P]),
    "mla_linear_dgrad_split": (_I, [_P] * 5 + [_I] * 8 + [_P]),
    "mla_linear_wgrad_split_ws_bytes": (_Z, [_I, _I, _I]),
    "mla_linear_wgrad_split": (_I, [_P, _P, _P] + [_I] * 6 + [_P, _Z, _P]),
    "mla_linear_wgrad_split_bias": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "mla_colreduce_ws_elems": (_Z, [_I, _I]),
    "mla_colsum_rows": (_I, [_P, _P, _P, _I, _I, _P]),
    "mla_layernorm_fwd": (_I, [_P] * 6 + [_I, _I, _F, _P]),
    "mla_layernorm_bwd": (_I, [_P] * 10 + [_I, _I, _P]),
    "mla_bgemm": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _Z, _Z, _F, _P]),
    "mla_softmax_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "mla_softmax_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "mla_attention_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mla_attention_bwd": (_I, [_P] * 7 + [_I, _I, _I, _I, _P]),
    "mla_tokens_assemble": (_I, [_P] * 6 + [_I, _I, _I, _I, _P]),
    "mla_tokens_assemble_bwd_ws_bytes": (_Z, [_I, _I, _I]),
    "mla_tokens_assemble_bwd": (_I, [_P] * 6 + [_I, _I, _I, _I, _P, _Z, _P]),
    "mla_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libmla_hip.so (once).  Raises MLAHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MLAHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C multimodal-learning-with-alternating-unimodal-adaptation_amd/csrc`). "
            "mla_hip has no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise MLAHipError(f"libmla_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, who: str) -> None:
    if rc != 0:
        msg = load().mla_last_error()
        raise MLAHipError(f"{who} failed (status {rc}): {msg.decode() if msg else '?'}")
