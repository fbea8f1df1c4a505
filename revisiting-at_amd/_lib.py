"""ctypes binding of libapgd_hip.so (C ABI declared in include/apgd_hip.h).

There is NO fallback: if the shared library is missing or a call fails, the product path
raises.  (The CPU restatement in ``oracle/`` is test infrastructure and is never imported
from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libapgd_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

F32, BF16, F16, I8, I8_BLK = 0, 1, 2, 3, 4
FLAG_NEW_BEST, FLAG_MISCLS, FLAG_HALVE = 1, 2, 4

_p, _i64, _i32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_float

# name -> (restype, argtypes); must match include/apgd_hip.h (tests/test_capi_symbols.py checks it)
PROTOTYPES = {
    "apgd_hip_version": (C.c_int, []),
    "apgd_hip_strerror": (C.c_char_p, [C.c_int]),
    "apgd_init_f32": (C.c_int, [_p, _p, _p, _p, _i64, _p]),
    "apgd_linf_step_f32": (C.c_int, [_p, _p, _p, _p, C.c_int, _p, _p, _p, _i64, _i64, _f, _f, _p]),
    "apgd_linf_step_f32_ex": (C.c_int, [_p, _p, _p, _p, C.c_int, _p, _p, _p, _i64, _i64, _f, _f,
                                        _i32, _i32, _i32, _p]),
    "apgd_linf_step_track_f32": (C.c_int, [_p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _p, _p, _i64, _i64, _f, _f, _p]),
    "apgd_l2_parts": (C.c_int, []),
    "apgd_l2_step_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f, _f, _p]),
    "apgd_loss_pred": (C.c_int, [_p, C.c_int, _i64, _p, _p, C.c_int, _p, _p, _p, _i64, _i64, _p]),
    "apgd_loss_pred_targeted": (C.c_int, [_p, C.c_int, _i64, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "apgd_state_update": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _f, _p]),
    "apgd_track_rows": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i64, _i64, _i32, _p]),
    "apgd_check_imgs_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _p]),
    "apgd_fgsm_start_f32": (C.c_int, [_p, _p, _p, _i64, _f, _f, _i32, _p]),
    "apgd_fgsm_step_f32": (C.c_int, [_p, _p, _p, _i32, _p, _i64, _f, _f, _i32, _p]),
    # include/convnext_hip.h
    "cnx_dwconv7x7_nhwc": (C.c_int, [_p, C.c_int, _p, _p, _p, _p, C.c_int, _i64, _i32, _i32, _i32, _i32, _p]),
    "cnx_dwconv7x7_win_policy": (C.c_int, [C.c_int]),
    "cnx_runtime_switch": (C.c_int, [_i32, _i32]),
    "cnx_dwconv7x7_wgrad_ws_floats": (C.c_int64, [_i32]),
    "cnx_dwconv7x7_wgrad_nhwc": (C.c_int, [_p, C.c_int, _p, C.c_int, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_block_dgamma": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _i64, _i32, _i32, _p]),
    "cnx_block_dln_ws_floats": (C.c_int64, [_i32]),
    "cnx_block_dln": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "cnx_block_mlp_bwd_train_hpre_ln": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_block_mlp_bwd_acc_ln": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_layernorm_fwd": (C.c_int, [_p, C.c_int, _p, _p, _f, _p, C.c_int, _p, _p, _i64, _i32, _i32, _p]),
    "cnx_layernorm_bwd_ws_floats": (C.c_int64, [_i32]),
    "cnx_block_mlp_supported": (C.c_int, [_i32]),
    "cnx_mlp_packed_elems": (C.c_int64, [_i32]),
    "cnx_mlp_pack_weights": (C.c_int, [_p, _p, C.c_int, _p, _i32, _p]),
    "cnx_block_mlp_fwd": (C.c_int, [_p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, C.c_int, _p, C.c_int, _p, _i64, _i32, _p]),
    "cnx_block_mlp_bwd_supported": (C.c_int, [_i32]),
    "cnx_mlp_packed_bwd_elems": (C.c_int64, [_i32]),
    "cnx_mlp_pack_weights_bwd": (C.c_int, [_p, _p, C.c_int, _p, _i32, _p]),
    "cnx_block_mlp_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _i32, _p]),
    "cnx_block_mlp_bwd_input": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_conv3x3s2_supported": (C.c_int, [_i32, _i32, _i32, _i32]),
    "cnx_conv3x3s2_packed_elems": (C.c_int64, [_i32, _i32]),
    "cnx_conv3x3s2_pack": (C.c_int, [_p, C.c_int, _p, _i32, _i32, _p]),
    "cnx_conv3x3s2_fwd": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "cnx_conv3x3s2_dgrad": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "cnx_conv3x3s2_wgrad_supported": (C.c_int, [_i32, _i32, _i32, _i32]),
    "cnx_conv3x3s2_wgrad_ws_floats": (C.c_int64, [_i32, _i32]),
    "cnx_conv3x3s2_wgrad": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "cnx_block_mlp_hpre_supported": (C.c_int, [_i32]),
    "cnx_block_mlp_hpre_elems": (C.c_int64, [_i64, _i32]),
    "cnx_block_mlp_fwd_hpre": (C.c_int, [_p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, C.c_int, _p, C.c_int, _p, _i64, _i32, _p]),
    "cnx_block_mlp_bwd_input_hpre": (C.c_int, [_p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_block_mlp_fwd_train": (C.c_int, [_p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, C.c_int, _p, C.c_int, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_block_mlp_bwd_train_hpre": (C.c_int, [_p, C.c_int, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_block_mlp_bwd_acc": (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_attention_supported": (C.c_int, [_i32, _i32]),
    "cnx_stem_conv_supported": (C.c_int, [_i32]),
    "cnx_stem_conv_packed_bytes": (C.c_int64, [_i32]),
    "cnx_stem_conv_pack": (C.c_int, [_p, C.c_int, _p, _i32, _p]),
    "cnx_stem_conv_fwd": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_stem_conv_ln_gelu_fwd": (C.c_int, [_p, _p, _p, _p, _p, C.c_float, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_stem_conv_wgrad_ws_floats": (C.c_int64, [_i32]),
    "cnx_stem_conv_wgrad": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_stem_conv_dgrad": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_stem_conv_dgrad_sign": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_stem_conv_dgrad_sign_blk": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_colsum_ws_floats": (C.c_int64, [_i32]),
    "cnx_sum_parts_bf16": (C.c_int, [_p, _p, _i64, _i64, _p]),
    "cnx_scale_residual": (C.c_int, [_p, C.c_int, _p, _p, _p, C.c_int, _i64, _i32, _p]),
    "cnx_scale_residual_bwd": (C.c_int, [_p, C.c_int, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_gelu_bwd_colsum": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _p]),
    "cnx_gelu_fwd": (C.c_int, [_p, _p, _i64, _p]),
    "cnx_attention_bwd_supported": (C.c_int, [_i32, _i32]),
    "cnx_attention_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _f, _p]),
    "cnx_attention_fwd": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _f, _p]),
    "cnx_layernorm_bwd": (C.c_int, [_p, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _i64, _i32,
                                    _i32, _p]),
    "cnx_layernorm_bwd_add": (C.c_int, [_p, C.c_int, _p, C.c_int, _p, _p, _p, _p, _p, _p, C.c_int, _p, _p, _p, _i64, _i32,
                                        _i32, _p]),
    "cnx_layernorm_fwd_patch2": (C.c_int, [_p, C.c_int, _p, _p, C.c_float, _p, C.c_int, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "cnx_gemm_tn_supported": (C.c_int, [_i64, _i32, _i32]),
    "cnx_gemm_tn_ws_floats": (C.c_int64, [_i64, _i32, _i32]),
    "cnx_gemm_tn": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i32, _i32, _p]),
    "cnx_gemm_tn_ex": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _p, _p, _p, _i64, _i32, _i32, _p]),
    "cnx_gemm_tn_pair_supported": (C.c_int, [_i64, _i32, _i32]),
    "cnx_gemm_tn_pair_ws_floats": (C.c_int64, [_i64, _i32, _i32]),
    "cnx_gemm_tn_pair": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "cnx_gemm_nt_supported": (C.c_int, [_i64, _i32, _i32]),
    "cnx_gemm_nt": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, C.c_int, _i64, _i32, _i32, _i32, _p, _p, _p, _i64, C.c_int, _p, _p, _i64, _p]),
    "cnx_layernorm_bwd_patch2": (C.c_int, [_p, C.c_int, _p, C.c_int, _p, _p, _p, _p, C.c_int, _p, _p, _p, _i64, _i32, _i32, _i32,
                                           _p]),
}


class ApgdHipError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


def load(path: str | None = None):
    """Load (once) and return the ctypes handle; raises ApgdHipError if it cannot."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get("APGD_HIP_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise ApgdHipError(
                f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C revisiting-at_amd/csrc`. There is no CPU fallback for the APGD path.")
        try:
            lib = C.CDLL(p)
        except OSError as e:  # pragma: no cover
            raise ApgdHipError(f"cannot load {p}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise ApgdHipError(f"{p} does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        if path is None:
            _lib = lib
        return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().apgd_hip_strerror(code)
        raise ApgdHipError(f"{what} failed: {code} ({msg.decode() if msg else '?'})")


def dtype_code(t) -> int:
    import torch
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    if t == torch.float16:
        return F16
    if t == torch.int8:
        return I8                                   # gradient signs (apgd_linf_step_f32 / cnx_stem_conv_dgrad_sign)
    raise ApgdHipError(f"unsupported dtype {t}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
