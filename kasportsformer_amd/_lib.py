"""ctypes binding of libkasf_hip.so (C-ABI in include/kasf.h).

The library is built in-tree by ``kasportsformer_amd/csrc/Makefile`` (see ``__graft_entry__.build``).
There is NO fallback: if the shared object is missing or a symbol is absent the import raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KASF_LIB: another build of the same library (same-box A/B comparisons of kernel variants: tools/ab.sh); the default is the in-tree product build
LIB_PATH = os.environ.get("KASF_LIB") or os.path.join(_HERE, "libkasf_hip.so")

DTYPE_F32, DTYPE_BF16 = 0, 1
FLAG_TRAIN, FLAG_RETURN_REP, FLAG_KEEP = 1, 2, 4
EVAL_COLS = 22
ABI_VERSION = 8          # kasf_version() of the library these prototypes describe (a stale in-tree .so is refused)


class KasfConfig(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_frames", C.c_int32), ("num_heads", C.c_int32), ("neighbour_num", C.c_int32),
                ("use_adaptive_fusion", C.c_int32), ("dtype", C.c_int32)]


class KasfError(RuntimeError):
    pass


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_pi64, _pi32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol declared in include/kasf.h
SIGNATURES = {
    "kasf_last_error": (C.c_char_p, []),
    "kasf_version": (_i32, []),
    "kasf_set_single_stream": (None, [_i32]),
    "kasf_set_fused_wgrad_min_tokens": (None, [_i64]),
    "kasf_get_fused_wgrad_min_tokens": (_i64, []),
    "kasf_get_single_stream": (_i32, []),
    "kasf_set_fused_attn_bwd": (None, [_i32]),
    "kasf_get_fused_attn_bwd": (_i32, []),
    "kasf_set_deterministic": (None, [_i32]),
    "kasf_get_deterministic": (_i32, []),
    "kasf_model_create": (_i32, [C.POINTER(KasfConfig), C.POINTER(_vp)]),
    "kasf_model_create_layout_only": (_i32, [C.POINTER(KasfConfig), C.POINTER(_vp)]),
    "kasf_model_destroy": (None, [_vp]),
    "kasf_model_status": (_i32, [_vp, _pi32]),
    "kasf_param_count": (_i64, [_vp]),
    "kasf_param_live_count": (_i64, [_vp]),
    "kasf_param_entries": (_i32, [_vp]),
    "kasf_param_entry": (_i32, [_vp, _i32, C.c_char_p, _i32, _pi64, _pi32, _pi64]),
    "kasf_buffer_count": (_i64, [_vp]),
    "kasf_buffer_entries": (_i32, [_vp]),
    "kasf_buffer_entry": (_i32, [_vp, _i32, C.c_char_p, _i32, _pi64, _pi32, _pi64]),
    "kasf_backward_stages": (_i32, [_vp]),
    "kasf_stage_grad_range": (_i32, [_vp, _i32, _pi64, _pi64]),
    "kasf_packed_bytes": (_i64, [_vp]),
    "kasf_pack_weights": (_i32, [_vp, _vp, _vp, _vp]),
    "kasf_workspace_bytes": (_i64, [_vp, _i32, _i32]),
    "kasf_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "kasf_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "kasf_loss3": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _vp]),
    "kasf_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "kasf_gather_clips": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "kasf_joint_flip": (_i32, [_vp, _vp, _i64, _vp]),
    "kasf_tta_merge": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "kasf_eval_metrics": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kasf_ws_entries": (_i32, [_vp, _i32, _i32]),
    "kasf_ws_entry": (_i32, [_vp, _i32, _i32, _i32, C.c_char_p, _i32, _pi64, _pi64, _pi32]),
    "kasf_op_linear": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp]),
    "kasf_op_mlp_fwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "kasf_op_mlp_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "kasf_op_mlp_bwd_fused": (_i32, [_vp] * 17 + [_i64, _vp]),
    "kasf_op_wgrad": (_i32, [_i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "kasf_op_dgrad_lnbwd": (_i32, [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "kasf_op_attention_fwd": (_i32, [_i32, _vp, _i64, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp]),
    "kasf_op_attention_bwd": (_i32, [_i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "kasf_op_attention_fwd_heads": (_i32, [_i32, _vp, _i64, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    "kasf_op_attention_bwd_heads": (_i32, [_i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "kasf_op_attention_bwd_fused_do": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "kasf_op_cast": (_i32, [_i32, _vp, _vp, _i64, _i32, _vp]),
}
_EXTRA = {}

_lib = None


def load():
    """Loads the shared library (once) and attaches prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KasfError(
            f"{LIB_PATH} is missing: build it with `make -C kasportsformer_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "kasportsformer_amd has no CPU or eager-PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in {**SIGNATURES, **_EXTRA}.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.kasf_version() != ABI_VERSION:
        raise KasfError(f"{LIB_PATH} is version {lib.kasf_version()}, the host code expects {ABI_VERSION}: rebuild it (make -C kasportsformer_amd/csrc)")
    _lib = lib
    return lib


def check(code: int):
    if code != 0:
        msg = load().kasf_last_error()
        raise KasfError(f"libkasf_hip error {code}: {msg.decode() if msg else ''}")


def _entries(count_fn, entry_fn, handle, *pre):
    out = []
    name = C.create_string_buffer(256)
    off, ndim, shape = C.c_int64(), C.c_int32(), (C.c_int64 * 4)()
    for i in range(count_fn(handle, *pre)):
        check(entry_fn(handle, *pre, i, name, 256, C.byref(off), C.byref(ndim), shape))
        out.append((name.value.decode(), off.value, tuple(shape[k] for k in range(ndim.value))))
    return out


def param_entries(handle):
    lib = load()
    return _entries(lib.kasf_param_entries, lib.kasf_param_entry, handle)


def buffer_entries(handle):
    lib = load()
    return _entries(lib.kasf_buffer_entries, lib.kasf_buffer_entry, handle)


def ws_entries(handle, batch, flags):
    lib = load()
    out = {}
    name = C.create_string_buffer(256)
    off, numel, kind = C.c_int64(), C.c_int64(), C.c_int32()
    for i in range(lib.kasf_ws_entries(handle, batch, flags)):
        check(lib.kasf_ws_entry(handle, batch, flags, i, name, 256, C.byref(off), C.byref(numel), C.byref(kind)))
        out[name.value.decode()] = (off.value, numel.value, kind.value)
    return out
