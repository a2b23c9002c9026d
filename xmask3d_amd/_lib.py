"""ctypes binding of libxm3d_hip.so (C ABI declared in include/xm3d.h).

There is NO fallback: if the shared library is missing or a call fails, the
product path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C xmask3d_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XM3D_LIB") or os.path.join(_HERE, "libxm3d_hip.so")  # XM3D_LIB: diagnostic builds (tools/conv_ablate.sh)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "xm3d.h")

c_i32, c_i64, c_sz, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p


class Xm3dError(RuntimeError):
    pass


_SIGS = {
    "xm3d_version": (ctypes.c_int, []),
    "xm3d_last_error": (ctypes.c_char_p, []),
    "xm3d_device_info": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_char_p]),
    "xm3d_check_flag": (ctypes.c_int, []),
    "xm3d_voxelize_ws_bytes": (ctypes.c_int, [c_i64, ctypes.POINTER(c_sz)]),
    "xm3d_voxelize": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "xm3d_fnv_keys": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "xm3d_stride_ws_bytes": (ctypes.c_int, [c_i64, ctypes.POINTER(c_sz)]),
    "xm3d_coords_stride": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, ctypes.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "xm3d_coords_order": (ctypes.c_int, [c_vp, c_i64, c_vp, ctypes.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "xm3d_hash_build": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "xm3d_kernel_map": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_kernel_map_invert": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i64, c_vp, c_vp]),
    "xm3d_spconv_fwd": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp]),
    "xm3d_rulebook_tiles": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_spconv_fwd_tiles": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "xm3d_spconv_fwd_split": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "xm3d_spconv_fwd_split2": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "xm3d_spconv_fwd_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "xm3d_spconv_pack_weight_split": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_spconv_split_channels": (ctypes.c_int, [c_i32]),
    "xm3d_spconv_tile_channels": (ctypes.c_int, [c_i32, c_i32]),
    "xm3d_spconv_pack_weight": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_spconv_bwd_data": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp]),
    "xm3d_spconv_bwd_weight": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "xm3d_bn_stats": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "xm3d_affine_act": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "xm3d_bn_finalize": (ctypes.c_int, [c_vp, c_i32, ctypes.c_double, c_vp, c_vp, ctypes.c_float, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_bn_bwd_reduce": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_bn_bwd_apply": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_group_norm": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, ctypes.c_float, c_i32, c_vp, c_vp, c_vp]),
    "xm3d_group_norm_nhwc": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, ctypes.c_float, c_i32, c_vp, c_vp, c_vp]),
    "xm3d_group_norm_nhwc_res": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, ctypes.c_float, c_i32, c_vp, c_vp,
                                              c_vp, c_vp]),
    "xm3d_bias_residual_stats_nhwc": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "xm3d_group_norm_nhwc_apply": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, ctypes.c_float, c_i32, c_vp, c_vp,
                                                c_vp, c_vp]),
    "xm3d_conv3x3_cout_tile": (ctypes.c_int, [c_i32]),
    "xm3d_conv3x3_packed_elems": (ctypes.c_int64, [c_i32, c_i32, c_i32]),
    "xm3d_conv3x3_pack_weight": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_conv3x3_ws_bytes": (ctypes.c_int64, [c_i64, c_i32]),
    "xm3d_conv3x3_stats_doubles": (ctypes.c_int64, [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "xm3d_gn_stats_doubles_nchw": (ctypes.c_int64, [c_i64, c_i32, c_i32, c_i32]),
    "xm3d_gn_stats_doubles_nhwc": (ctypes.c_int64, [c_i64, c_i32, c_i32, c_i32, c_i32]),
    "xm3d_conv3x3_nhwc": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, ctypes.c_float, c_i32, c_i32,
                                         c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_conv3x3_default_waves": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32]),
    "xm3d_split_bf16_nhwc": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, ctypes.c_float, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_conv3x3_nhwc_f32acc": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "xm3d_split_f16_nhwc": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, ctypes.c_float, c_i32, c_i32, ctypes.c_float, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_conv3x3_nhwc_f32acc2": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                                 ctypes.c_float, c_vp]),
    "xm3d_gemm_col_tile": (ctypes.c_int, [c_i32]),
    "xm3d_gemm_packed_elems": (ctypes.c_int64, [c_i32, c_i32, c_i32]),
    "xm3d_gemm_pack_weight": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_gemm_default_waves": (ctypes.c_int, [c_i64, c_i32, c_i32]),
    "xm3d_gemm_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    "xm3d_conv_gemm_ws_bytes": (ctypes.c_int64, [c_i64, c_i32, c_i32, c_i32]),
    "xm3d_conv_gemm_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                           c_vp, c_vp]),
    "xm3d_split_f16t_nhwc": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, ctypes.c_float, c_i32, c_i32, ctypes.c_float, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_gemm_f32": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_i32, c_vp, c_i32, ctypes.c_float, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                     c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "xm3d_gemm_f32x": (ctypes.c_int, [c_vp, ctypes.c_float, c_i64, c_i32, c_i64, c_vp, c_vp, c_i32, c_vp, c_i32, ctypes.c_float, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                      c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "xm3d_gemm_f32acc": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_i32, c_i32, c_vp, c_i32, ctypes.c_float, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                        c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "xm3d_group_norm_nhwc_stats": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_bias_residual_nhwc": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "xm3d_attn_mask_bias": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "xm3d_mask_pool_chunks": (ctypes.c_int32, [c_i64]),
    "xm3d_mask_logits_bias": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "xm3d_mask_pool": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "xm3d_point_class": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp,
                                        ctypes.c_float, ctypes.c_float, c_vp, c_vp]),
    "xm3d_attention_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                          ctypes.c_float, c_vp]),
    "xm3d_attention_fwd_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_float, c_vp]),
    "xm3d_attention_fwd_lse": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                              ctypes.c_float, c_vp, c_vp]),
    "xm3d_attention_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                          c_i32, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_linear_sum_assignment": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_compute_mapping": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, ctypes.c_double, c_vp, c_vp]),
    "xm3d_geglu": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "xm3d_layer_norm": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp]),
    "xm3d_add_layer_norm": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, ctypes.c_float, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "xm3d_layer_norm_bwd_ws_floats": (ctypes.c_int64, [c_i64, c_i32]),
    "xm3d_layer_norm_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_group_norm_bwd_ws_floats": (ctypes.c_int64, [c_i64, c_i32, c_i32]),
    "xm3d_group_norm_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, ctypes.c_float, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_column_sum_ws_floats": (ctypes.c_int64, [c_i64, c_i32]),
    "xm3d_column_sum": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "xm3d_clip_mask_blocked": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_pad_nhwc": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "xm3d_quick_gelu": (ctypes.c_int, [c_vp, c_i32, c_i64, c_vp, c_vp]),
    "xm3d_softmax_rows_f32_bf16": (ctypes.c_int, [c_vp, c_i64, c_i32, ctypes.c_float, c_vp, c_vp]),
    "xm3d_nearest_index": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_nearest_index_segmented": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "xm3d_scene_votes": (ctypes.c_int, [c_vp, c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_nearest_valid_fill_workspace_bytes": (ctypes.c_int64, [c_i64]),
    "xm3d_nearest_valid_fill": (ctypes.c_int, [c_vp, c_i64, c_vp, ctypes.c_float, c_vp, c_vp, c_vp]),
    "xm3d_nearest_valid_fill_sorted_workspace_bytes": (ctypes.c_int64, [c_i64]),
    "xm3d_nearest_valid_fill_sorted": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "xm3d_msda_forward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp] + [c_i32] * 7 + [c_vp, c_vp]),
    "xm3d_msda_backward": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp] + [c_i32] * 7 + [c_vp, c_vp, c_vp, c_vp]),
    "xm3d_msda_forward_f64": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp] + [c_i32] * 7 + [c_vp, c_vp]),
    "xm3d_msda_backward_f64": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp] + [c_i32] * 7 + [c_vp, c_vp, c_vp, c_vp]),
    "xm3d_mask_point_fuse": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "xm3d_mask_owner": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_vp]),
}

_lib = None


def header_symbols():
    """Every entry point include/xm3d.h declares (used by the CPU test that checks the .so exports them)."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(xm3d_[a-z0-9_]+)\s*\(", txt)))


def lib():
    """Load (once) and return the shared library with argtypes set.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Xm3dError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `make -C xmask3d_amd/csrc` (or __graft_entry__.build()). There is no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name, None)
            if fn is None:
                continue  # declared for a later milestone; calling it raises AttributeError loudly
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().xm3d_last_error()
        raise Xm3dError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
