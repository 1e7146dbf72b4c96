"""ctypes binding of libftx.so (include/ftx.h).

There is no CPU fallback: if the library is missing or a call fails, this
module raises.  Every wrapper takes torch tensors that already live on the
GPU, checks dtype / contiguity / device on the host (a wrong shape must never
reach a kernel) and enqueues on torch's current stream."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libftx.so")

_i32, _i64, _f32, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/ftx.h one to one
SIGNATURES = {
    "ftx_version": (C.c_int, []),
    "ftx_last_error": (C.c_char_p, []),
    "ftx_stream_scratch_bytes": (_sz, []),
    "ftx_stream_scratch_attach": (C.c_int, [_vp, _vp, _sz]),
    "ftx_stream_scratch_reset": (C.c_int, [_vp]),
    "ftx_stream_scratch_release": (C.c_int, [_vp]),
    "ftx_hash": (C.c_int, [_vp, _i64, _vp, _vp]),
    "ftx_hash_kernel": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "ftx_floor_coords": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "ftx_hashtable_capacity": (_i64, [_i64]),
    "ftx_hashtable_build": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp]),
    "ftx_hashtable_query": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "ftx_count": (C.c_int, [_vp, _i64, _vp, _i64, _vp]),
    "ftx_unique_workspace_bytes": (_sz, [_i64]),
    "ftx_unique_sorted": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ftx_rotate_points": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "ftx_sorted_rank": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "ftx_downsample_coords": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "ftx_gather_coords": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "ftx_levels_workspace_bytes": (_sz, [_i64, _i32]),
    "ftx_levels_unique": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ftx_level_segments": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "ftx_level_coords": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "ftx_kernel_map_build": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _i64, _vp, _vp]),
    "ftx_kernel_map_count_workspace_bytes": (_sz, [_i64, _i32]),
    "ftx_kernel_map_count": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ftx_kernel_map_pairs": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ftx_trilinear_weights": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "ftx_voxelize_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ftx_voxelize_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ftx_devoxelize_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ftx_devoxelize_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ftx_segment_workspace_bytes": (_sz, [_i64, _i64]),
    "ftx_segment_build": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ftx_voxelize_fwd_sorted": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ftx_devoxelize_bwd_sorted": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ftx_segment_sum": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "ftx_lift_cells": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ftx_lift_gather_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ftx_lift_gather_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ftx_resample_nearest_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ftx_resample_nearest_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ftx_sample_down_workspace_bytes": (_sz, []),
    "ftx_sample_down_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ftx_sample_down_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ftx_spconv_pairs_gemm": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "ftx_spconv_pairs_gemm_scatter": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "ftx_spconv_ostat_supported": (_i32, [_i32, _i32, _i32, _i32]),
    "ftx_spconv_ostat_blocks": (_i32, [_i64]),
    "ftx_spconv_ostat": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "ftx_rows_gemm": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "ftx_spconv_reduce": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "ftx_spconv_reduce_stats_blocks": (_i32, [_i64, _i32]),
    "ftx_spconv_reduce_stats": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i32, _vp]),
    "ftx_bn_train_fwd_totals": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ftx_spconv_pairs_wgrad_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "ftx_spconv_pairs_wgrad": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "ftx_spconv_wgrad_resident_blocks": (_i32, [_i32, _i32]),
    "ftx_spconv_wgrad_table_blocks": (_i32, [_i32, _i32, _i32, _i32]),
    "ftx_bn_workspace_bytes": (_sz, [_i64, _i32]),
    "ftx_bn_train_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ftx_bn_eval_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _vp, _vp]),
    "ftx_adam_chunk_elements": (_i32, []),
    "ftx_adam_tensor_bytes": (_i32, []),
    "ftx_adam_step": (C.c_int, [_vp, _vp, _vp, _i32, C.c_double, C.c_double, _f32, _f32, _vp]),
    "ftx_add_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ftx_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i32]),
    "ftx_add_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ftx_colsum_workspace_bytes": (_sz, [_i64, _i32]),
    "ftx_colsum": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "ftx_attn_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "ftx_attn_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "ftx_attn_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]),
    "ftx_attn_fwd_tiled": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _i32, _i32, _vp]),
    "ftx_attn_bwd_tiled": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _i32, _i32, _vp]),
    "ftx_fusion_loss_workspace_bytes": (_sz, []),
    "ftx_fusion_loss": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ftx_fusion_loss_mix": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ftx_project_points": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp]),
    "ftx_eval_scatter_back": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ftx_bn_train_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def load():
    """Load libftx.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -m fusiontransformer_amd.build_ext` (or __graft_entry__.build()). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().ftx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")


def stream() -> int:
    """Raw hipStream_t of torch's current stream (the C bindings: torch.cuda.current_stream() costs ~9 us)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def ptr(t):
    return 0 if t is None else t.data_ptr()


def req(t: torch.Tensor, dtype, name: str, ndim=None):
    """Host-side operand validation before anything is launched."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA (HIP) tensor; the product path has no CPU fallback")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    return t
