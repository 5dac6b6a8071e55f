"""ctypes binding of the gfx950 C-ABI library (include/medtok_vq.h).

The library is the product; there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or an entry point fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_SO = Path(__file__).resolve().parent / "csrc" / "libmedtok_vq.so"


def use_library(path) -> None:
    """Dev tools only (tools/ab_filter.py ...): bind an alternative BUILD of the same ABI before the first load().
    Still a HIP library, never a fallback; the product never reads an environment variable for this."""
    global _SO, _lib
    if _lib is not None:
        raise MedTokLibraryError("use_library() must be called before the library is first loaded")
    _SO = Path(path)
_lib = None

ABI_VERSION = 2
MAX_TOPK = 16
PATH_AUTO, PATH_F32_MFMA, PATH_F16_FILTER = 0, 1, 2


def plan_path(path: int = PATH_AUTO, filter_splits: int = 0, filter_xcd=None, filter_tail=None, search_max_splits: int = 0,
              filter_rows64=None) -> int:
    """Test hook (MEDTOK_PLAN_* in include/medtok_vq.h): a `path` argument that also forces launch-plan branches -- code-range
    splits, XCD-aware block order on/off, tail launch on/off, the exact kernel's split cap, the D <= 64 filter kernel on/off -- for
    THIS call only (the library
    keeps no plan state).  Results are bit-identical under every plan."""
    p = path & 0xF
    p |= (filter_splits & 0xFF) << 8
    if filter_xcd is not None:
        p |= (2 if filter_xcd else 1) << 16
    if filter_tail is not None:
        p |= (2 if filter_tail else 1) << 18
    p |= (search_max_splits & 0xFF) << 20
    if filter_rows64 is not None:      # False / True; "wide": the 128 x 64 wave-tile form of the D <= 64 kernel
        p |= (3 if filter_rows64 == "wide" else 2 if filter_rows64 else 1) << 28
    return p

_vp, _i64, _int, _sz, _f, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/medtok_vq.h one to one
SIGNATURES = {
    "medtok_abi_version": (_int, []),
    "medtok_last_error": (C.c_char_p, []),
    "medtok_profile_begin": (_int, []),
    "medtok_profile_begin_kinds": (_int, [C.c_uint]),
    "medtok_profile_end": (_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "medtok_soft_vq_multi_eligible": (_int, [_i64, _i64, _int, _int]),
    "medtok_soft_vq_forward_multi_workspace_bytes": (_sz, [_vp, _int, _int, _int]),
    "medtok_soft_vq_forward_multi_f32": (_int, [_vp, _int, _int, _int, _vp, _sz, _vp]),
    "medtok_usage_multi_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "medtok_usage_update_multi": (_int, [_vp, _i64, _vp, _vp, _int, _i64, _vp, _vp, _sz, _vp]),
    "medtok_usage_update_multi_word": (_int, [_vp, _i64, _vp, _vp, _int, _i64, _vp, _vp, _vp, _sz, _vp]),
    "medtok_debug_clock_probe": (_int, [_vp, C.c_uint64, _vp, _vp]),
    "medtok_rownorm_f32": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp]),
    "medtok_search_workspace_bytes": (_sz, [_i64, _i64, _int, _int, _int]),
    "medtok_topk_search_f32": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _sz, _int, _vp]),
    "medtok_merge_topk_lists_f32": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _vp]),
    "medtok_debug_set_attention_probe": (None, [_vp]),
    "medtok_debug_set_half_gemm_k32": (None, [_int]),
    "medtok_debug_filter_probe": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _int, _int, _vp, _sz, _vp, _sz, C.POINTER(C.c_int64), _vp]),
    "medtok_debug_filter_scores_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "medtok_debug_filter_scores_f32": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "medtok_debug_filter_fallback_count_offset": (_sz, [_i64, _i64, _int, _int, _int]),
    "medtok_debug_filter_stats": (_int, [_vp, _sz, _int, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "medtok_soft_assign_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp, _i64, _vp, _vp]),
    "medtok_sum_scale_f32": (_int, [_vp, _i64, _dbl, _vp, _vp]),
    "medtok_soft_vq_backward_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp]),
    "medtok_normalize_backward_f32": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp]),
    "medtok_normalize_backward_sparse_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _vp, _vp]),
    "medtok_info_nce_workspace_bytes": (_sz, [_i64, _int]),
    "medtok_info_nce_forward_f32": (_int, [_vp, _vp, _i64, _int, _f, _vp, _vp, _vp, _sz, _vp]),
    "medtok_info_nce_backward_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _f, _vp, _vp, _vp, _sz, _vp]),
    "medtok_row_dot_f32": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "medtok_small_gemm_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "medtok_frobenius_workspace_bytes": (_sz, [_i64]),
    "medtok_frobenius_f32": (_int, [_vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "medtok_scale_by_device_scalar_f32": (_int, [_vp, _i64, _vp, _vp, _f, _vp, _vp]),
    "medtok_shared_kv_attention_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f, _vp, _vp, _vp, _int, _vp]),
    "medtok_split_half_f32": (_int, [_vp, _i64, _int, _i64, _int, _f, _vp, _vp, _vp, _int, _vp]),
    "medtok_shared_kv_attention_split_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f, _vp, _vp, _vp, _int, _vp]),
    "medtok_split_gemm_f16": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _i64, _int, _int, _int, _int, _int, _vp, _f, _vp, _int, _vp, _vp, _int, _vp]),
    "medtok_half_image_f32": (_int, [_vp, _i64, _int, _i64, _i64, _int, _i64, _int, _vp, _vp]),
    "medtok_half_image_pair_f32": (_int, [_vp, _i64, _int, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp]),
    "medtok_half_image_pair_sums_f32": (_int, [_vp, _i64, _int, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp]),
    "medtok_half_image_t_sums_f32": (_int, [_vp, _i64, _int, _i64, _i64, _i64, _int, _vp, _vp, _vp]),
    "medtok_half_gemm_f32": (_int, [_vp, _i64, _int, _int, _vp, _i64, _int, _int, _int, _int, _int, _vp, _f, _vp, _int, _int, _vp]),
    "medtok_absmax_f32": (_int, [_vp, _i64, _vp, _vp]),
    "medtok_split_half_scaled_f32": (_int, [_vp, _i64, _int, _i64, _i64, _vp, _int, _i64, _vp, _vp, _vp]),
    "medtok_split_gemm_scaled_f16": (_int, [_vp, _vp, _i64, _int, _int, _vp, _vp, _i64, _int, _int, _int, _int, _int, _vp, _f, _vp, _vp, _vp, _int, _vp]),
    "medtok_residual_layernorm_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _f, _vp, _vp]),
    "medtok_residual_layernorm_split_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _f, _vp, _vp, _vp, _int, _vp]),
    "medtok_cross_attention_layer_workspace_bytes": (_sz, [_i64, _int, _int, _int, _int]),
    "medtok_cross_attention_layer_f32": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int,
                                                _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp,
                                                _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _f, _int, _vp, _vp, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "medtok_cross_attention_small_f32": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _vp, _i64, _int, _int, _int, _vp, _f, _f, _vp, _vp, _i64, _i64, _vp, _vp]),
    "medtok_debug_cross_attention_small_exact_f32": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _vp, _i64, _int, _int, _int, _vp, _f, _f, _vp, _vp, _i64, _i64, _vp, _vp]),
    "medtok_segment_mean_f32": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp]),
    "medtok_filter_image_width": (_int, [_int]),
    "medtok_rownorm_image_f32": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _i64, _int, _vp]),
    "medtok_codebook_prepare_f32": (_int, [_vp, _vp, _int, _vp]),
    "medtok_codebook_image_f32": (_int, [_vp, _i64, _int, _vp, _i64, _int, _vp]),
    "medtok_search_resolved_path": (_int, [_i64, _i64, _int, _int, _int]),
    "medtok_soft_vq_forward_prepared_f32": (_int, [_vp, _i64, _int, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "medtok_pack_codes_workspace_bytes": (_sz, [_i64]),
    "medtok_pack_codes": (_int, [_vp, _int, _i64, _i64, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "medtok_pack_codes_checked": (_int, [_vp, _int, _i64, _i64, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "medtok_shared_kv_attention_train_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f, _f, C.c_uint32, _vp, _vp, _vp]),
    "medtok_shared_kv_attention_train_split_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f, _f, C.c_uint32, _vp, _vp, _vp]),
    "medtok_shared_kv_attention_backward_workspace_bytes": (_sz, [_i64]),
    "medtok_shared_kv_attention_backward_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _f, _f, C.c_uint32,
                                                       _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "medtok_shared_kv_attention_backward_half_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _f, _f, C.c_uint32,
                                                            _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp]),
    "medtok_shared_kv_attention_backward_acc_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _f, _f, C.c_uint32,
                                                           _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int, _int, _vp]),
    "medtok_shared_kv_attention_dkv_multi_f32": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _int, _vp]),
    "medtok_ema_stats_workspace_bytes": (_sz, [_i64, _i64]),
    "medtok_ema_stats_f32": (_int, [_vp, _vp, _i64, _int, _i64, _vp, _vp, _vp, _sz, _vp]),
    "medtok_code_histogram_workspace_bytes": (_sz, [_i64]),
    "medtok_code_histogram_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "medtok_ema_apply_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _f, _f, _vp]),
    "medtok_ema_cluster_size_f32": (_int, [_vp, _vp, _i64, _f, _f, _vp]),
    "medtok_usage_workspace_bytes": (_sz, [_i64, _i64]),
    "medtok_usage_update": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "medtok_normalized_search_workspace_bytes": (_sz, [_i64, _i64, _int, _int, _int]),
    "medtok_normalized_search_f32": (_int, [_vp, _i64, _int, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "medtok_soft_vq_workspace_bytes": (_sz, [_i64, _i64, _int, _int, _int]),
    "medtok_soft_vq_forward_f32": (_int, [_vp, _i64, _int, _vp, _vp, _i64, _int, _int,
                                          _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
}


class MedTokLibraryError(RuntimeError):
    pass


class RegionDesc(C.Structure):
    """medtok_region_desc (include/medtok_vq.h)"""
    _fields_ = [("lo", C.c_int64), ("k", C.c_int64), ("wsqp", C.c_void_p), ("en_max", C.c_void_p)]


PREP_MAX_REGIONS = 4


class SearchDesc(C.Structure):
    """medtok_search_desc (include/medtok_vq.h)"""
    _fields_ = [("x", _vp), ("n", _i64), ("what", _vp), ("wsq", _vp), ("k_codes", _i64), ("xhat", _vp), ("idx", _vp), ("dist", _vp), ("w", _vp),
                ("zq", _vp), ("zq_stride", _i64), ("x_stride", _i64), ("row_sqerr", _vp)]


class DkvSource(C.Structure):
    """medtok_dkv_source (include/medtok_vq.h)"""
    _fields_ = [("q", _vp), ("d_out", _vp), ("lse", _vp), ("delta", _vp), ("q_start", _vp), ("q_len", _vp), ("scale", _f), ("dropout_p", _f),
                ("seed", C.c_uint32), ("reserved_", C.c_uint32)]


DKV_SOURCES_MAX = 4
MULTI_SEARCH_MAX = 6
USAGE_MULTI_MAX = 6


def library_path() -> Path:
    return _SO


def load():
    """dlopen the HIP library and type its entry points. Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not _SO.exists():
        raise MedTokLibraryError(
            f"{_SO} is missing: build it with `python medtok_amd/csrc/build.py` "
            "(hipcc --offload-arch=gfx950). medtok_amd has no CPU fallback.")
    # Load order matters: PyTorch-ROCm ships its own libamdhip64 with the same soname as /opt/rocm's.  Whichever is
    # mapped first serves BOTH this library and torch; if ours pulled in /opt/rocm's copy before torch was imported,
    # torch would later run on a runtime it was not built against ("no ROCm-capable device is detected").  torch owns
    # the device memory and streams this library works on, so its runtime goes first.
    import torch  # noqa: F401
    lib = C.CDLL(str(_SO))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    got = lib.medtok_abi_version()
    if got != ABI_VERSION:
        raise MedTokLibraryError(f"ABI version mismatch: library {got}, binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().medtok_last_error().decode(errors="replace")
        raise MedTokLibraryError(f"{what} failed: {msg}")
