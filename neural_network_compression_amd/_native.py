"""ctypes binding of include/nnc.h (csrc/libnnc_hip.so).

There is NO CPU fallback: if the HIP library is missing or fails to load, every product
entry point raises ``NativeLibraryError``.
"""
from __future__ import annotations

import ctypes
import os

from . import build as _build

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_size = ctypes.c_size_t

NNC_OK = 0
NNC_KMAX = 1040
NNC_CHUNK = 8192
FOLD_SUM, FOLD_MEAN, FOLD_STD = 0, 1, 2
NNC_KM_TWO_LAUNCH = 1   # nnc_kmeans_params.flags: iterate launch by launch (include/nnc.h)
NNC_KM_LOOP = 2         # ... inside one resident workgroup whatever K
NNC_KM_MASS_IN_PLACE = 4  # ... nnc_kmeans_fit: mass empty-cluster events settled by the finalize step (experiment; include/nnc.h)
NNC_KM_LOOP_KMAX = 64  # up to here the library takes the loop by itself


class NativeLibraryError(RuntimeError):
    pass


class NncError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"nnc error {code}: {message}")
        self.code = code


class KMeansParams(ctypes.Structure):
    _fields_ = [
        ("n", c_i64), ("n_total", c_i64), ("k", c_i32), ("max_iter", c_i32), ("fix_shift", c_i32),
        ("grid_log2", c_i32), ("replicas_log2", c_i32), ("flags", c_i32),
        ("x_mean", c_f32), ("tol", c_f32), ("lo", c_f32), ("hi", c_f32),
        ("prefix_dev", c_void_p),
    ]


class KMeansStatus(ctypes.Structure):
    _fields_ = [
        ("iter", c_i32), ("done", c_i32), ("paused", c_i32), ("n_empty", c_i32),
        ("shift_tot", c_f32), ("tol", c_f32), ("k", c_i32), ("same_counts", c_i32),
        ("reloc_ties", c_i32), ("reloc_multi", c_i32), ("n_relocated", c_i32), ("n_unproven", c_i32),
        ("n_in_place", c_i32), ("reserved", c_i32),
    ]


class LayerParams(ctypes.Structure):
    _fields_ = [("q", c_f32), ("prune", c_i32), ("std_smooth", c_i32), ("bits", c_i32), ("mode", c_i32), ("want_values", c_i32),
                ("km_flags", c_i32), ("reserved", c_i32)]


class LayerResult(ctypes.Structure):
    _fields_ = [
        ("status", c_i32), ("k", c_i32), ("label_bytes", c_i32), ("arith", c_i32),
        ("n_iter", c_i32), ("stop", c_i32), ("n_relocations", c_i32), ("n_reloc_windowed", c_i32), ("reloc_ties", c_i32), ("reloc_multi", c_i32),
        ("sigma", c_f32), ("threshold", c_f32),
        ("n_zeroed", c_i64), ("total_bits", c_i64),
        ("centers", c_f32 * NNC_KMAX), ("counts", c_i64 * NNC_KMAX), ("code_lengths", ctypes.c_uint8 * NNC_KMAX),
    ]


# name -> (restype, argtypes); every symbol include/nnc.h declares
SIGNATURES = {
    "nnc_version": (c_int, []),
    "nnc_last_error": (ctypes.c_char_p, []),
    "nnc_device_info": (c_int, [ctypes.c_char_p, c_size, ctypes.POINTER(c_int)]),
    "nnc_chunk_sums_f32": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "nnc_fold_f32": (c_int, [c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "nnc_prune_workspace_bytes": (c_size, [c_i64]),
    "nnc_prune_f32": (c_int, [c_void_p, c_i64, c_f32, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_sort_pruned_bounded_bits": (c_i32, [c_f32, c_f32, c_f32, c_i64, c_i64]),
    "nnc_sort_pruned_bounded_workspace_bytes": (c_size, [c_i64]),
    "nnc_sort_pruned_bounded_flag": (c_void_p, [c_void_p, c_i64]),
    "nnc_sort_pruned_bounded_f32": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_f32, c_f32, c_f32, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_prune_stats_workspace_bytes": (c_size, [c_i64]),
    "nnc_prune_stats_f32": (c_int, [c_void_p, c_i64, c_f32, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_threshold_mask_f32": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nnc_apply_mask_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "nnc_minmax_workspace_bytes": (c_size, [c_i64]),
    "nnc_minmax_f32": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_minmax_signs_f32": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_layer_stats_workspace_bytes": (c_size, [c_i64]),
    "nnc_layer_stats_f32": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_rank_sorted_f32": (c_int, [c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_void_p]),
    "nnc_hist31_f32": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "nnc_sort_workspace_bytes": (c_size, [c_i64]),
    "nnc_sort_f32": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_sort_pruned_workspace_bytes": (c_size, [c_i64, c_i64, c_i64]),
    "nnc_sort_pruned_f32": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_fix_shift": (c_i32, [c_f32, c_i64]),
    "nnc_kmeans_workspace_bytes": (c_size, [c_i32]),
    "nnc_kmeans_prefix_bytes": (c_size, [c_i64]),
    "nnc_kmeans_loop_stats": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nnc_kmeans_prefix_build": (c_int, [c_void_p, ctypes.POINTER(KMeansParams), c_void_p, c_void_p]),
    "nnc_kmeans_init": (c_int, [c_void_p, c_size, ctypes.POINTER(KMeansParams), c_void_p, c_void_p]),
    "nnc_kmeans_set_centers": (c_int, [c_void_p, ctypes.POINTER(KMeansParams), c_void_p, c_int, c_void_p]),
    "nnc_kmeans_accumulate": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_void_p]),
    "nnc_kmeans_partials": (c_void_p, [c_void_p]),
    "nnc_kmeans_finalize": (c_int, [c_void_p, c_int, c_void_p]),
    "nnc_kmeans_iterate": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_void_p]),
    "nnc_kmeans_status_async": (c_int, [c_void_p, ctypes.POINTER(KMeansStatus), c_void_p]),
    "nnc_kmeans_status_publish": (c_int, [c_void_p, c_void_p, ctypes.c_uint64, c_void_p]),
    "nnc_kmeans_iterate_publish": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_void_p, ctypes.c_uint64, c_void_p]),
    "nnc_kmeans_fit": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_i32, c_void_p, c_size, c_void_p,
                               ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(KMeansStatus), ctypes.POINTER(c_i32), c_void_p]),
    "nnc_kmeans_fit_sharded": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(KMeansParams), ctypes.c_int64, c_i32, c_i32, c_void_p, c_size,
                                       c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(KMeansStatus), ctypes.POINTER(c_i32), c_void_p]),
    "nnc_kmeans_set_done": (c_int, [c_void_p, c_i32, c_void_p]),
    "nnc_kmeans_label_counts": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_int, c_void_p, c_void_p]),
    "nnc_kmeans_get_centers": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nnc_kmeans_assign": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nnc_topm_hist_f32": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_i32, ctypes.c_uint32, c_void_p, c_void_p]),
    "nnc_topm_compact_f32": (c_int, [c_void_p, c_void_p, c_i64, ctypes.c_uint32, c_void_p, c_i64, c_void_p, c_void_p]),
    "nnc_kmeans_relocate": (c_int, [c_void_p, c_void_p, c_i32, c_void_p]),
    "nnc_kmeans_reloc_candidates": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "nnc_kmeans_relocate_checked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p]),
    "nnc_kmeans_reloc_window": (c_i32, [c_i64, c_i32]),
    "nnc_kmeans_reloc_scratch_bytes": (c_size, [c_i32, c_i32]),
    "nnc_kmeans_relocate_windowed": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_void_p, c_size, c_void_p]),
    "nnc_kmeans_reloc_select_local": (c_int, [c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_void_p, c_size, c_void_p, c_void_p]),
    "nnc_kmeans_reloc_flag": (c_void_p, [c_void_p]),
    "nnc_kmeans_relocate_if_proven": (c_int, [c_void_p, c_void_p, c_i32, c_void_p]),
    "nnc_labels_equal": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "nnc_ref_sums_f32": (c_int, [c_void_p, c_i64, ctypes.c_float, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "nnc_kmeans_set_done_if": (c_int, [c_void_p, c_void_p, c_i32, c_void_p]),
    "nnc_bincount": (c_int, [c_void_p, c_int, c_i64, c_i32, c_void_p, c_void_p]),
    "nnc_kmeans_fit_reference_f32": (c_int, [c_void_p, c_i32, c_void_p, c_i32, c_i32, c_f32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nnc_compress_layer_workspace_bytes": (c_size, [c_i64, c_i32]),
    "nnc_compress_layer_host_bytes": (c_size, []),
    "nnc_compress_layer_f32": (c_int, [c_void_p, c_i64, ctypes.POINTER(LayerParams), c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p, c_size,
                                       ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(LayerResult), c_void_p]),
    "nnc_host_linspace_f32": (c_int, [c_f32, c_f32, c_i32, c_void_p]),
    "nnc_host_cdf": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nnc_host_density_init": (c_int, [c_void_p, c_void_p, c_i32, c_void_p]),
    "nnc_kmeanspp_trials": (c_i32, [c_i32]),
    "nnc_kmeanspp_workspace_bytes": (c_size, [c_i64, c_i32]),
    "nnc_kmeanspp_seed_f32": (c_int, [c_void_p, c_i64, c_f32, c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "nnc_centroid_grad_f32": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "nnc_gather_f32": (c_int, [c_void_p, c_i32, c_void_p, c_int, c_i64, c_void_p, c_void_p]),
    "nnc_huffman_codes": (c_int, [ctypes.POINTER(ctypes.c_uint8), c_i32, ctypes.POINTER(ctypes.c_uint32)]),
    "nnc_codec_chunks": (c_size, [c_i64]),
    "nnc_huffman_chunk_offsets": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_i32, c_void_p, c_void_p]),
    "nnc_huffman_encode": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_i64, c_void_p]),
    "nnc_huffman_decode_tables_bytes": (c_size, []),
    "nnc_huffman_decode_tables": (c_int, [ctypes.POINTER(ctypes.c_uint8), c_i32, c_void_p, c_size]),
    "nnc_huffman_decode": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_int, c_void_p, c_void_p]),
    "nnc_sparse_entry_offsets": (c_int, [c_void_p, c_int, c_i64, c_i32, c_i32, c_void_p, c_void_p]),
    "nnc_sparse_emit": (c_int, [c_void_p, c_int, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nnc_sparse_expand": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p]),
    "nnc_comm_unique_id": (c_int, [c_void_p, c_size]),
    "nnc_comm_init": (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_size, c_i32, c_i32]),
    "nnc_comm_destroy": (c_int, [c_void_p]),
    "nnc_comm_rank": (c_int, [c_void_p]),
    "nnc_comm_world": (c_int, [c_void_p]),
    "nnc_comm_allreduce": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_void_p]),
    "nnc_comm_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "nnc_kmeans_iterate_sharded": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_void_p, ctypes.c_uint64, c_void_p]),
    "nnc_merge_keys": (c_int, [c_void_p, c_i32, c_i32, c_void_p, c_i32, c_void_p]),
    "nnc_kmeans_reloc_scratch_bytes_sharded": (c_size, [c_i32, c_i32, c_i32]),
    "nnc_kmeans_relocate_windowed_sharded": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(KMeansParams), c_i32, c_void_p, c_size, c_void_p]),
    "nnc_comm_available": (c_int, []),
    "nnc_profile_tags": (c_int, [ctypes.c_uint32]),
    "nnc_profile_begin": (c_int, [c_i32]),
    "nnc_profile_end": (c_int, [ctypes.POINTER(c_f32), ctypes.POINTER(c_i32), c_i64, ctypes.POINTER(c_i64)]),
    "nnc_huffman_lengths": (c_int, [ctypes.POINTER(c_i64), c_i32, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
}

# exported only by the diagnostics build (NNC_DIAG=1: libnnc_hip_diag.so, see build.py); bound when present
DIAG_SIGNATURES = {
    "nnc_debug_set_ablation": (c_int, [c_int]),
    "nnc_debug_set_trace": (c_int, [c_void_p]),
    "nnc_debug_clock": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "nnc_debug_reloc_fail": (c_int, [c_void_p, ctypes.POINTER(c_i32)]),
    "nnc_debug_kl_trace": (c_int, [c_void_p, c_void_p]),
    "nnc_debug_os_trace": (c_int, [c_void_p]),
}

_lib = None


def lib_path() -> str:
    return _build.LIB


def load():
    """Load libnnc_hip.so (must have been built: ``__graft_entry__.build()`` or
    ``python -m neural_network_compression_amd.build``).  Raises NativeLibraryError."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -m neural_network_compression_amd.build`); there is no CPU fallback")
    try:
        L = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise NativeLibraryError(f"cannot load {path}: {e}; there is no CPU fallback") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DIAG_SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = L
    return L


def check(rc: int):
    if rc != NNC_OK:
        raise NncError(rc, load().nnc_last_error().decode("utf-8", "replace"))
