"""Loads libpats_amd.so (the C-ABI of include/pats_amd.h) with ctypes.

There is NO fallback: if the HIP library is missing or a call fails, this raises.  Nothing under
oracle/ is ever imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpats_amd.so")
# PATS_AMD_DIAG_LIB=1 (profiling tools only): the diagnostic twin with every third-level sweep variant and the timing
# ablations (`python -m pats_amd.build --diag`).  The production library refuses those variants.
if os.environ.get("PATS_AMD_DIAG_LIB", "") not in ("", "0"):
    _sfx = os.environ["PATS_AMD_DIAG_LIB"]
    LIB_PATH = os.path.join(_HERE, "libpats_amd_diag%s.so" % ("" if _sfx == "1" else _sfx))

c_void_p, c_int, c_i64, c_f, c_size = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                       ctypes.c_size_t)

ABI_VERSION = 6      # include/pats_amd.h PATS_ABI_VERSION

# name -> (restype, argtypes); must list every symbol include/pats_amd.h declares
SIGNATURES = {
    "pats_version": (ctypes.c_char_p, []),
    "pats_abi_version": (c_int, []),
    "pats_last_error": (ctypes.c_char_p, []),
    "pats_device_count": (c_int, []),
    "pats_set_sinkhorn_mode": (c_int, [c_int]),
    "pats_set_fine_fused": (c_int, [c_int]),
    "pats_sinkhorn_fallbacks": (c_int, [ctypes.POINTER(ctypes.c_int64), c_int]),
    "pats_set_gnn_redo_mode": (c_int, [c_int]),
    "pats_gnn_overflows": (c_int, [ctypes.POINTER(ctypes.c_int64), c_int]),
    "pats_cost_f32": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pats_sinkhorn_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "pats_sinkhorn_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p, c_size, c_void_p]),
    "pats_ot_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "pats_log_optimal_transport_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p, c_size, c_void_p]),
    "pats_ot2_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "pats_log_optimal_transport2_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_int,
                                                c_f, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_cost_ot_f32": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p,
                                 c_void_p, c_int, c_f, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_cost_ot_flags_f32": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_int, c_f, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_set_cost_ot_mid_event": (c_int, [c_void_p]),
    "pats_cost_ot_flags_counted_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                               c_void_p, c_int, c_f, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_iterative_expand_counted_f32": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                                  c_int, c_int, c_f, c_int, c_void_p, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pats_fine_descriptors_counted_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_int,
                                                  c_void_p, c_void_p]),
    "pats_log_optimal_transport2_flags_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_int,
                                                      c_f, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_colmass_flags_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pats_cost_ot_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int]),
    "pats_colmass_sqrt_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p]),
    "pats_dustbin_bias_inplace_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_f, c_void_p]),
    "pats_exp_f32": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "pats_argmax_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pats_iterative_expand_f32": (c_int, [c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_int,
                                          c_int, c_int, c_f, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pats_split_patches": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pats_split_patches_device": (c_int, [c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pats_compute_imgs_bounds_batch_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pats_left_crops_counted_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_i64, c_void_p, c_int, c_int,
                                            c_void_p, c_void_p]),
    "pats_tensor_resize_hwc_counted_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_i64, c_void_p,
                                                   c_void_p, c_void_p, c_void_p]),
    "pats_compute_imgs_bounds_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                             c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "pats_left_crops_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_i64, c_int, c_int, c_void_p,
                                    c_void_p]),
    "pats_tensor_resize_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_i64, c_void_p,
                                       c_void_p, c_void_p]),
    "pats_tensor_resize_hwc_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_i64,
                                           c_void_p, c_void_p, c_void_p]),
    "pats_compute_result_f32": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "pats_compute_result_ws_f32": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_size, c_void_p]),
    "pats_fine_descriptors_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "pats_third_descriptors_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pats_third_level_f32": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "pats_third_descriptors_counted_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pats_fine_descriptors_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "pats_third_descriptors_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pats_third_level_counted_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "pats_chunk_rows_workspace_bytes": (c_size, [c_i64, c_int]),
    "pats_chunk_rows_device": (c_int, [c_void_p, c_i64, c_int, c_int, c_int, c_int, c_i64] + [c_void_p] * 13 +
                               [c_size, c_void_p]),
    "pats_profile_marker": (c_int, [c_int, c_void_p]),
    "pats_stream_create_cu_mask": (c_int, [c_void_p, c_int, c_void_p]),
    "pats_stream_destroy": (c_int, [c_void_p]),
    "pats_merge_batch_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "pats_merge_patches_batch": (c_int, [c_int, c_int, c_i64, c_int, c_int, c_i64] + [c_void_p] * 7 + [c_int, c_void_p, c_void_p,
                                         c_size, c_void_p]),
    "pats_merge_patches_chunks": (c_int, [c_int, c_int, c_int, c_int, c_i64, c_int, c_int, c_i64, c_i64] + [c_void_p] * 7 +
                                  [c_int, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_chunk_fine_tail_workspace_bytes": (c_size, [c_i64, c_i64, c_int, c_int]),
    "pats_chunk_fine_tail_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_f, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_i64, c_int, c_int, c_i64] + [c_void_p] * 5 + [c_int] + [c_void_p] * 16 + [c_void_p, c_size, c_void_p]),
    "pats_chunk_third_tail_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "pats_chunk_third_tail_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                          c_i64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p] + [c_void_p] * 10 +
                                  [c_void_p, c_size, c_void_p]),
    "pats_get_result_chunks_f32": (c_int, [c_int, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
                                           ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_merge_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "pats_merge_patches": (c_int, [c_int, c_i64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size, c_void_p]),
    "pats_compact_workspace_bytes": (c_size, [c_i64]),
    "pats_third_inputs_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_void_p,
                                      c_void_p, c_size, c_void_p]),
    "pats_refine_scatter_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p,
                                        c_void_p, c_void_p, c_size, c_void_p]),
    "pats_attention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p]),
    "pats_attentional_propagation_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "pats_attentional_propagation_f32": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p, c_int, c_f,
                                                 c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_propagation_packed_bytes": (c_size, [c_int, c_int]),
    "pats_propagation_pack_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size, c_void_p]),
    "pats_attentional_propagation_packed_f32": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                        c_int, c_f, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_attentional_gnn_packed_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "pats_attentional_gnn_packed_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                                c_void_p, c_f, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_attentional_propagation_packed_counted_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_int, c_int,
                                                                c_void_p, c_void_p, c_int, c_f, c_void_p, c_void_p, c_void_p, c_size,
                                                                c_void_p]),
    "pats_matches_by_pair_workspace_bytes": (c_size, [c_int, c_i64]),
    "pats_matches_by_pair_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_matches_by_pair_summary_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int,
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "pats_conv1x1_workspace_bytes": (c_size, []),
    "pats_conv1x1_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_size, c_void_p]),
    "pats_bn_fold_workspace_bytes": (c_size, [c_int]),
    "pats_bn_fold_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p, c_f, c_void_p, c_void_p, c_void_p,
                                 c_size, c_void_p]),
    "pats_scale_head_f32": (c_int, [c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p]),
    "pats_get_result_workspace_bytes": (c_size, [c_i64, c_i64, c_i64]),
    "pats_get_result_f32": (c_int, [c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                                    ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_i64, c_void_p, c_void_p, c_size, c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pats_amd: %s is missing - the HIP extension was not built (run "
                "`python -m pats_amd.build` / __graft_entry__.build()). There is no CPU fallback."
                % LIB_PATH)
        # PyTorch ships its own libamdhip64; if this library were loaded first it would bring in /opt/rocm's copy, torch
        # would then load its own beside it, and the kernels launched here would talk to a runtime that has no
        # device context ("no ROCm-capable device is detected" - seen when build() and smoke() ran in one process).
        # Loading torch's runtime first makes both resolve to the same one.
        try:
            import torch  # noqa: F401
        except ImportError:      # a C-ABI-only consumer without torch: nothing to clash with
            pass
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)     # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.pats_abi_version() != ABI_VERSION:
            raise ImportError("pats_amd: %s exports ABI %d, these bindings were written for ABI %d - rebuild the library "
                              "(python -m pats_amd.build)" % (LIB_PATH, handle.pats_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pats_last_error().decode()
        raise RuntimeError("pats_amd.%s failed (code %d): %s" % (what, rc, msg))
