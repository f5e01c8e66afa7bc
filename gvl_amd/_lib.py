"""ctypes binding of libgvl_msda.so (C ABI: include/gvl_msda.h).  Fails loudly when the library is missing."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# GVL_LIB_PATH: a dev build (gvl_amd.build.build_dev: timing / ablation variants of single kernels) instead of the shipped library
LIB_PATH = os.environ.get("GVL_LIB_PATH") or os.path.join(_HERE, "libgvl_msda.so")
ABI_VERSION = 17          # include/gvl_msda.h GVL_MSDA_ABI_VERSION
_lock = threading.Lock()
_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_I64 = ctypes.c_int64
_SZ = ctypes.c_size_t

# name -> (restype, argtypes); also the authoritative list of exported symbols for tests/test_abi.py
SIGNATURES = {
    "gvl_msda_abi_version": (_I, []),
    "gvl_last_error": (ctypes.c_char_p, []),
    "gvl_msda_set_impl": (None, [_I]),
    "gvl_msda_last_impl": (_I, []),
    "gvl_msda_last_kernel": (ctypes.c_char_p, []),
    "gvl_reload_env": (None, []),
    "gvl_msda_forward_f32": (_I, [_P] * 5 + [_I] * 8 + [_P, _P, _P, _P]),
    "gvl_msda_forward_f64": (_I, [_P] * 5 + [_I] * 8 + [_P, _P, _P, _P]),
    "gvl_msda_sample_f32": (_I, [_P] * 4 + [_I] * 8 + [_P, _P]),
    "gvl_msda_sample_f64": (_I, [_P] * 4 + [_I] * 8 + [_P, _P]),
    "gvl_msda_backward_workspace_bytes": (_SZ, [_I] * 8 + [_P]),
    "gvl_msda_backward_f32": (_I, [_P] * 6 + [_I] * 8 + [_P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gvl_msda_backward_f64": (_I, [_P] * 6 + [_I] * 8 + [_P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gvl_msda_forward_bf16": (_I, [_P] * 5 + [_I] * 8 + [_P, _P, _P, _P]),
    "gvl_msda_backward_bf16": (_I, [_P] * 6 + [_I] * 8 + [_P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gvl_prof_enable": (_I, [_I]),
    "gvl_msda_debug_stamps": (None, [_P]),
    "gvl_clock_probe": (_I, [_P, _I, _P]),
    "gvl_f16_products": (_I, [_I]),
    "gvl_residual_dropout_layer_norm_forward_f32": (_I, [_P, _I64, _I64, _P, _I64, _I64, _I, _I, _I, _P, _P, ctypes.c_float, ctypes.c_float,
                                                         ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "gvl_residual_dropout_layer_norm_backward_f32": (_I, [_P, _P, _P, _P, _I, _I, _P, ctypes.c_float, ctypes.c_uint32, _P, _P, _P,
                                                          _P, _P, _P, _P]),
    "gvl_residual_dropout_layer_norm_backwardn_f32": (_I, [_P, _I, _P, _P, _P, _I, _I, _P, ctypes.c_float, ctypes.c_uint32, _P, _P,
                                                           _P, _P, _P, _P, _P]),
    "gvl_rdln_backward_max_grads": (_I, []),
    "gvl_relu_dropout_rows_forward_f32": (_I, [_P, _I, _I, ctypes.c_float, ctypes.c_uint32, _P, _P, _P, _P, _P]),
    "gvl_relu_dropout_rows_backward_f32": (_I, [_P, _P, _I, _I, ctypes.c_float, _P, _P, _P]),
    "gvl_rdln_backward_blocks": (_I, [_I]),
    "gvl_advance_step": (_I, [_P, _P]),
    "gvl_relu_dropout_forward_f32": (_I, [_P, _I64, ctypes.c_float, ctypes.c_uint32, _P, _P, _P]),
    "gvl_relu_dropout_backward_f32": (_I, [_P, _P, _I64, ctypes.c_float, _P, _P]),
    "gvl_prof_collect": (_I, [_P, _P, _P, _P, _I]),
    "gvl_msda_sample_backward_f32": (_I, [_P] * 5 + [_I] * 8 + [_P, _P, _P]),
    "gvl_msda_sample_backward_f64": (_I, [_P] * 5 + [_I] * 8 + [_P, _P, _P]),
    "gvl_cap_attend_f32": (_I, [_P] * 9 + [ctypes.c_float] + [_I] * 8 + [_P, _P, _P, _P]),
    "gvl_lstm_cell_f32": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P]),
    "gvl_lstm_cell_split_f32": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "gvl_cap_attend_split_f32": (_I, [_P] * 9 + [ctypes.c_float] + [_I] * 8 + [_P, _P, _P, _P]),
    "gvl_cap_attend_split_levels_f32": (_I, [_P] * 9 + [ctypes.c_float] + [_I] * 8 + [_P, _P, _P, _P, _P]),
    "gvl_cap_attend_pre_applicable": (_I, [_I, _I, _I, _P]),
    "gvl_cap_attend_pre_f32": (_I, [_P] * 6 + [_I] + [_P, _P, ctypes.c_float] + [_I] * 8 + [_P, _P, _P, _P, _P]),
    "gvl_cap_attend_train_forward_f32": (_I, [_P] * 6 + [_I, _P, _I, _P, _P] + [_I] * 7 + [_P, _P, _P, _P]),
    "gvl_cap_attend_train_backward_f32": (_I, [_P] * 6 + [_I, _P, _I, _P, _P, _P, _I] + [_I] * 7
                                          + [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P]),
    "gvl_lstm_cell_train_forward_f32": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P]),
    "gvl_lstm_cell_train_backward_f32": (_I, [_P] * 6 + [_I, _I, _P, _I, _P, _P]),
    "gvl_lstm_cell_train_backward_sum_f32": (_I, [_P] * 6 + [_I, _I, _P, _I, _P, _P, _I, _P]),
    "gvl_col_sum_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "gvl_ce_rows_forward_f32": (_I, [_P, _I64, _I, _I, _P, _P, _P, _P, _P]),
    "gvl_ce_rows_backward_f32": (_I, [_P, _I64, _I, _I, _P, _P, _P, _P, _P, _P]),
    "gvl_proj_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "gvl_split_rows_f16": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "gvl_gemm_f16x3_gates_applicable": (_I, [_I, _I]),
    "gvl_gemm_f16x3_gates_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gvl_gemm_f16x3_lstm_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gvl_gemm_f16x3_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, ctypes.c_int64, _P]),
    "gvl_gemm_f16x3_argmax_chunks": (_I, [_I]),
    "gvl_gemm_f16x3_argmax_f32": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "gvl_greedy_step_partials_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "gvl_pos_embed_sine_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, ctypes.c_float, _P, _P]),
    "gvl_match_cost_f32": (_I, [_P] * 4 + [_I] * 5 + [ctypes.c_float] * 5 + [_P, _P, _P]),
    "gvl_match_cost_padded_f32": (_I, [_P] * 5 + [_I] * 5 + [ctypes.c_float] * 5 + [_P, _P, _P]),
    "gvl_set_criterion_forward_f32": (_I, [_P] * 12 + [_I] * 7 + [ctypes.c_float] * 4 + [_I, _P, _P, _P, _P]),
    "gvl_set_criterion_backward_f32": (_I, [_P] * 12 + [_I] * 7 + [ctypes.c_float] * 4 + [_I, _P, _P, _P, _P, _P, _P, _P]),
    "gvl_row_argmax_lse_f32": (_I, [_P, _I, _I, _P, _P, _P]),
    "gvl_cap_attend_bf16": (_I, [_P] * 9 + [ctypes.c_float] + [_I] * 8 + [_P, _P, _P, _P]),
    "gvl_lstm_cell_bf16": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P]),
    "gvl_row_argmax_lse_bf16": (_I, [_P, _I, _I, _P, _P, _P]),
    "gvl_greedy_step_bf16": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "gvl_greedy_step_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "gvl_msda1d_fused_forward_f32": (_I, [_P] * 5 + [_I] * 9 + [_P, _P, _P, _P]),
    "gvl_msda1d_fused_forward_amax_f32": (_I, [_P] * 5 + [_I] * 9 + [_P, _P, _P, _P, _P]),
    "gvl_msda1d_fused_forward_shared_amax_f32": (_I, [_P] * 5 + [_I] * 9 + [_P, _P, _P, _P, _P]),
    "gvl_linear_f16x3_f32": (_I, [_P, _I64, _P, _I64, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "gvl_layer_norm_rows_f32": (_I, [_P, _I, _I, _P, _P, ctypes.c_float, _P, _I, _P, _P, _P, _P]),
    "gvl_row_absmax_f32": (_I, [_P, _I64, _I, _I, _P, _I64, _I, _P, _P, _P]),
    "gvl_mha_core_f32": (_I, [_P, _I64, _P, _I, _I, _I, _P, _P, _P]),
    "gvl_group_norm_rows_backward_f32": (_I, [_P, _I64, _I, _I, _I, _I, _I, _P, ctypes.c_float, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P]),
    "gvl_group_norm_rows_backward_amax_f32": (_I, [_P, _I64, _I, _I, _I, _I, _I, _P, ctypes.c_float, _P, _I64, _P, _I64, _P, _I64, _P, _P,
                                                   _P, _P]),
    "gvl_conv_taps_to_rows_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "gvl_group_norm_rows_f32": (_I, [_P, _I64, _I, _I, _I, _I, _I, _P, _P, ctypes.c_float, _P, _I64, _P, _I64, _P]),
    "gvl_pyramid_geometry_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, ctypes.c_float, _P, _P, _P]),
    "gvl_encoder_geometry_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "gvl_greedy_step_partials_alive_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "gvl_greedy_step_partials_gemm_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _I64, _P]),
    "gvl_box_refine_f32": (_I, [_P, _I64, _P, _I, _P, _I, _I, _I, _P, _P, _P]),
    "gvl_count_head_f32": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "gvl_count_pool_f32": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "gvl_batch_sum_f32": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "gvl_level_sums_f32": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "gvl_mask_rows_f32": (_I, [_P, _P, _I, _I, _P]),
    "gvl_index_add_rows_f32": (_I, [_P, _I64, _P, _I, _I, _P, _I64, _I, _P]),
    "gvl_mask_rows_backward_f32": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "gvl_count_pool_backward_f32": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "gvl_msda1d_fused_backward_workspace_bytes": (_SZ, [_I] * 7 + [_P]),
    "gvl_msda1d_fused_backward_f32": (_I, [_P] * 6 + [_I] * 9 + [_P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gvl_msda1d_fused_forward_bf16": (_I, [_P] * 5 + [_I] * 9 + [_P, _P, _P, _P]),
    "gvl_msda1d_fused_backward_bf16": (_I, [_P] * 6 + [_I] * 9 + [_P, _P, _P, _P, _P, _P, _SZ, _P]),
    "gvl_wgrad_workspace_bytes": (_SZ, [_I, _I, _I]),
    "gvl_box_refine_backward_f32": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "gvl_caption_rows": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gvl_wgrad_group_max": (_I, []),
    "gvl_wgrad_group_workspace_bytes": (_SZ, [_P, _I]),
    "gvl_wgrad_group_f16x3_f32": (_I, [_P, _I, _P, _SZ, _P]),
    "gvl_wgrad_f16x3_f32": (_I, [_P, _I64, _P, _I, _P, _I64, _P, _I, _I, _I, _I, _P, _P, _I, _P, _SZ, _P]),
    "gvl_wgrad_f16x3_live_f32": (_I, [_P, _I64, _P, _I, _P, _I64, _P, _I, _I, _I, _I, _P, _P, _I, _P, _SZ, _P, _P]),
    "gvl_wgrad_live_ints": (_I, [_I]),
    "gvl_planes_chunk_elems": (_I, []),
    "gvl_planes_refresh_f16": (_I, [_P, _P, _I, _P, _I, _P, _P]),
    "gvl_mha_train_forward_f32": (_I, [_P, _I64, _P, _P, _P, _I, _I, _I, ctypes.c_float, ctypes.c_uint32, _P, _P, _P, _P, _P]),
    "gvl_mha_train_backward_f32": (_I, [_P, _I64, _P, _P, _P, _I, _I, _I, ctypes.c_float, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gvl_mha_train_backward_amax_f32": (_I, [_P, _I64, _P, _P, _P, _I, _I, _I, ctypes.c_float, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P,
                                             _P, _P, _P]),
    "gvl_linear_f16x3_splitk_workspace_bytes": (_SZ, [_I, _I, _I]),
    "gvl_linear_f16x3_splitk_f32": (_I, [_P, _I64, _P, _I, _I, _P, _P, _P, _I, _P, _P, _SZ, _P]),
    "gvl_linear_f16x3_splitk_bias_f32": (_I, [_P, _I64, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _SZ, _P]),
    "gvl_adam_chunk_elems": (_I, []),
    "gvl_adam_set_grads": (_I, [_P, _I, _P, _P]),
    "gvl_clip_adam_step_f32": (_I, [_P, _I, _P, _I, _P, _P, _P] + [ctypes.c_double] * 6 + [_P]),
    "gvl_lsap_solve_f64": (_I, [_P, _I64, _I64, _P, _P]),
    "gvl_lsap_solve_f32": (_I, [_P, _I64, _I64, _P, _P]),
    "gvl_lsap_batch_device_f32": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "gvl_hungarian_batch_f32": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I]),
}


class GvlLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises GvlLibraryError (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise GvlLibraryError(
                        f"{LIB_PATH} is missing: build it with `python -m gvl_amd.build` "
                        "(hipcc --offload-arch=gfx950).  gvl_amd has no CPU / PyTorch fallback.")
                # libgvl_msda.so needs libamdhip64.so.7.  In a PyTorch process that must be the HIP runtime PyTorch ships
                # and allocates with: loaded first it is found by SONAME; loaded after this library, the process would
                # hold two runtimes and every launch here would fail with "no ROCm-capable device".
                import torch  # noqa: F401
                try:
                    handle = ctypes.CDLL(LIB_PATH)
                except OSError as e:  # pragma: no cover
                    raise GvlLibraryError(f"cannot load {LIB_PATH}: {e}") from e
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                if handle.gvl_msda_abi_version() != ABI_VERSION:
                    raise GvlLibraryError("libgvl_msda.so ABI version mismatch; rebuild")
                _lib = handle
    return _lib


def reload_env():
    """the library re-reads its GVL_* switches on their next use (they are cached per process); a no-op before it is loaded"""
    if _lib is not None:
        _lib.gvl_reload_env()


def check(rc, what):
    if rc != 0:
        msg = lib().gvl_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
