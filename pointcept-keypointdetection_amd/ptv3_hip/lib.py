"""Loads lib/libptv3_hip.so and declares the C ABI of include/ptv3_hip.h for ctypes."""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint32, c_void_p

PTV3_F32, PTV3_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
ORDER_IDS = {"z": 0, "z-trans": 1, "hilbert": 2, "hilbert-trans": 3}

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    return os.environ.get("PTV3_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libptv3_hip.so"))


P = c_void_p


class BlockTrain(ctypes.Structure):
    """struct ptv3_block_train of include/ptv3_hip.h (field for field)"""
    _fields_ = ([("n", c_int64), ("n_pad", c_int64)] +
                [(k, ctypes.c_int32) for k in ("c", "hidden", "heads", "patch", "kvol", "num_windows", "dtype", "reserved")] +
                [("scale", c_float), ("eps", c_float), ("keep1", c_float), ("keep2", c_float), ("sum_len_sq", c_double)] +
                [(k, P) for k in ("nbr", "row_order", "win_order", "win_inverse", "cu_seqlens", "feat", "conv_feat",
                                  "w_conv", "w_lin", "w_qkv", "w_proj", "w_fc1", "w_fc2",
                                  "wt_conv", "wt_lin", "wt_qkv", "wt_proj", "wt_fc1", "wt_fc2",
                                  "b_conv", "b_lin", "b_qkv", "b_proj", "b_fc1", "b_fc2", "g0", "b0", "g1", "b1", "g2", "b2",
                                  "mask1", "mask2",
                                  "c1", "c2", "f1", "t3", "qkv", "a", "f2", "t5", "h0", "h", "out",
                                  "dout", "dfeat", "dconv_feat",
                                  "dw_conv", "dw_lin", "dw_qkv", "dw_proj", "dw_fc1", "dw_fc2",
                                  "db_conv", "db_lin", "db_qkv", "db_proj", "db_fc1", "db_fc2",
                                  "dln0", "dln1", "dln2", "workspace")] +
                [("workspace_bytes", c_size_t), ("attn_lse", P)])


# name -> (restype, argtypes); must list every symbol include/ptv3_hip.h declares
SIGNATURES = {
    "ptv3_last_error": (c_char_p, []),
    "ptv3_version": (c_int, []),
    "ptv3_sfc_encode": (c_int, [P, c_int, P, c_int64, c_int, P, c_int, P, P]),
    "ptv3_argsort_workspace_bytes": (c_size_t, [c_int, c_int64]),
    "ptv3_argsort_i64": (c_int, [P, c_int, c_int64, c_int, P, P, P, c_size_t, P]),
    "ptv3_pad_plan": (c_int, [P, c_int, c_int64, c_int64, c_int, P, P, P, P]),
    "ptv3_window_maps": (c_int, [P, P, P, P, c_int64, c_int64, P, P, P]),
    "ptv3_window_plan": (c_int, [P, P, P, c_int, c_int, c_int64, c_int64, c_int, P, P, P, P]),
    "ptv3_window_attn_fwd": (c_int, [P, P, P, P, c_int64, c_int64, c_int, c_int, c_int, c_float, P, c_int, P]),
    "ptv3_window_attn_varlen_fwd": (c_int, [P, P, P, P, c_int, P, c_int64, c_int64, c_int, c_int, c_int, c_float,
                                            c_double, c_int, P]),
    "ptv3_window_attn_rpe_fwd": (c_int, [P, P, P, P, c_int64, c_int64, c_int, c_int, c_int, c_float, P, P, c_int,
                                         c_int, P]),
    "ptv3_subm_table_slots": (c_int64, [c_int64]),
    "ptv3_subm_build_table": (c_int, [P, c_int64, P, c_int64, P]),
    "ptv3_subm_neighbors": (c_int, [P, c_int64, P, c_int64, c_int, P, P]),
    "ptv3_gemm_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "ptv3_gemm": (c_int, [P, P, P, c_int64, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P, c_int, P, c_size_t,
                          P]),
    "ptv3_gemm_splits": (c_int, [c_int64, c_int, c_int, c_int, c_int]),
    "ptv3_block_fusable": (c_int, [c_int, c_int, c_int, c_int64]),
    "ptv3_rows_linear_capable": (c_int, [c_int, c_int, c_int, c_int64]),
    "ptv3_rows_linear": (c_int, [P, P, P, P, P, P, P, P, c_int, P, P, P, c_int64, c_int, c_int, c_float, c_int, P]),
    "ptv3_block_head": (c_int, [P, P, c_int, P, P, P, P, P, P, P, P, P, P, c_int64, c_int, c_float, c_int, P]),
    "ptv3_block_tail": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_float, c_int, P]),
    "ptv3_mlp2_fusable": (c_int, [c_int, c_int, c_int, c_int]),
    "ptv3_mlp2": (c_int, [P, P, P, P, P, c_int, P, P, P, c_int, c_int64, c_int, c_int, c_int, c_int, P]),
    "ptv3_layernorm": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_float, c_int, P]),
    "ptv3_layernorm_slabs": (c_int, [P, c_int, P, P, P, P, P, P, P, P, c_int64, c_int, c_float, c_int, P]),
    "ptv3_affine_act": (c_int, [P, P, P, c_int, P, c_int64, c_int, c_int, P]),
    "ptv3_cast": (c_int, [P, c_int, P, c_int, c_int64, P]),
    "ptv3_pool_workspace_bytes": (c_size_t, [c_int64]),
    "ptv3_pool_segments": (c_int, [P, P, c_int64, c_int, P, P, P, P, P, P, c_size_t, P]),
    "ptv3_pool_reduce": (c_int, [P, P, P, P, P, c_int, P, P, c_int64, c_int64, c_int, c_int, P, P, c_int, P, P, P,
                                 P, P, P, c_int, P]),
    "ptv3_forward_workspace_bytes": (c_size_t, [P, c_int64, c_int]),
    "ptv3_forward": (c_int, [P, P, c_int, P, P, c_size_t, P]),
    "ptv3_executor_create": (c_void_p, []),
    "ptv3_executor_destroy": (c_int, [P]),
    "ptv3_block_train_workspace_bytes": (c_size_t, [P, c_int]),
    "ptv3_block_train_fwd": (c_int, [P, P]),
    "ptv3_block_train_bwd": (c_int, [P, P]),
    "ptv3_gemm_tn_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "ptv3_gemm_tn": (c_int, [P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "ptv3_col_reduce_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "ptv3_col_reduce": (c_int, [P, P, P, P, c_float, c_int, P, c_int64, c_int, c_int, P, c_size_t, P]),
    "ptv3_bn_finalize": (c_int, [P, P, c_int64, P, P, P, P, c_float, c_float, P, P, P, P, c_int, P]),
    "ptv3_bn_bwd_coeffs": (c_int, [P, c_int64, P, P, P, P, P, P, c_int, P]),
    "ptv3_layernorm_bwd": (c_int, [P, P, P, P, c_float, P, P, c_int64, c_int, c_int, P, c_size_t, P]),
    "ptv3_act_bwd": (c_int, [P, P, P, P, c_int, P, c_int64, c_int, c_int, P]),
    "ptv3_affine2": (c_int, [P, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "ptv3_pool_max_bwd": (c_int, [P, P, P, P, c_int64, c_int, P, c_int, P]),
    "ptv3_segment_sum": (c_int, [P, P, P, c_int64, c_int, P, c_int, P]),
    "ptv3_window_attn_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int]),
    "ptv3_window_attn_bwd": (c_int, [P, P, P, P, P, P, c_int64, c_int64, c_int, c_int, c_int, c_float, c_int, P,
                                     c_size_t, P]),
    "ptv3_window_attn_varlen_bwd": (c_int, [P, P, P, P, P, P, c_int, P, c_int64, c_int64, c_int, c_int, c_int, c_float,
                                            c_int, P, c_size_t, P]),
    "ptv3_window_attn_train_fwd": (c_int, [P, P, P, P, c_int, P, P, c_int64, c_int64, c_int, c_int, c_int, c_float,
                                           c_double, c_int, P]),
    "ptv3_window_attn_train_bwd": (c_int, [P, P, P, P, P, P, P, c_int, P, c_int64, c_int64, c_int, c_int, c_int, c_float,
                                           c_int, P, c_size_t, P]),
    "ptv3_window_attn_drop_fwd": (c_int, [P, P, P, P, c_int, P, c_int64, c_int64, c_int, c_int, c_int, c_float, c_float,
                                          c_uint32, c_int, P]),
    "ptv3_window_attn_drop_bwd": (c_int, [P, P, P, P, P, P, c_int, P, c_int64, c_int64, c_int, c_int, c_int, c_float,
                                          c_float, c_uint32, c_int, P, c_size_t, P]),
    "ptv3_window_attn_rpe_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int, c_int, c_int]),
    "ptv3_window_attn_rpe_bwd": (c_int, [P, P, P, P, P, P, P, c_int, P, P, c_int64, c_int64, c_int, c_int, c_int,
                                         c_float, c_int, P, c_size_t, P]),
    "ptv3_adamw_entry_bytes": (c_size_t, []),
    "ptv3_adamw_chunk": (c_int, []),
    "ptv3_adamw_fill_entry": (c_int, [P, P, P, P, P, c_int64, c_int, c_int]),
    "ptv3_adamw_fill_shadow": (c_int, [P, P, P, c_int, c_int, c_int, c_int]),
    "ptv3_adamw_fill_step_lag": (c_int, [P, c_int64]),
    "ptv3_adamw_step": (c_int, [P, c_int, c_int, P, P, c_int, c_float, c_float, c_float, c_int64, c_float, P, P, c_int,
                                c_int, P]),
    "ptv3_adamw_shadow_tiles": (c_int, [c_int, c_int, c_int]),
    "ptv3_adamw_fill_first_tile": (c_int, [P, c_int]),
    "ptv3_grad_sqnorm": (c_int, [P, c_int, c_int, P, P, P, P, P]),
    "ptv3_grid_hash": (c_int, [P, c_int64, c_double, c_int, P, P, P, P]),
    "ptv3_keypoint_aggregate": (c_int, [P, P, P, c_int, c_int, P, P, c_int, c_float, P, P, P]),
    "ptv3_profile_enable": (c_int, [c_int]),
    "ptv3_profile_collect": (c_int, [P, P, P, P]),
    "ptv3_profile_hint_flops": (c_int, [ctypes.c_double]),
    "ptv3_profile_kernel_count": (c_int, []),
    "ptv3_profile_kernel_name": (c_char_p, [c_int]),
    "ptv3_profile_collect_kernels": (c_int, [P, P, P, P]),
    "ptv3_swin_window_keys": (c_int, [P, c_int64, c_int, c_int, c_int, P, P, P]),
    "ptv3_swin_attn_fwd": (c_int, [P, P, P, P, P, P, P, c_int, P, P, c_int, P, P, c_int64, c_int, c_int, c_int, c_int, P]),
    "ptv3_swin_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, P, P, c_int, P, P, P, P, P, P, P, c_int64, c_int, c_int,
                                   c_int, c_int, P]),
    "ptv3_knn_query": (c_int, [c_int, c_int, P, P, P, P, c_int, P, P, P]),
    "ptv3_knn_query_cells": (c_int, [c_int, c_int, P, P, P, P, c_int64, P, P, P, c_float, P, P, P]),
    "ptv3_grouping_forward": (c_int, [c_int, c_int, c_int, P, P, P, P]),
    "ptv3_grouping_backward": (c_int, [c_int, c_int, c_int, P, P, P, P]),
    "ptv3_interpolation_forward": (c_int, [c_int, c_int, c_int, P, P, P, P, P]),
    "ptv3_interpolation_backward": (c_int, [c_int, c_int, c_int, P, P, P, P, P]),
}


class _Lib:
    """Lazy handle: importing the package on a box without the .so works, using an op does not."""

    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            path = library_path()
            if not os.path.exists(path):
                raise ImportError(
                    f"libptv3_hip.so not found at {path}: build it with `make -C pointcept-keypointdetection_amd` "
                    "(or __graft_entry__.build()); there is no CPU fallback for the PTv3 HIP path")
            dll = ctypes.CDLL(path)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(dll, name)
                fn.restype, fn.argtypes = res, args
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        return getattr(self.load(), name)

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed (code {rc}): {self.load().ptv3_last_error().decode()}")


lib = _Lib()
