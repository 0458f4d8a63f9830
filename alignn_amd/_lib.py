"""ctypes binding of libalignn_hip.so (the C ABI declared in include/alignn_hip.h).

The product path has NO fallback: if the shared library is missing or a GPU call fails, an
exception is raised.  PyTorch appears here only as the owner of device memory and streams.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libalignn_hip.so")
LIB_PATH = os.environ.get("ALIGNN_AMD_LIB_PATH", LIB_PATH)  # (A/B runs of differently compiled libraries: tools/gpu/*.sh)

_p, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must mirror include/alignn_hip.h (tests/test_abi.py checks the set)
SIGNATURES = {
    "alignn_version": (C.c_char_p, []),
    "alignn_gemm_nt": (_i32, [_p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i32, _i32, _p]),
    "alignn_gemm_nn": (_i32, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _p]),
    "alignn_gemm_nn_split_workspace": (_sz, [_i64, _i32, _i32]),
    "alignn_gemm_nn_split": (_i32, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _p, _sz, _p]),
    "alignn_gemm_tn_workspace": (_sz, [_i64, _i32, _i32]),
    "alignn_gemm_tn": (_i32, [_p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _i32, _i32, _p, _sz, _p]),
    "alignn_split_bf16x3_bytes": (_sz, [_i32, _i32]),
    "alignn_split_bf16x3": (_i32, [_p, _i64, _i32, _i32, _i32, _p, _p]),
    "alignn_gemm_nt_x6_supported": (_i32, [_i64, _i32, _i32]),
    "alignn_gemm_nt_x6": (_i32, [_p, _i64, _p, _p, _p, _i64, _p, _i64, _i64, _i32, _i32, _p]),
    "alignn_gemm_tn_x6_supported": (_i32, [_i64, _i32, _i32]),
    "alignn_gemm_tn_x6_workspace": (_sz, [_i64, _i32, _i32]),
    "alignn_gemm_tn_x6_splits": (_i32, [_i64, _i32, _i32]),
    "alignn_gemm_tn_x6_partials": (_i32, [_p, _i64, _p, _p, _i64, _p, _i64, _i32, _i32, _p, _sz, _p]),
    "alignn_absmax": (_i32, [_p, _i64, _i64, _i32, _p, _p]),
    "alignn_split_f16x2_bytes": (_sz, [_i32, _i32]),
    "alignn_split_f16x2": (_i32, [_p, _i64, _i32, _i32, _i32, _p, _p, _p]),
    "alignn_split_f16x2_both": (_i32, [_p, _i64, _i32, _i32, _p, _p, _p, _p]),
    "alignn_prepare_weights": (_i32, [_p, _i32, _p, _p]),
    "alignn_absmax_raise": (_i32, [_p, _i64, _i64, _i32, _p, _p]),
    "alignn_gemm_nt_f16x3": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i32, _i32, _p]),
    "alignn_gemm_nt_f16x3_gather": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i64, _p, _p, _p, _p]),
    "alignn_gemm_nt_f16x3_gather2": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i64, _p, _p, _i64, _p, _p, _p]),
    "alignn_gather_rows_ld": (_i32, [_p, _i64, _p, _p, _i64, _i64, _i32, _p]),
    "alignn_gemm_nt_f16x3_stats": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _p]),
    "alignn_egc_gate_fwd_pre_norm": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "alignn_egc_gate_fwd_pre": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_gemm_nt_x6_row_tiles": (_i32, [_i64, _i32, _i32]),
    "alignn_gemm_nt_f16x3_bnred": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _i64, _i32, _i32, _p, _i64, _p, _p, _p]),
    "alignn_gemm_dgrad_wgrad_supported": (_i32, [_i64, _i32, _i32]),
    "alignn_gemm_dgrad_wgrad_slabs": (_i32, [_i64]),
    "alignn_gemm_dgrad_wgrad_workspace": (_sz, [_i64]),
    "alignn_gemm_dgrad_wgrad_f16x3": (_i32, [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64,
                                              _p, _sz, _p]),
    "alignn_dual_slabs": (_i32, [_i64]),
    "alignn_ln_silu_dual_fwd": (_i32, [_p, _p, _i64, _p, _p, _i64, _p, _p, _f32, _p, _p, _i64, _p, _i64, _i32, _p, _p]),
    "alignn_ln_silu_dual_bwd": (_i32, [_p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _i32, _p, _p]),
    "alignn_ln_silu_dual_bwd_node": (_i32, [_p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i64, _i32, _p] + [_p] * 8 + [_p]),
    "alignn_egc_gate_dual_fwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "alignn_egc_gate_dual_fwd_tangent": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_egc_node_dual_bwd": (_i32, [_p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "alignn_egc_dual_bwd_dst": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "alignn_egc_dual_bwd_src": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p]),
    "alignn_egc_dual_bwd_lg_dense": (_i32, [_p] * 10 + [_i64, _p, _p, _i64, _p, _p, _i32] + [_p] * 7 + [_p]),
    "alignn_egc_ln_fused_supported": (_i32, [_i32, _i64]),
    "alignn_egc_ln_dst_slabs": (_i32, [_i64]),
    "alignn_egc_ln_dst_supported": (_i32, [_i32]),
    "alignn_egc_bwd_dst_ln": (_i32, [_p] * 11 + [_i64, _i32] + [_p] * 6 + [_p]),
    "alignn_egc_dual_bwd_dst_ln": (_i32, [_p] * 16 + [_i64, _i32] + [_p] * 8 + [_p]),
    "alignn_egc_gate_fwd_pre_ln": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _f32, _p, _p, _p, _p, _p]),
    "alignn_egc_bwd_lg_dense_ln": (_i32, [_p] * 8 + [_i64, _p, _p, _i64, _i32, _p, _p, _i32] + [_p] * 6 + [_p]),
    "alignn_egc_gate_dual_tan_ln": (_i32, [_p] * 7 + [_i64, _i64, _i32] + [_p] * 11 + [_p]),
    "alignn_egc_dual_bwd_lg_dense_ln": (_i32, [_p] * 13 + [_i64, _p, _p, _i64, _p, _p, _i32] + [_p] * 8 + [_p]),
    "alignn_slab_fold_slabs": (_i32, []),
    "alignn_slab_fold": (_i32, [_p, _i32, _i32, _p, _p]),
    "alignn_col_stats_slabs": (_i32, [_i64]),
    "alignn_col_stats": (_i32, [_p, _i64, _i64, _i32, _p, _p]),
    "alignn_col_stats_welford": (_i32, [_p, _i64, _i64, _i32, _p, _p]),
    "alignn_bn_finalize_welford": (_i32, [_p, _i32, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _p]),
    "alignn_col_sum": (_i32, [_p, _i64, _i64, _i32, _p, _p, _p]),
    "alignn_bn_finalize": (_i32, [_p, _i32, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _p]),
    "alignn_bn_silu_fwd": (_i32, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i32, _p, _p]),
    "alignn_bn_silu_bwd_reduce": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i32, _p, _p]),
    "alignn_bn_bwd_finalize": (_i32, [_p, _i32, _i32, _p, _p]),
    "alignn_bn_silu_bwd_apply": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _i32, _p, _i64, _i64, _i32, _p, _p]),
    "alignn_bn_silu_bwd_apply_node": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _i32, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_bn_silu_bwd_apply_sum": (_i32, [_p, _i64, _p, _i64, _p, _p, _i32, _p, _i64, _i64, _i32, _p, _p, _p]),
    "alignn_egc_slabs": (_i32, [_i64]),
    "alignn_egc_gate_fwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_egc_node_bwd": (_i32, [_p, _i64, _p, _p, _p, _p, _i64, _i32, _p]),
    "alignn_egc_bwd_dst": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_slab_sum": (_i32, [_p, _i32, _i32, _p, _p]),
    "alignn_egc_bwd_lg_fused": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i32, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_ln_slabs": (_i32, [_i64]),
    "alignn_ln_silu_fwd": (_i32, [_p, _i64, _p, _i64, _p, _p, _f32, _p, _i64, _p, _i64, _i32, _p, _p]),
    "alignn_ln_silu_bwd": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _i32, _p, _p]),
    "alignn_ln_silu_bwd_node": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_bond_cosine_fwd": (_i32, [_p, _p, _p, _p, _i64, _p]),
    "alignn_egc_gate_infer": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_rbf_bwd": (_i32, [_p, _p, _f32, _p, _p, _i64, _i32, _p]),
    "alignn_norm3_bwd": (_i32, [_p, _p, _p, _i64, _p]),
    "alignn_bond_cosine_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _p]),
    "alignn_egc_bwd_lg_dense_supported": (_i32, [_i32]),
    "alignn_egc_bwd_lg_dense": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i32, _i64, _p, _p, _i64, _i32, _p, _p, _i32, _p, _p, _p, _p, _p, _p]),
    "alignn_egc_bwd_src": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p]),
    "alignn_rbf_fwd": (_i32, [_p, _p, _f32, _p, _i64, _i32, _p]),
    "alignn_norm3_fwd": (_i32, [_p, _p, _i64, _p]),
    "alignn_segment_mean_fwd": (_i32, [_p, _p, _p, _i32, _i32, _p]),
    "alignn_segment_mean_bwd": (_i32, [_p, _p, _p, _i32, _i32, _p]),
    "alignn_gather_rows": (_i32, [_p, _p, _p, _i64, _i32, _p]),
    "alignn_segment_sum": (_i32, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _i32, _p]),
    "alignn_egc_conv_fwd_scratch": (_sz, [_i64, _i64, _i32, _i32, _i32]),
    "alignn_egc_conv_fwd": (_i32, [_p, _p]),
    "alignn_egc_conv_bwd_scratch": (_sz, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "alignn_egc_conv_bwd": (_i32, [_p, _p]),
    "alignn_egc_conv_wgrad_scratch": (_sz, [_i64, _i64, _i32, _i32]),
    "alignn_egc_conv_wgrad": (_i32, [_p, _p]),
    "alignn_egc_args_sizeof": (_sz, [_i32]),
    "alignn_fork_events_init": (_i32, []),
    "alignn_knn_levels": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i64, _p, _p]),
    "alignn_knn_kth": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i64, _p, _p, _p]),
    "alignn_knn_count": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i64, _p, _p, _p, _p]),
    "alignn_radius_levels": (_i32, [_p, _p, _p, _p, _p, _f32, _i32, _i32, _p, _p]),
    "alignn_radius_count": (_i32, [_p, _p, _p, _p, _p, _p, _f32, _i32, _i64, _p, _p, _p]),
    "alignn_radius_emit": (_i32, [_p, _p, _p, _p, _p, _p, _f32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "alignn_stage_batch_workspace": (_sz, [_i64, _i64]),
    "alignn_stage_batch": (_i32, [_p, _p, _p, _i64, _i64, _i64] + [_p] * 17 + [_p, _sz, _p]),
    "alignn_map_line_graph_rows": (_i32, [_p] * 8 + [_i64, _i64, _p, _p, _p, _p]),
    "alignn_model_init": (_i32, []),
    "alignn_model_sizeof": (_sz, [_i32]),
    "alignn_angle_embed_supported": (_i32, [_i32, _i32, _i32]),
    "alignn_angle_embed_workspace": (_sz, [_i64, _i32, _i32]),
    "alignn_angle_args_sizeof": (_sz, []),
    "alignn_angle_embed_scal_floats": (_i32, []),
    "alignn_angle_embed_fwd": (_i32, [_p, _p]),
    "alignn_angle_embed_bwd": (_i32, [_p, _p]),
    "alignn_angle_embed_infer": (_i32, [_p, _p]),
    "alignn_model_plan": (_i32, [_p, _p, _p, _p]),
    "alignn_model_fwd": (_i32, [_p, _p, _p, _sz, _p, _p]),
    "alignn_model_bwd": (_i32, [_p, _p, _p, _sz, _p, _p]),
    "alignn_model_infer_workspace": (_sz, [_p, _p]),
    "alignn_model_infer": (_i32, [_p, _p, _p, _sz, _p, _p]),
    "alignn_knn_emit": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    # whole-model force field (csrc/model.hip) and the small kernels of its head (csrc/ff.hip)
    "alignn_ff_desc_sizeof": (_sz, []),
    "alignn_debug_allocs": (_i32, [_p, _i32]),
    "alignn_ff_plan": (_i32, [_p, _p, _p, _p, _p]),
    "alignn_ff_eval": (_i32, [_p, _p, _p, _p, _sz, _p, _p, _p, _p]),
    "alignn_ff_grad": (_i32, [_p, _p, _p, _p, _sz, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _p]),
    "alignn_pair_force_reduce": (_i32, [_p, _f32, _p, _p, _p, _i32, _p, _i64, _p]),
    "alignn_virial_stress": (_i32, [_p, _p, _f32, _p, _p, _p, _f32, _p, _i32, _p]),
    "alignn_ff_energy": (_i32, [_p, _p, _p, _i32, _i64, _i32, _i32, _f32, _f32, _p, _p, _p]),
    "alignn_ff_penalty_bwd": (_i32, [_p, _p, _i64, _i32, _f32, _f32, _p]),
    "alignn_ff_pair_weights": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _f32, _i32, _i32, _i64, _p, _p, _p]),
    "alignn_ff_tangent_geometry": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _p]),
    "alignn_rbf_tangent": (_i32, [_p, _p, _p, _f32, _p, _i64, _i32, _p]),
    "alignn_bond_cosine_tangent": (_i32, [_p, _p, _p, _p, _p, _i64, _p]),
    "alignn_ff_readout_seed": (_i32, [_p, _f32, _i32, _p, _p, _p, _p, _p, _i32, _i64, _i32, _p]),
    "alignn_ff_fc_grad": (_i32, [_p, _f32, _i32, _p, _p, _p, _p, _p, _p, _i32, _i32, _p]),
    "alignn_add_inplace": (_i32, [_p, _p, _i64, _p]),
    "alignn_add3": (_i32, [_p, _p, _p, _p, _i64, _p]),
}

# argument blocks of the composite entry points (include/alignn_hip.h: alignn_egc_fwd_args / _bwd_args / _wgrad_args), packed
# with the platform's C layout rules ("@": native sizes and alignment); load() checks the sizes against the library's
import struct as _struct

EGC_FWD_ARGS = _struct.Struct("@5P2q6i2f32PN")
EGC_BWD_ARGS = _struct.Struct("@8P3q8i22Pq14PN")
EGC_WGRAD_ARGS = _struct.Struct("@2q4i14PN")

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built - there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m alignn_amd.build` "
                "(hipcc --offload-arch=gfx950); alignn_amd has no CPU/eager fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        for which, st in enumerate((EGC_FWD_ARGS, EGC_BWD_ARGS, EGC_WGRAD_ARGS)):
            if lib.alignn_egc_args_sizeof(which) != st.size:
                raise RuntimeError(f"argument block {which} of the composite entry points: library says "
                                   f"{lib.alignn_egc_args_sizeof(which)} bytes, the binding packs {st.size}")
        _lib = lib
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw handle of torch's current HIP stream (what every launch goes to).  ``torch.cuda.current_stream()`` builds a
    Python Stream object per call (~8 us, ~650 calls per training step); the raw getter is a plain C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(t):
    """Every launch goes to the CURRENT device's current stream (``stream()``): make the tensors' device current for
    the duration of a module forward when it is not already (a model living on a non-current GPU).  The autograd
    engine replays backward nodes under the device of their forward, so guarding the forward covers both."""
    if t.is_cuda and t.device.index != torch.cuda.current_device():
        return torch.cuda.device(t.device)
    return _NO_GUARD


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"libalignn_hip: {what} failed with hipError {rc}")


def require_f32(*tensors):
    for t in tensors:
        if t is None:
            continue
        if t.dtype != torch.float32 or not t.is_cuda:
            raise TypeError(
                f"alignn_amd kernels take float32 CUDA(HIP) tensors, got {t.dtype} on {t.device}; "
                "there is no CPU fallback"
            )
