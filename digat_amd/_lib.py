"""ctypes binding of ``libdigat_hip.so`` (the C ABI of include/digat_hip.h).

The library is built in-tree by ``digat_amd/build.py`` (``hipcc --offload-arch=gfx950``).  There is
no CPU fallback: if the shared object is missing, or a call returns a non-zero status, this module
raises.  PyTorch only supplies device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIGAT_HIP_LIB: another build of the same ABI (A/B measurements of two kernel variants in one run)
LIB_PATH = os.environ.get("DIGAT_HIP_LIB") or os.path.join(_HERE, "lib", "libdigat_hip.so")
DIGAT_MAX_DEPTH = 16
DIGAT_MAX_NODES = 128
GEMM_BF16X6, GEMM_F16X3 = 0, 1          # operand format of a split weight image (include/digat_hip.h)
PARAMS_GEMM_F16X3 = 64                  # digat_params.flags: the block's wsplit images are GEMM_F16X3
PARAMS_BD_TILED = 128                   # digat_params.flags: the [B,d] linears run on the tiled kernel at every row count
PARAMS_PQ_FP8 = 256                     # digat_params.flags: P', Q of Eq. 8 stored as block-scaled e4m3 (DIGAT_PQ_FP8)
PARAMS_SIDE_STREAM_OFF, PARAMS_SIDE_STREAM_ON = 512, 1024      # digat_params.flags: never / always (neither: by pass size)
PARAMS_NO_LIVE_ROWS = 2048              # digat_params.flags: project every user-graph node in every layer

_f = C.c_void_p  # every device pointer crosses as void*


class LayerParams(C.Structure):
    _fields_ = [(k, _f) for k in ("W", "bW", "F1", "F2", "F3", "b3", "a", "wsplit", "f3_wsplit")]


class MsaParams(C.Structure):
    """digat_msa_params (include/digat_hip.h)."""
    _fields_ = ([(k, C.c_int32) for k in ("word_embedding_dim", "head_num", "head_dim", "attention_dim")]
                + [(k, C.c_void_p) for k in ("word_embedding", "W_Q", "b_Q", "W_K", "W_V", "b_V", "A1", "b1", "a2",
                                             "qkv_wsplit", "a1_wsplit")])


class SplitJob(C.Structure):
    """digat_split_job (include/digat_hip.h)."""
    _fields_ = [("w0", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("layout", C.c_int32),
                ("reserved", C.c_int32), ("image", C.c_void_p)]


class GatherJob(C.Structure):
    """digat_gather_job (include/digat_hip.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64), ("rows", C.c_int64),
                ("idx", C.c_void_p), ("idx2", C.c_void_p), ("inner", C.c_int64)]


class Params(C.Structure):
    _fields_ = ([("d", C.c_int32), ("depth", C.c_int32), ("category_num", C.c_int32), ("flags", C.c_int32)]
                + [(k, _f) for k in ("topic_node_embedding", "cand_K", "cand_Q", "cand_bQ",
                                     "news_graph_W", "news_graph_b", "user_news_K", "user_news_Q",
                                     "user_news_bQ", "featureAffine_W", "featureAffine_b",
                                     "userAtt_K", "userAtt_Q", "userAtt_bQ")]
                + [("news", LayerParams * DIGAT_MAX_DEPTH), ("user", LayerParams * DIGAT_MAX_DEPTH)]
                + [(k, _f) for k in ("cand_fold_W", "cand_fold_b", "user_news_fold_W", "user_news_fold_b",
                                     "userAtt_fold_W", "userAtt_fold_b", "featureAffine_wsplit", "cand_fold_wsplit", "gate_wsplit")]
                + [("ctx_wsplit", _f * (DIGAT_MAX_DEPTH + 1)), ("range_flag", _f), ("featureAffine_fsplit", _f)])


class DigatHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

_SIGNATURES = {
    "digat_version": (C.c_int, []),
    "digat_error_string": (C.c_char_p, [C.c_int]),
    "digat_linear_f32": (C.c_int, [_f, C.c_int64, _f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, _f]),
    "digat_xattn_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_xattn_fwd": (C.c_int, [_f] * 12 + [C.c_int] * 3 + [_f, C.c_size_t, _f]),
    "digat_xattn_fwd_mode": (C.c_int, [_f] * 11 + [C.c_int] * 4 + [_f, C.c_size_t, _f]),
    "digat_xattn_fwd_lowprec": (C.c_int, [_f] * 11 + [C.c_int, _f] + [C.c_int] * 4 + [_f, C.c_size_t, _f]),
    "digat_xattn_pairwise_fwd": (C.c_int, [_f] * 8 + [C.c_int] * 3 + [_f]),
    "digat_news_ctx_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_news_ctx_fwd": (C.c_int, [_f] * 9 + [C.c_int] * 3 + [_f, C.c_size_t, _f]),
    "digat_user_ctx_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "digat_user_ctx_fwd": (C.c_int, [_f] * 14 + [C.c_int] * 5 + [_f, C.c_size_t, _f]),
    "digat_topic_pool_fwd": (C.c_int, [_f] * 4 + [C.c_int] * 5 + [_f]),
    "digat_encoder_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "digat_encoder_fwd": (C.c_int, [C.POINTER(Params)] + [_f] * 10 + [C.c_int] * 3 + [_f, C.c_size_t, _f]),
    "digat_row_logits": (C.c_int, [_f] * 3 + [C.c_int] * 2 + [_f]),
    "digat_profile_pause": (C.c_int, [C.c_int]),
    "digat_set_train_precision": (C.c_int, [C.c_int]),
    "digat_gather_tables": (C.c_int, [C.POINTER(GatherJob), C.c_int, _f]),
    "digat_profile_live_row_fraction": (C.c_double, []),
    "digat_rank_metrics": (C.c_int, [_f] * 3 + [C.c_int] + [_f] * 4),
    "digat_format_rank_file": (C.c_int64, [_f, _f, C.c_int64, _f, C.c_int64]),
    "digat_gat_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_gat_fwd": (C.c_int, [_f] * 7 + [C.c_int] * 3 + [_f, C.c_size_t, _f]),
    "digat_gat_train_save_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_gat_train_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_gat_fwd_train": (C.c_int, [_f] * 7 + [C.c_float, C.c_uint32] + [C.c_int] * 3 + [_f, C.c_size_t, _f, C.c_size_t, _f]),
    "digat_gat_bwd": (C.c_int, [_f] * 7 + [C.c_float, _f, C.c_size_t] + [_f] * 5 + [C.c_int] * 3 + [_f, C.c_size_t, _f]),
    "digat_sag_cos_topk_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int]),
    "digat_sag_cos_topk": (C.c_int, [_f, _f, C.c_int64, _f, _f, C.c_int64, C.c_int, C.c_int, _f, _f, _f, C.c_size_t, _f]),
    "digat_sag_news_graph": (C.c_int, [_f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float] + [_f] * 5),
    "digat_msa_split_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_split_msa_weights": (C.c_int, [_f] * 3 + [C.c_int] * 2 + [_f, _f]),
    "digat_msa_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "digat_msa_fwd": (C.c_int, [C.POINTER(MsaParams), _f, _f, _f, C.c_int, C.c_int, _f, C.c_size_t, _f]),
    "digat_msa_train_save_bytes": (C.c_size_t, [C.c_int] * 6),
    "digat_msa_train_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "digat_msa_fwd_train": (C.c_int, [C.POINTER(MsaParams), _f, _f, _f, C.c_float, C.c_uint32, C.c_int, C.c_int, _f, C.c_size_t, _f,
                                      C.c_size_t, _f]),
    "digat_msa_bwd": (C.c_int, [C.POINTER(MsaParams), _f, _f, _f, C.c_float, _f, C.c_size_t, _f, C.c_int64] + [_f] * 8
                      + [C.c_int, C.c_int, _f, C.c_size_t, _f]),
    "digat_msa_row_grad_ld": (C.c_int64, [C.c_int] * 3),
    "digat_embedding_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "digat_embedding_bwd": (C.c_int, [_f, C.c_int64, _f, _f, C.c_int64, C.c_int, _f, _f, C.c_size_t, _f]),
    "digat_embedding_bwd_unsorted_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int64]),
    "digat_embedding_bwd_unsorted": (C.c_int, [_f, _f, C.c_int64, C.c_int64, _f, _f, C.c_int64, C.c_int64, C.c_int, C.c_int64, _f, _f, C.c_size_t,
                                               _f]),
    "digat_encoder_grouped_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "digat_encoder_fwd_grouped": (C.c_int, [C.POINTER(Params)] + [_f] * 11 + [C.c_int] * 4 + [_f, C.c_size_t, _f]),
    "digat_encoder_fwd_grouped_cached": (C.c_int, [C.POINTER(Params)] + [_f] * 14 + [C.c_int64] + [_f] * 2 + [C.c_int] * 4 + [_f, C.c_size_t, _f]),
    "digat_news_context_queries": (C.c_int, [C.POINTER(Params), _f, _f, C.c_int, _f]),
    "digat_user_project0": (C.c_int, [C.POINTER(Params), _f, _f, C.c_int, _f]),
    "digat_news_project0": (C.c_int, [C.POINTER(Params), _f, _f, C.c_int, C.c_int, _f]),
    "digat_split_weights_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "digat_forget_split_image": (C.c_int, [_f]),
    "digat_split_proj_weights": (C.c_int, [_f, _f, _f, C.c_int, _f, C.c_int, _f]),
    "digat_split_weights": (C.c_int, [_f, C.c_int, C.c_int, _f, C.c_int, _f]),
    "digat_linear_f32x3": (C.c_int, [_f, C.c_int64, _f, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f]),
    "digat_fold_workspace_bytes": (C.c_size_t, [C.c_int]),
    "digat_fold_attention": (C.c_int, [_f] * 5 + [C.c_int, _f, C.c_size_t, _f]),
    "digat_linear_bwd_input": (C.c_int, [_f, C.c_int64, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _f]),
    "digat_linear_bwd_input_x3": (C.c_int, [_f, C.c_int64, _f, _f, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _f, _f]),
    "digat_linear_bwd_weight_workspace": (C.c_size_t, [C.c_int] * 3),
    "digat_linear_bwd_weight": (C.c_int, [_f, C.c_int64, _f, C.c_int64, _f, _f, C.c_int, C.c_int, C.c_int, C.c_int, _f,
                                          C.c_size_t, _f]),
    "digat_colsum": (C.c_int, [_f, C.c_int64, _f, C.c_int, C.c_int, C.c_int, _f]),
    "digat_dropout_fwd": (C.c_int, [_f, _f, _f, C.c_int64, C.c_float, C.c_uint32, _f]),
    "digat_dropout_bwd": (C.c_int, [_f, _f, _f, C.c_int64, C.c_float, _f]),
    "digat_gate_fwd": (C.c_int, [_f, _f, C.c_int64, _f, _f, C.c_int, C.c_int, _f]),
    "digat_gate_bwd": (C.c_int, [_f, _f, _f, C.c_int64, _f, _f, _f, _f, C.c_int, C.c_int, _f]),
    "digat_relu_res_fwd": (C.c_int, [_f, _f, _f, C.c_int64, _f]),
    "digat_relu_mask": (C.c_int, [_f, _f, _f, C.c_int64, _f]),
    "digat_attn_pool_fwd": (C.c_int, [_f, C.c_int64, _f, _f, _f, _f, C.c_int, C.c_int, C.c_int, _f]),
    "digat_attn_pool_bwd": (C.c_int, [_f, C.c_int64, _f, _f, _f, _f, _f, C.c_int64, _f, C.c_int, C.c_int, C.c_int,
                                      C.c_int, _f]),
    "digat_topic_pool_fwd_train": (C.c_int, [_f] * 5 + [C.c_int] * 5 + [_f]),
    "digat_topic_pool_bwd": (C.c_int, [_f] * 7 + [C.c_int] * 5 + [_f]),
    "digat_xattn_project": (C.c_int, [_f] * 9 + [C.c_int] * 3 + [_f]),
    "digat_xattn_project_x3": (C.c_int, [_f] * 9 + [C.c_int] * 3 + [_f, _f]),
    "digat_xattn_pairwise_fwd_train": (C.c_int, [_f] * 11 + [C.c_float, C.c_uint32, C.c_int, C.c_int, C.c_int, _f]),
    "digat_xattn_pairwise_bwd_workspace": (C.c_size_t, [C.c_int] * 3),
    "digat_xattn_pairwise_bwd": (C.c_int, [_f] * 11 + [C.c_float] + [_f] * 4 + [C.c_int] * 4 + [_f, C.c_size_t, _f]),
    "digat_sum_nodes": (C.c_int, [_f, _f, C.c_int, C.c_int, C.c_int, _f]),
    "digat_xattn_train_save_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_xattn_train_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_xattn_fwd_train": (C.c_int, [_f] * 11 + [C.c_float, C.c_uint32, C.c_float, C.c_uint32] + [C.c_int] * 3 + [_f, C.c_size_t, _f, C.c_size_t, _f, C.c_int,
                                        _f]),
    "digat_xattn_bwd": (C.c_int, [_f] * 10 + [C.c_float, C.c_float, _f, C.c_size_t] + [_f] * 9 + [C.c_int] * 3 + [_f, C.c_size_t, _f, C.c_int, _f]),
    "digat_opt_chunk": (C.c_int, []),
    "digat_clip_adam_step": (C.c_int, [_f, _f, _f, C.c_int, _f] + [C.c_float] * 5 + [C.c_int, _f]),
    "digat_row_logits_bwd": (C.c_int, [_f] * 5 + [C.c_int, C.c_int, _f]),
    "digat_click_loss": (C.c_int, [_f, C.c_int, C.c_int, _f, _f, _f]),
    "digat_split_job_bytes": (C.c_size_t, [C.c_int] * 4),
    "digat_split_jobs": (C.c_int, [C.POINTER(SplitJob), C.c_int, _f]),
    "digat_news_ctx_train_save_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_news_ctx_train_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "digat_news_ctx_fwd_train": (C.c_int, [_f] * 8 + [C.c_float, C.c_uint32] + [C.c_int] * 3 + [_f, C.c_size_t, _f, C.c_size_t, _f, _f]),
    "digat_news_ctx_bwd": (C.c_int, [_f] * 6 + [C.c_float, _f, C.c_size_t] + [_f] * 6 + [C.c_int] * 4 + [_f, C.c_size_t, _f]),
    "digat_user_ctx_train_save_bytes": (C.c_size_t, [C.c_int] * 5),
    "digat_user_ctx_train_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "digat_user_ctx_fwd_train": (C.c_int, [_f] * 13 + [C.c_float, C.c_uint32] + [C.c_int] * 5 + [_f, C.c_size_t, _f, C.c_size_t, _f, _f, _f]),
    "digat_user_ctx_bwd": (C.c_int, [_f] * 10 + [C.c_float, _f, C.c_size_t] + [_f] * 10 + [C.c_int] * 6 + [_f, C.c_size_t, _f, _f]),
    "digat_profile_start": (C.c_int, [C.c_int]),
    "digat_profile_stop": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "digat_profile_set_kinds": (C.c_int, [C.c_uint]),
    "digat_profile_gemm_bytes": (C.c_int, [C.POINTER(C.c_double)]),
    "digat_profile_marker": (C.c_int, [C.c_int, _f]),
    "digat_encoder_shared_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "digat_encoder_fwd_shared": (C.c_int, [C.POINTER(Params)] + [_f] * 10 + [C.c_int] * 3 + [_f, C.c_size_t, _f]),
    "digat_user_row_runs": (C.c_int, [_f] * 4 + [C.c_int] * 5 + [_f] * 4 + [C.c_size_t, _f]),
    "digat_profile_xattn_parts": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "digat_split_ctx_fused_bytes": (C.c_size_t, [C.c_int]),
    "digat_split_ctx_fused_weights": (C.c_int, [_f, C.c_int, _f, _f]),
}
_LAB_SIGNATURES = {"digat_set_staged_xattn": (C.c_int, [C.c_int])}
KERNEL_KINDS = ("proj", "linear", "xattn", "pool", "topic", "glue", "agg")
XATTN_PARTS = ("twin", "l0", "news", "other")      # digat_profile_xattn_parts: the Eq. 8 launches by kernel
EXPORTED = tuple(_SIGNATURES)
ABI_VERSION = 4          # include/digat_hip.h: DIGAT_ABI_VERSION the signature table above was written for


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DigatHipError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m digat_amd.build` or __graft_entry__.build()). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        handle.digat_version.restype, handle.digat_version.argtypes = C.c_int, []
        got = handle.digat_version()
        if got != ABI_VERSION:
            # an older or newer build (DIGAT_HIP_LIB, a stale .so): its entry points may take other argument lists — refuse it rather
            # than call through shifted arguments
            raise DigatHipError(f"{LIB_PATH} reports ABI version {got}, this loader's signature table is for version {ABI_VERSION}: "
                                "rebuild it from this tree (python -m digat_amd.build)")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in _LAB_SIGNATURES.items():          # LAB builds only (-DDIGAT_LAB through DIGAT_HIP_LIB)
            if hasattr(handle, name):
                fn = getattr(handle, name)
                fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


EXT_PATH = os.path.join(_HERE, "lib", "digat_torch_ext.so")
USE_TORCH_EXT = os.environ.get("DIGAT_TORCH_EXT", "1") != "0"     # 0: bind the hot entry points through ctypes too (A/B, tests)
_ext = None


def ext():
    """The thin torch extension over the same C ABI (csrc/digat_torch_ext.cpp: tensors in, raw pointers + the current HIP stream
    out), or None when it has not been built or is switched off — the ctypes table above then binds the same entry points.  A
    different library build named by DIGAT_HIP_LIB is only reachable through ctypes (the extension is linked to lib/libdigat_hip.so)."""
    global _ext
    if not USE_TORCH_EXT or os.environ.get("DIGAT_HIP_LIB"):
        return None
    if _ext is None:
        if not os.path.exists(EXT_PATH):
            _ext = False
        else:
            import importlib.util
            lib()                                   # libdigat_hip.so first: the extension's rpath finds the same file
            spec = importlib.util.spec_from_file_location("digat_torch_ext", EXT_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != lib().digat_version():
                raise DigatHipError("digat_torch_ext.so and libdigat_hip.so disagree about the ABI version: rebuild (python -m digat_amd.build)")
            _ext = mod
    return _ext or None


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().digat_error_string(code).decode()
        raise DigatHipError(f"{what} failed: [{code}] {msg}")


def addressof(struct) -> int:
    """Address of a ctypes structure (the parameter block handed to the torch extension as an integer)."""
    return C.addressof(struct)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = tensors[0].device
    if dev.type != "cuda":
        raise DigatHipError("digat_amd runs on the GPU only (got a %s tensor); there is no CPU path" % dev.type)
    for t in tensors:
        if t.device != dev:
            raise DigatHipError("all tensors must live on the same device")
    return dev


def f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise DigatHipError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def as_bytes(t: torch.Tensor) -> torch.Tensor:
    """bool / uint8 mask -> contiguous one-byte-per-element view (no copy for torch.bool)."""
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype == torch.uint8:
        return t.contiguous()
    return (t != 0).contiguous().view(torch.uint8)


def profile_start(max_launches: int = 1 << 16) -> None:
    check(lib().digat_profile_start(max_launches), "digat_profile_start")


def profile_stop():
    """-> {kind: {"ms": total ms, "work": flops or bytes, "launches": n}}"""
    n = len(KERNEL_KINDS)
    ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int * n)()
    check(lib().digat_profile_stop(ms, work, cnt), "digat_profile_stop")
    gb = (C.c_double * n)()
    check(lib().digat_profile_gemm_bytes(gb), "digat_profile_gemm_bytes")
    out = {k: {"ms": ms[i], "work": work[i], "launches": cnt[i], "gemm_bytes": gb[i]} for i, k in enumerate(KERNEL_KINDS)}
    m = len(XATTN_PARTS)
    pms, pby, pcnt = (C.c_double * m)(), (C.c_double * m)(), (C.c_int * m)()
    check(lib().digat_profile_xattn_parts(pms, pby, pcnt), "digat_profile_xattn_parts")
    out["xattn"]["parts"] = {k: {"ms": pms[i], "work": pby[i], "launches": pcnt[i]} for i, k in enumerate(XATTN_PARTS)}
    return out


def split_buffer(nbytes: int, device: torch.device) -> torch.Tensor:
    """Device bytes for a split weight image.  The library keeps the format of every image it has split by ADDRESS; when this
    buffer dies the address is forgotten (digat_forget_split_image), so that a later allocation at the same address — an image
    copied there, say — is not held to a stale format."""
    import weakref
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if device.type == "cuda":
        weakref.finalize(buf, lib().digat_forget_split_image, C.c_void_p(buf.data_ptr())).atexit = False
    return buf


_workspaces = {}


def workspace(nbytes: int, device: torch.device, tag: str = "") -> torch.Tensor:
    """A cached scratch buffer per (device, stream, tag), grown on demand.  Kernels launched on one stream
    run in order, so reusing it across calls is safe; calls on different streams (util.score_rows alternates two so
    that one batch's prologue runs under the previous batch's last layer) get different buffers."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
