"""ctypes binding of the C-ABI library (include/tokenreduction_hip.h).

The library is the product: there is NO CPU / PyTorch-eager fallback.  If the shared object is
missing or a symbol is absent, importing the ops raises -- loudly -- with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TOKENREDUCTION_HIP_LIB: load another build of the same library (kernel experiments, tools/attn_lab.py); same ABI required
LIB_PATH = os.environ.get("TOKENREDUCTION_HIP_LIB") or os.path.join(_HERE, "csrc", "libtokenreduction_hip.so")

TR_MAX_DEPTH = 32
TR_EPI_BF16, TR_EPI_GELU_BF16, TR_EPI_RESID_F32, TR_EPI_F32, TR_EPI_PATCH_F32 = 0, 1, 2, 3, 4
TR_FAMILY_DEIT, TR_FAMILY_TOPK, TR_FAMILY_EVIT, TR_FAMILY_TOME, TR_FAMILY_DYVIT, TR_FAMILY_SIT, \
    TR_FAMILY_DPCKNN, TR_FAMILY_ATS, TR_FAMILY_SINKHORN, TR_FAMILY_KMEDOIDS, \
    TR_FAMILY_PATCHMERGER, TR_FAMILY_HEURISTIC = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
TR_PREC_BF16, TR_PREC_FP32, TR_PREC_BF16X3 = 0, 1, 2

_vp, _i, _f, _l, _sz = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_size_t


class TrBlockWeights(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                   "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "mlp_pk")]


class TrStageWeights(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln_g", "ln_b", "w0", "b0", "w1", "b1", "w2", "b2", "w3", "b3")] + [("scale", _f), ("n_pad", _i),
                                                                                                          ("h_pad", _i), ("reserved_", _i)]


class TrVitWeights(C.Structure):
    _fields_ = [(n, _vp) for n in ("patch_w", "patch_b", "cls_token", "pos_embed", "norm_g", "norm_b",
                                   "head_w", "head_b")] + [("blocks", TrBlockWeights * TR_MAX_DEPTH),
                                                           ("stage", TrStageWeights * TR_MAX_DEPTH)]


class TrVitConfig(C.Structure):
    _fields_ = [("family", _i), ("img_size", _i), ("patch", _i), ("in_chans", _i), ("embed_dim", _i),
                ("depth", _i), ("num_heads", _i), ("mlp_hidden", _i), ("num_classes", _i), ("ln_eps", _f),
                ("keep", _i * TR_MAX_DEPTH), ("precision", _i), ("knn_k", _i), ("cluster_iters", _i), ("sinkhorn_eps", _f),
                ("kmed_init", _i * TR_MAX_DEPTH), ("ats_dynamic", _i), ("concurrent", _i)]


class TrLinearGrad(C.Structure):     # tr_linear_grad: one layer of tr_linear_bwd_group
    _fields_ = [("dY", _vp), ("ldy", _l), ("X", _vp), ("ldx", _l), ("dW", _vp), ("db", _vp), ("M", _i), ("N", _i), ("K", _i)]


# every symbol include/tokenreduction_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "tr_version": (_i, []),
    "tr_last_error": (C.c_char_p, []),
    "tr_profile_begin": (_i, [_vp]),
    "tr_profile_end": (_i, [_i, _vp, _vp, _vp, _vp]),
    "tr_im2col_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_im2col_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_gemm_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_attention_bwd_long_workspace_floats": (_sz, [_i, _i, _i]),
    "tr_attention_bwd_long_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "tr_soft_dweights": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_soft_dsrc": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "tr_token_softmax_bwd_workspace_floats": (_sz, [_i, _i]),
    "tr_token_softmax_bwd": (_i, [_vp, _vp, _vp, _i, _f, _vp, _i, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "tr_sinkhorn_bwd": (_i, [_vp, _vp, _i, _f, _i, _vp, _i, _i, _i, _i, _vp]),
    "tr_rownorm_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tr_add_into_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "tr_gemm_gelu_keep_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_gemm_dgelu_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_gemm_split": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_layernorm_f32": (_i, [_vp, _l, _vp, _l, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "tr_attention_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_attention_split": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_attention_policy_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_attention_policy_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_tome_match": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_tome_merge_layernorm": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tr_gather_layernorm_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tr_cls_pos_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_patch_embed_supported": (_i, [_i, _i, _i, _i]),
    "tr_patch_embed_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_mlp_fused_supported": (_i, [_i, _i]),
    "tr_set_mlp_fused": (_i, [_i]),
    "tr_mlp_pack_bytes": (_sz, [_i, _i]),
    "tr_mlp_pack_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tr_mlp_fused_scratch_bytes": (_sz, [_i, _i]),
    "tr_mlp_fused_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "tr_set_mlp_poll_max": (_i, [_i]),
    "tr_mlp_fused_status": (_i, [_vp, _sz, _i, _i, _vp]),
    "tr_set_mlp_ln": (_i, [_i]),
    "tr_mlp_fused_ln_bf16": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "tr_gemm_clock_probe_read": (_i, [_vp]),
    "tr_mlp_clock_probe_read": (_i, [_vp]),
    "tr_lnlin_supported": (_i, [_i, _i]),
    "tr_lnlin_pack_bytes": (_sz, [_i, _i]),
    "tr_lnlin_scratch_bytes": (_sz, [_i, _i]),
    "tr_lnlin_pack_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "tr_lnlin_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "tr_set_mlp_resid_ln": (_i, [_i]),
    "tr_mlp_fused_resid_ln_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "tr_layernorm_bf16": (_i, [_vp, _l, _vp, _l, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "tr_attention_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_cls_topk": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_gather_layernorm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tr_vit_workspace_bytes": (_sz, [C.POINTER(TrVitConfig), _i]),
    "tr_vit_forward_status": (_i, [C.POINTER(TrVitConfig), _vp, _sz, _i, _vp]),
    "tr_pool_broadcast": (_i, [_vp, _i, _i, _i, _i, _f, _vp]),
    "tr_dyvit_score": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "tr_sit_merge": (_i, [_vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_softassign_merge": (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_softassign_merge_fast": (_i, [_vp, _i, _f, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_rownorm": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_sinkhorn": (_i, [_vp, _i, _f, _i, _vp, _vp, _i, _i, _i, _vp]),
    "tr_weighted_merge": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_ats_sample": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_ats_gather": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_ats_width": (_i, [_vp, _vp, _i, _i, _vp]),
    "tr_ats_narrow": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_dpcknn_workspace_floats": (_sz, [_i, _i]),
    "tr_kmedoids": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tr_kmedoids_equal": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tr_dpcknn_cluster": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tr_dpcknn_fused_supported": (_i, [_i, _i, _i]),
    "tr_dpcknn_cluster_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tr_cluster_merge_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "tr_broadcast_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "tr_residual_snapshot": (_i, [_vp, _vp, _i, _vp, _sz, _vp]),
    "tr_wgrad_workspace_floats": (_sz, [_i, _i, _i]),
    "tr_wgrad_bf16": (_i, [_vp, _l, _i, _vp, _l, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "tr_linear_bwd_params": (_i, [_vp, _l, _i, _vp, _l, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "tr_linear_bwd_group_workspace_floats": (_sz, [_vp, _i]),
    "tr_linear_bwd_group": (_i, [_vp, _i, _i, _vp, _sz, _vp]),
    "tr_linear_bwd_params2_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "tr_linear_bwd_params2": (_i, [_vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "tr_colsum_workspace_floats": (_sz, [_i, _i]),
    "tr_colsum_bf16": (_i, [_vp, _l, _i, _vp, _i, _vp, _sz, _i, _i, _vp]),
    "tr_gelu_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "tr_gelu_bwd_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "tr_layernorm_bwd_workspace_floats": (_sz, [_i, _i]),
    "tr_layernorm_bwd": (_i, [_vp, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _f, _vp]),
    "tr_layernorm_bwd_scatter_add": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _sz, _i, _i, _f, _vp]),
    "tr_attention_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_head_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "tr_embed_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_evit_fuse_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_tome_merge_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_f32_to_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "tr_cluster_merge_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _i, _vp]),
    "tr_ats_scatter": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tr_layernorm_bf16_to": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "tr_layernorm2_bf16": (_i, [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "tr_vit_tape_bytes": (_sz, [C.POINTER(TrVitConfig), _i]),
    "tr_vit_forward_train": (_i, [C.POINTER(TrVitConfig), C.POINTER(TrVitWeights), _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, C.POINTER(_i), _i, _vp,
                                  _vp, _f]),
    "tr_vit_dropout_mask_bytes": (_sz, [C.POINTER(TrVitConfig), _i]),
    "tr_cast_pack_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "tr_adamw_step": (_i, [_vp, _vp, _i, _i, C.c_double, C.c_double, C.c_double, C.c_float, C.c_float, _vp, _vp, _i, _vp]),
    "tr_dropout_bf16": (_i, [_vp, _vp, _vp, _f, _sz, _vp]),
    "tr_dropout_f32": (_i, [_vp, _vp, _vp, _f, _sz, _vp]),
    "tr_vit_tape_layout": (_i, [C.POINTER(TrVitConfig), _i, _i, C.POINTER(_sz)]),
    "tr_vit_backward_workspace_bytes": (_sz, [C.POINTER(TrVitConfig), _i]),
    "tr_vit_backward": (_i, [C.POINTER(TrVitConfig), C.POINTER(TrVitWeights), C.POINTER(TrVitWeights), C.POINTER(TrVitWeights), _vp, _vp, _vp,
                             _vp, _vp, _sz, _vp, _sz, _i, _i, _i, _i, _vp, _vp, _f]),
    "tr_rowscale_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_reduce_partials_f32": (_i, [_vp, _i, _sz, _vp, _i, _vp]),
    "tr_pool_policy": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "tr_pool_policy_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_dyvit_decide": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_dyvit_decide_bwd_workspace_floats": (_sz, [_i, _i, _i]),
    "tr_dyvit_decide_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i, _i, _i, _vp]),
    "tr_attention_policy_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tr_attention_policy_bwd_long_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "tr_head_sum": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tr_fill_f32": (_i, [_vp, _f, _sz, _vp]),
    "tr_add_patch_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "tr_vit_forward": (_i, [C.POINTER(TrVitConfig), C.POINTER(TrVitWeights), _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp,
                            C.POINTER(_i), _i, _vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and type the library.  Raises HipLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path of tokenreduction_amd "
            f"(no CPU fallback). Build it with `make -C {os.path.dirname(LIB_PATH)}` "
            f"or `python -c 'import __graft_entry__ as g; g.build()'`.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().tr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
