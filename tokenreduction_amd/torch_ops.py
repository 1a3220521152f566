"""`torch.library` registration of the C-ABI entry points ("exported entry points wrapped as torch.library ops").

    import tokenreduction_amd.torch_ops            # registers the namespace `torch.ops.tokenreduction_amd`
    y = torch.ops.tokenreduction_amd.linear(x_bf16, w_bf16, bias_f32, 0)       # differentiable: its backward is the HIP dgrad / wgrad
    out, cls = torch.ops.tokenreduction_amd.attention(qkv, B, N, H, True)

Each op has (a) a CUDA (= HIP on ROCm) implementation that calls the library through `ops.py` -- no CPU implementation is
registered, so a CPU tensor fails loudly with the dispatcher's "no kernel for backend CPU" error instead of falling back;
(b) a fake (meta) implementation giving output shapes / dtypes, so `torch.compile`, FakeTensorMode and `torch.export` trace
through the ops without running them; (c) for `linear`, `layernorm`, `attention` and `gelu` an autograd formula whose
backward is again a HIP entry point -- the hook-up point for a user who composes their own blocks from the ops instead of
using the whole-model executor (`models.py`, whose single autograd node lives in `training.py`).

PyTorch is plumbing here as everywhere else in the package: schemas, allocation and the dispatcher; all arithmetic is in csrc/.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops as _ops

NS = "tokenreduction_amd"
_lib_def = torch.library.Library(NS, "DEF")


def _define(schema: str, impl, fake):
    name = schema.split("(")[0]
    _lib_def.define(schema)
    torch.library.impl(f"{NS}::{name}", "CUDA")(impl)
    torch.library.register_fake(f"{NS}::{name}")(fake)


# ------------------------------------------------------------------------------------------------------------ forward ops
def _linear(a: Tensor, w: Tensor, bias: Tensor, epilogue: int) -> Tensor:
    return _ops.gemm(a, w, bias, epilogue)


def _linear_fake(a, w, bias, epilogue):
    dt = torch.bfloat16 if epilogue in (_ops.TR_EPI_BF16, _ops.TR_EPI_GELU_BF16) else torch.float32
    return a.new_empty((a.shape[0], w.shape[0]), dtype=dt)


_define("linear(Tensor a, Tensor w, Tensor bias, int epilogue) -> Tensor", _linear, _linear_fake)


def _layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    return _ops.layernorm(x.contiguous(), gamma, beta, eps)


_define("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps) -> Tensor", _layernorm,
        lambda x, gamma, beta, eps: x.new_empty((x.numel() // x.shape[-1], x.shape[-1]), dtype=torch.bfloat16))


def _gelu(pre: Tensor) -> Tensor:
    return _ops.gelu(pre)


_define("gelu(Tensor pre) -> Tensor", _gelu, lambda pre: torch.empty_like(pre))


def _attention(qkv: Tensor, B: int, N: int, H: int, want_cls: bool, size: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    out, cls = _ops.attention(qkv, B, N, H, want_cls=want_cls, size=size)
    return out, (cls if cls is not None else qkv.new_empty((0,), dtype=torch.float32))


def _attention_fake(qkv, B, N, H, want_cls, size):
    return (qkv.new_empty((B * N, H * 64), dtype=torch.bfloat16),
            qkv.new_empty((B, H, N) if want_cls else (0,), dtype=torch.float32))


_define("attention(Tensor qkv, int B, int N, int H, bool want_cls, Tensor? size) -> (Tensor, Tensor)", _attention, _attention_fake)


def _cls_topk(cls_rows: Tensor, K: int, want_compl: bool) -> Tuple[Tensor, Tensor, Tensor]:
    idx, compl, scores = _ops.cls_topk(cls_rows, K, want_compl)
    return idx, (compl if compl is not None else idx.new_empty((0,))), scores


def _cls_topk_fake(cls_rows, K, want_compl):
    B, H, N = cls_rows.shape
    return (cls_rows.new_empty((B, K), dtype=torch.int32), cls_rows.new_empty((B, N - 1 - K) if want_compl else (0,), dtype=torch.int32),
            cls_rows.new_empty((B, N - 1), dtype=torch.float32))


_define("cls_topk(Tensor cls_rows, int K, bool want_compl) -> (Tensor, Tensor, Tensor)", _cls_topk, _cls_topk_fake)


def _gather_layernorm(x: Tensor, idx: Tensor, compl: Optional[Tensor], scores: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float,
                      delta: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    return _ops.gather_layernorm(x, idx, compl, scores, gamma, beta, eps, delta=delta)


def _gather_layernorm_fake(x, idx, compl, scores, gamma, beta, eps, delta):
    B, _, D = x.shape
    n_out = idx.shape[1] + 1 + (1 if compl is not None else 0)
    return x.new_empty((B, n_out, D)), x.new_empty((B, n_out, D), dtype=torch.bfloat16)


_define("gather_layernorm(Tensor x, Tensor idx, Tensor? compl, Tensor? scores, Tensor gamma, Tensor beta, float eps, Tensor? delta) -> (Tensor, Tensor)",
        _gather_layernorm, _gather_layernorm_fake)


def _tome_match(qkv: Tensor, B: int, N: int, H: int, r: int) -> Tuple[Tensor, Tensor, Tensor]:
    return _ops.tome_match(qkv, B, N, H, r)


def _tome_match_fake(qkv, B, N, H, r):
    na = (N + 1) // 2
    return (qkv.new_empty((B, na - r), dtype=torch.int32), qkv.new_empty((B, r), dtype=torch.int32), qkv.new_empty((B, r), dtype=torch.int32))


_define("tome_match(Tensor qkv, int B, int N, int H, int r) -> (Tensor, Tensor, Tensor)", _tome_match, _tome_match_fake)


def _dpcknn_cluster(x: Tensor, K: int, noise: Optional[Tensor], k: int) -> Tuple[Tensor, Tensor, Tensor]:
    return _ops.dpcknn_cluster(x, K, noise=noise, k=k, fast_dist=True)


def _dpcknn_cluster_fake(x, K, noise, k):
    B, N, _ = x.shape
    return (x.new_empty((B, K), dtype=torch.int32), x.new_empty((B, N - 1), dtype=torch.int32), x.new_empty((B, N - 1), dtype=torch.float32))


_define("dpcknn_cluster(Tensor x, int K, Tensor? noise, int k) -> (Tensor, Tensor, Tensor)", _dpcknn_cluster, _dpcknn_cluster_fake)


def _softassign_merge(logits: Tensor, scale: float, x: Tensor, K: int, apply_softmax: bool, src: Optional[Tensor]) -> Tensor:
    return _ops.softassign_merge_fast(logits.clone(), scale, x, K, apply_softmax=apply_softmax, src=src)[0]


_define("softassign_merge(Tensor logits, float scale, Tensor x, int K, bool apply_softmax, Tensor? src) -> Tensor", _softassign_merge,
        lambda logits, scale, x, K, apply_softmax, src: x.new_empty((x.shape[0], K + 1, x.shape[2])))


# ------------------------------------------------------------------------------------------------------------ backward ops
def _linear_bwd(dy: Tensor, a: Tensor, w: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(d a bf16 [M,K], dW fp32 [N,K], d bias fp32 [N]) of y = a w^T + bias for dy bf16 [M,N]."""
    dw, db = _ops.linear_bwd_params(dy, a)
    zeros = torch.zeros(a.shape[1], dtype=torch.float32, device=a.device)
    da = _ops.gemm(dy, w.t().contiguous(), zeros, _ops.TR_EPI_BF16)
    return da, dw, db


_define("linear_bwd(Tensor dy, Tensor a, Tensor w) -> (Tensor, Tensor, Tensor)", _linear_bwd,
        lambda dy, a, w: (torch.empty_like(a), w.new_empty(w.shape, dtype=torch.float32), w.new_empty((w.shape[0],), dtype=torch.float32)))


def _layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    g, _, dg, db = _ops.layernorm_bwd(dy, x.reshape(-1, x.shape[-1]), gamma, eps)
    return g.reshape(x.shape), dg, db


_define("layernorm_bwd(Tensor dy, Tensor x, Tensor gamma, float eps) -> (Tensor, Tensor, Tensor)", _layernorm_bwd,
        lambda dy, x, gamma, eps: (torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)))


def _attention_bwd(qkv: Tensor, dout: Tensor, B: int, N: int, H: int, size: Optional[Tensor]) -> Tensor:
    return _ops.attention_bwd(qkv, dout, B, N, H, size=size)


_define("attention_bwd(Tensor qkv, Tensor dout, int B, int N, int H, Tensor? size) -> Tensor", _attention_bwd,
        lambda qkv, dout, B, N, H, size: torch.empty_like(qkv))


def _gelu_bwd(pre: Tensor, dh: Tensor) -> Tensor:
    return _ops.gelu_bwd(pre, dh.clone())


_define("gelu_bwd(Tensor pre, Tensor dh) -> Tensor", _gelu_bwd, lambda pre, dh: torch.empty_like(dh))


# ------------------------------------------------------------------------------------------------------------ autograd formulas
def _linear_setup(ctx, inputs, output):
    a, w, bias, epilogue = inputs
    ctx.save_for_backward(a, w)
    ctx.epilogue = epilogue


def _linear_backward(ctx, dy):
    if ctx.epilogue != _ops.TR_EPI_BF16:
        raise NotImplementedError("tokenreduction_amd::linear is differentiable with the plain bf16 epilogue (compose gelu as its own op)")
    a, w = ctx.saved_tensors
    da, dw, db = torch.ops.tokenreduction_amd.linear_bwd(dy.contiguous(), a, w)
    return da, dw.to(w.dtype), db, None


torch.library.register_autograd(f"{NS}::linear", _linear_backward, setup_context=_linear_setup)


def _layernorm_setup(ctx, inputs, output):
    x, gamma, beta, eps = inputs
    ctx.save_for_backward(x, gamma)
    ctx.eps = eps


def _layernorm_backward(ctx, dy):
    x, gamma = ctx.saved_tensors
    g, dg, db = torch.ops.tokenreduction_amd.layernorm_bwd(dy.contiguous(), x, gamma, ctx.eps)
    return g, dg, db, None


torch.library.register_autograd(f"{NS}::layernorm", _layernorm_backward, setup_context=_layernorm_setup)


def _attention_setup(ctx, inputs, output):
    qkv, B, N, H, want_cls, size = inputs
    ctx.save_for_backward(qkv, size)
    ctx.dims = (B, N, H)


def _attention_backward(ctx, dout, dcls):
    qkv, size = ctx.saved_tensors
    B, N, H = ctx.dims
    if dcls is not None and bool(dcls.ne(0).any()):
        # the second output (the CLS query's softmax rows, per head) is a side output for selection scores.  The native backward takes the
        # gradient of their head MEAN only (tr_attention_bwd_bf16's `dcls`, what EViT's fused token needs: the model executor wires it);
        # a per-head gradient has no kernel -- refuse instead of returning a gradient that silently lacks this path
        raise NotImplementedError("tokenreduction_amd::attention: a gradient reached the `cls` output; only `out` is differentiable "
                                  "through this op (EViT's cls_attn gradient is handled inside the model's training executor)")
    return torch.ops.tokenreduction_amd.attention_bwd(qkv, dout.contiguous(), B, N, H, size), None, None, None, None, None


torch.library.register_autograd(f"{NS}::attention", _attention_backward, setup_context=_attention_setup)


def _gelu_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _gelu_backward(ctx, dh):
    return torch.ops.tokenreduction_amd.gelu_bwd(ctx.saved_tensors[0], dh.contiguous())


torch.library.register_autograd(f"{NS}::gelu", _gelu_backward, setup_context=_gelu_setup)

OPS = ("linear", "layernorm", "gelu", "attention", "cls_topk", "gather_layernorm", "tome_match", "dpcknn_cluster", "softassign_merge",
       "linear_bwd", "layernorm_bwd", "attention_bwd", "gelu_bwd")
