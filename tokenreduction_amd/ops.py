"""Thin tensor-level wrappers over the C ABI (one per SURVEY.md section 8a row group).

PyTorch is plumbing here: it owns device memory and the stream; every byte of arithmetic happens
in the hand-written gfx950 kernels behind include/tokenreduction_hip.h.  All tensors must live
on a ROCm device -- there is no CPU path.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import (TR_EPI_BF16, TR_EPI_F32, TR_EPI_GELU_BF16, TR_EPI_PATCH_F32,  # noqa: F401
                   TR_EPI_RESID_F32)


def _stream(t: torch.Tensor = None) -> int:
    """The launch stream: the current stream of the tensor's device (not of whatever device happens to be current)."""
    return torch.cuda.current_stream(None if t is None else t.device).cuda_stream


def _same_device(*tensors):
    devs = {t.device for t in tensors if t is not None}
    if len(devs) > 1:
        raise ValueError(f"operands live on different devices: {sorted(str(d) for d in devs)}")


def _dev(t: torch.Tensor, dtype, name: str) -> int:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; tokenreduction_amd has no CPU path (HIP kernels only)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _opt(t, dtype, name):
    return None if t is None else _dev(t, dtype, name)


def im2col(img: torch.Tensor, patch: int) -> torch.Tensor:
    """PatchEmbed unfold (topk.py:181): fp32 [B,C,H,W] -> bf16 [B*P, C*p*p]."""
    B, Cc, H, W = img.shape
    cols = torch.empty(B * (H // patch) * (W // patch), Cc * patch * patch, dtype=torch.bfloat16, device=img.device)
    _lib.check(_lib.load().tr_im2col_bf16(_dev(img, torch.float32, "img"), cols.data_ptr(), B, Cc, H, W, patch, _stream()),
               "tr_im2col_bf16")
    return cols


def cls_pos_rows(cls_token, pos_embed, x, B, N, D):
    _lib.check(_lib.load().tr_cls_pos_rows(_dev(cls_token, torch.float32, "cls_token"), _dev(pos_embed, torch.float32, "pos_embed"),
                                           _dev(x, torch.float32, "x"), B, N, D, _stream()), "tr_cls_pos_rows")


def patch_embed(img: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, cls_token: torch.Tensor, pos_embed: torch.Tensor, patch: int = 16) -> torch.Tensor:
    """PatchEmbed + cls_token + pos_embed in one launch (topk.py:181-186): img fp32 [B,C,H,H], w bf16 [D, C*p*p] -> x fp32 [B, P+1, D]."""
    B, Cc, H, _ = img.shape
    D = w.shape[0]
    x = torch.empty(B, (H // patch) ** 2 + 1, D, dtype=torch.float32, device=img.device)
    _lib.check(_lib.load().tr_patch_embed_bf16(_dev(img, torch.float32, "img"), _dev(w, torch.bfloat16, "w"), _dev(bias, torch.float32, "bias"),
                                               _dev(cls_token, torch.float32, "cls_token"), _dev(pos_embed, torch.float32, "pos_embed"), x.data_ptr(),
                                               B, Cc, H, patch, D, _stream()), "tr_patch_embed_bf16")
    return x


def gemm(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, epilogue: int, out: torch.Tensor = None,
         aux: torch.Tensor = None, aux_i: int = 0) -> torch.Tensor:
    """nn.Linear on MFMA: epilogue(a[M,K] @ w[N,K]^T + bias).  For RESID/PATCH `out` is required (updated in place)."""
    M, K = a.shape
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K:
        raise ValueError(f"gemm: weight is {tuple(w.shape)}, expected [N, {K}] for an activation of {tuple(a.shape)}")
    if bias.numel() != N:
        raise ValueError(f"gemm: bias has {bias.numel()} entries for {N} output columns")
    _same_device(a, w, bias, out, aux)
    if out is not None and epilogue not in (TR_EPI_PATCH_F32,) and tuple(out.shape) != (M, N):
        raise ValueError(f"gemm: out is {tuple(out.shape)}, expected {(M, N)}")
    if out is None:
        if epilogue in (TR_EPI_BF16, TR_EPI_GELU_BF16):
            out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
        elif epilogue == TR_EPI_F32:
            out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        else:
            raise ValueError("this epilogue needs an explicit `out`")
    odt = torch.bfloat16 if epilogue in (TR_EPI_BF16, TR_EPI_GELU_BF16) else torch.float32
    _lib.check(_lib.load().tr_gemm_bf16(_dev(a, torch.bfloat16, "a"), _dev(w, torch.bfloat16, "w"), _dev(bias, torch.float32, "bias"),
                                        _dev(out, odt, "out"), _opt(aux, torch.float32, "aux"), aux_i, M, N, K, epilogue, _stream(a)),
               "tr_gemm_bf16")
    return out


def set_mlp_fused(mode: int) -> int:
    """Which Mlp the eval executor runs: 1 the fused launch wherever supported, 0 never, -1 (default) where its block schedule fills the chip
    (tr_set_mlp_fused; the two are bit-identical).  Process-wide, read when the launches are enqueued: a captured hipGraph keeps the mode it
    was captured with (drop `model._ws` to re-capture).  Returns the previous mode."""
    return int(_lib.load().tr_set_mlp_fused(int(mode)))


def mlp_pack(fc1_w: torch.Tensor, fc2_w: torch.Tensor, fc2_b: torch.Tensor) -> torch.Tensor:
    """Fragment-major copy of a block's Mlp weights (bf16 [Hd,D], [D,Hd]) and fc2's bias (fp32 [D]: the accumulators' start image) for
    mlp_fused; repack whenever they change."""
    Hd, D = fc1_w.shape
    if tuple(fc2_w.shape) != (D, Hd):
        raise ValueError(f"mlp_pack: fc2 weight is {tuple(fc2_w.shape)}, expected {(D, Hd)}")
    _same_device(fc1_w, fc2_w, fc2_b)
    if fc2_b.numel() != D:
        raise ValueError(f"mlp_pack: fc2 bias has {fc2_b.numel()} entries for {D} output columns")
    lib = _lib.load()
    n = lib.tr_mlp_pack_bytes(D, Hd)
    if n == 0 or not lib.tr_mlp_fused_supported(D, Hd):
        raise ValueError(f"mlp_pack: the fused Mlp kernel does not serve D={D}, Hd={Hd}")
    pk = torch.empty(n, dtype=torch.uint8, device=fc1_w.device)
    _lib.check(lib.tr_mlp_pack_bf16(_dev(fc1_w, torch.bfloat16, "fc1_w"), _dev(fc2_w, torch.bfloat16, "fc2_w"), _dev(fc2_b, torch.float32, "fc2_b"),
                                    pk.data_ptr(), D, Hd, _stream(fc1_w)), "tr_mlp_pack_bf16")
    return pk


_MLP_SCRATCH = {}


def _mlp_scratch(dev, D, Hd, streamk=True):
    """The stream-K hand-over scratch of a fused-Mlp launch on the CURRENT stream of `dev`: a scratch belongs to one launch at a time (its
    counters are zeroed in front of every launch), so launches that may overlap -- other streams -- must not share one: cached per
    (device, stream).  Returns (tensor | None, nbytes)."""
    if not streamk:
        return None, 0
    nbytes = int(_lib.load().tr_mlp_fused_scratch_bytes(D, Hd))
    if not nbytes:
        return None, 0
    with torch.cuda.device(dev):
        key = (dev, torch.cuda.current_stream().cuda_stream, nbytes)
    if key not in _MLP_SCRATCH:
        _MLP_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return _MLP_SCRATCH[key], nbytes


def mlp_fused(xn: torch.Tensor, packed: torch.Tensor, fc1_b: torch.Tensor, out: torch.Tensor = None, streamk: bool = True) -> torch.Tensor:
    """timm Mlp (fc1 -> GELU -> fc2, topk.py:95) of the eval forward in one launch: xn bf16 [M,D] -> bf16 [M,D]; bit-identical to
    gemm(GELU_BF16) followed by gemm(BF16).  streamk: give the launch its hand-over scratch (with more blocks of 128 rows than the device
    has compute units the steps are then dealt evenly over the workgroups); False: whole blocks round-robin.  Same bits."""
    M, D = xn.shape
    Hd = fc1_b.numel()
    _same_device(xn, packed, fc1_b, out)
    if out is None:
        out = torch.empty(M, D, dtype=torch.bfloat16, device=xn.device)
    scratch, nbytes = _mlp_scratch(xn.device, D, Hd, streamk)
    _lib.check(_lib.load().tr_mlp_fused_bf16(_dev(xn, torch.bfloat16, "xn"), _dev(packed, torch.uint8, "packed"), _dev(fc1_b, torch.float32, "fc1_b"),
                                             _dev(out, torch.bfloat16, "out"),
                                             None if scratch is None else scratch.data_ptr(), nbytes, M, D, Hd, _stream(xn)),
               "tr_mlp_fused_bf16")
    return out


def set_mlp_ln(mode: int) -> int:
    """Where the eval executor runs a lazy norm2 INSIDE the fused Mlp launch that follows it (mlp_fused_ln) instead of as its own LayerNorm
    launch -- the two are bit-identical: 1 (default) where that launch is one round of whole blocks and in every launch of a forward that
    runs beside others (model.forward_async: whole-block launches), 2 wherever the fused Mlp runs, 0 never.
    Process-wide, read when the launches are enqueued: a captured hipGraph keeps the form it was captured with (drop `model._ws` to
    re-capture).  Returns the previous setting."""
    return int(_lib.load().tr_set_mlp_ln(int(mode)))


def mlp_fused_ln(x: torch.Tensor, delta: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, packed: torch.Tensor,
                 fc1_b: torch.Tensor, out: torch.Tensor = None, streamk: bool = True) -> torch.Tensor:
    """topk.py:95's `self.mlp(self.norm2(x))` in one launch: Mlp(LayerNorm(x + delta)) with x the fp32 stream [M,D] and delta the pending
    bf16 attention residual [M,D] (neither is written) -> bf16 [M,D].  Bit-identical to layernorm2 (no write-back) followed by mlp_fused."""
    M, D = x.shape
    Hd = fc1_b.numel()
    _same_device(x, delta, gamma, beta, packed, fc1_b, out)
    if tuple(delta.shape) != (M, D) or not x.is_contiguous() or not delta.is_contiguous():
        raise ValueError(f"mlp_fused_ln: x {tuple(x.shape)} and delta {tuple(delta.shape)} must be contiguous [M, D]")
    if out is None:
        out = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    scratch, nbytes = _mlp_scratch(x.device, D, Hd, streamk)
    _lib.check(_lib.load().tr_mlp_fused_ln_bf16(_dev(x, torch.float32, "x"), _dev(delta, torch.bfloat16, "delta"), _dev(gamma, torch.float32, "gamma"),
                                                _dev(beta, torch.float32, "beta"), float(eps), _dev(packed, torch.uint8, "packed"),
                                                _dev(fc1_b, torch.float32, "fc1_b"), _dev(out, torch.bfloat16, "out"),
                                                None if scratch is None else scratch.data_ptr(), nbytes, M, D, Hd, _stream(x)),
               "tr_mlp_fused_ln_bf16")
    return out


def mlp_fused_status(dev=None, D: int = 384, Hd: int = 1536) -> None:
    """Status check of the fused-Mlp launches of the current stream of `dev` that went through this module's scratch: waits for the stream and
    raises if a stream-K hand-over poll ran out in one of them (tr_mlp_fused_status; the record is cleared)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    scratch, nbytes = _mlp_scratch(dev, D, Hd, True)
    if scratch is None:
        return
    with torch.cuda.device(dev):
        _lib.check(_lib.load().tr_mlp_fused_status(scratch.data_ptr(), nbytes, D, Hd, torch.cuda.current_stream().cuda_stream),
                   "tr_mlp_fused_status")


def set_mlp_poll_max(iterations: int) -> int:
    """Bound of the stream-K hand-over poll (iterations of an 8-tick sleep; default 2^24: seconds).  Tests shorten it.  Returns the previous bound."""
    return int(_lib.load().tr_set_mlp_poll_max(int(iterations)))


_LNLIN_SCRATCH = {}


def lnlin_pack(w: torch.Tensor) -> torch.Tensor:
    """Fragment-major copy of an nn.Linear weight [N, 384] (bf16) for lnlin; repack whenever it changes."""
    N, D = w.shape
    lib = _lib.load()
    n = int(lib.tr_lnlin_pack_bytes(D, N))
    if n == 0:
        raise ValueError(f"lnlin_pack: the LayerNorm + Linear kernel does not serve D={D}, N={N}")
    pk = torch.empty(n, dtype=torch.uint8, device=w.device)
    _lib.check(lib.tr_lnlin_pack_bf16(_dev(w, torch.bfloat16, "w"), pk.data_ptr(), D, N, _stream(w)), "tr_lnlin_pack_bf16")
    return pk


def lnlin(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, packed: torch.Tensor, bias: torch.Tensor,
          d1: torch.Tensor = None, d2: torch.Tensor = None, out: torch.Tensor = None):
    """The start of a block in one launch (topk.py:86-87 norm1 + :44 qkv): v = (x [+ d1]) [+ d2]; returns (out bf16 [M,N] =
    LayerNorm(v) @ W^T + bias, x_out = v as a NEW fp32 tensor, or None when nothing was pending).  Bit-identical to layernorm / layernorm2
    followed by gemm(TR_EPI_BF16)."""
    M, D = x.shape
    N = bias.numel()
    _same_device(x, gamma, beta, packed, bias, d1, d2, out)
    if not x.is_contiguous() or (d1 is not None and not d1.is_contiguous()) or (d2 is not None and not d2.is_contiguous()):
        raise ValueError("lnlin: x, d1, d2 must be contiguous [M, D]")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    x_out = torch.empty_like(x) if d1 is not None else None
    lib = _lib.load()
    nbytes = int(lib.tr_lnlin_scratch_bytes(D, N))
    with torch.cuda.device(x.device):
        key = (x.device, torch.cuda.current_stream().cuda_stream, nbytes)
    if key not in _LNLIN_SCRATCH:
        _LNLIN_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    scratch = _LNLIN_SCRATCH[key]
    _lib.check(lib.tr_lnlin_bf16(_dev(x, torch.float32, "x"), _opt(d1, torch.bfloat16, "d1"), _opt(d2, torch.bfloat16, "d2"),
                                 None if x_out is None else x_out.data_ptr(), _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta"),
                                 float(eps), _dev(packed, torch.uint8, "packed"), _dev(bias, torch.float32, "bias"), _dev(out, torch.bfloat16, "out"),
                                 scratch.data_ptr(), nbytes, M, D, N, _stream(x)), "tr_lnlin_bf16")
    return out, x_out


def set_mlp_resid_ln(on: bool) -> bool:
    """Whether the eval executor fuses the block tail (Mlp + residual add + the next block's norm1) into one launch where it can, or keeps
    the fused Mlp and the LayerNorm launch apart (default: the one-launch form measured 4 % slower in the model).  Process-wide, read when
    the launches are enqueued: a captured hipGraph keeps the form it was captured with (drop `model._ws` to re-capture).  Returns the
    previous setting."""
    return bool(_lib.load().tr_set_mlp_resid_ln(1 if on else 0))


def mlp_fused_resid_ln(xn: torch.Tensor, packed: torch.Tensor, fc1_b: torch.Tensor, fc2_b: torch.Tensor, x: torch.Tensor, next_g: torch.Tensor,
                       next_b: torch.Tensor, eps: float, xn_next: torch.Tensor = None, streamk: bool = True) -> torch.Tensor:
    """Block tail + next block's norm1 in one launch (topk.py:95, then :87): x (fp32 [M,D], updated IN PLACE) += Mlp(xn) and
    returns LayerNorm(x; next_g, next_b, eps) as bf16 [M,D]."""
    M, D = xn.shape
    Hd = fc1_b.numel()
    _same_device(xn, packed, fc1_b, fc2_b, x, next_g, next_b, xn_next)
    if tuple(x.shape) != (M, D) or not x.is_contiguous():
        raise ValueError(f"mlp_fused_resid_ln: x is {tuple(x.shape)}, expected contiguous {(M, D)}")
    if xn_next is None:
        xn_next = torch.empty(M, D, dtype=torch.bfloat16, device=xn.device)
    lib = _lib.load()
    scratch, nbytes = _mlp_scratch(xn.device, D, Hd, streamk)
    _lib.check(lib.tr_mlp_fused_resid_ln_bf16(_dev(xn, torch.bfloat16, "xn"), _dev(packed, torch.uint8, "packed"), _dev(fc1_b, torch.float32, "fc1_b"),
                                              _dev(fc2_b, torch.float32, "fc2_b"), _dev(x, torch.float32, "x"), _dev(next_g, torch.float32, "next_g"),
                                              _dev(next_b, torch.float32, "next_b"), float(eps), _dev(xn_next, torch.bfloat16, "xn_next"),
                                              None if scratch is None else scratch.data_ptr(), nbytes, M, D, Hd,
                                              _stream(xn)), "tr_mlp_fused_resid_ln_bf16")
    return xn_next


def layernorm(x: torch.Tensor, gamma, beta, eps: float, rows: int = None, ldx: int = None, delta: torch.Tensor = None,
              ldd: int = None) -> torch.Tensor:
    """[x += delta (bf16, written back);] nn.LayerNorm over the last dim of fp32 x -> bf16 [M,D]
    (optionally only `rows` rows at stride ldx / ldd)."""
    D = x.shape[-1]
    M = rows if rows is not None else x.numel() // D
    ld = ldx if ldx is not None else D
    y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().tr_layernorm_bf16(_dev(x, torch.float32, "x"), ld, _opt(delta, torch.bfloat16, "delta"),
                                             ldd if ldd is not None else D, _dev(gamma, torch.float32, "gamma"),
                                             _dev(beta, torch.float32, "beta"), y.data_ptr(), M, D, eps, _stream()),
               "tr_layernorm_bf16")
    return y


def attention(qkv: torch.Tensor, B: int, N: int, H: int, want_cls: bool = False, size: torch.Tensor = None,
              colsum_part: torch.Tensor = None):
    """softmax(q k^T / 8) v per (image, head) (topk.py:44-51).  qkv bf16 [B*N, 3*H*64] -> (out bf16 [B*N, H*64], cls_rows|None)."""
    out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=qkv.device)
    cls_rows = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device) if want_cls else None
    _lib.check(_lib.load().tr_attention_bf16(_dev(qkv, torch.bfloat16, "qkv"), out.data_ptr(),
                                             None if cls_rows is None else cls_rows.data_ptr(), _opt(size, torch.float32, "size"),
                                             _opt(colsum_part, torch.float32, "colsum_part"), B, N, H, _stream()),
               "tr_attention_bf16")
    return out, cls_rows


def cls_topk(cls_rows: torch.Tensor, K: int, want_compl: bool = False):
    """Top-K over head-mean CLS attention (topk.py:59-61) -> (idx int32 [B,K], compl int32 [B,P-K]|None, scores fp32 [B,P])."""
    B, H, N = cls_rows.shape
    P = N - 1
    idx = torch.empty(B, K, dtype=torch.int32, device=cls_rows.device)
    compl = torch.empty(B, P - K, dtype=torch.int32, device=cls_rows.device) if want_compl else None
    scores = torch.empty(B, P, dtype=torch.float32, device=cls_rows.device)
    _lib.check(_lib.load().tr_cls_topk(_dev(cls_rows, torch.float32, "cls_rows"), idx.data_ptr(),
                                       None if compl is None else compl.data_ptr(), scores.data_ptr(), B, H, N, K, _stream()),
               "tr_cls_topk")
    return idx, compl, scores


def gather_layernorm(x: torch.Tensor, idx, compl, scores, gamma, beta, eps: float, delta: torch.Tensor = None):
    """Top-K gather/compact (topk.py:89-93) [+ EViT fused token evit.py:111-123] fused with norm2.
    With delta (bf16 [B,N,D], pending attn.proj output) the gather reads x + delta.
    x fp32 [B,N,D] -> (x_out fp32 [B,N_out,D], y bf16 [B,N_out,D])."""
    B, N, D = x.shape
    K = idx.shape[1]
    N_out = K + 1 + (1 if compl is not None else 0)
    x_out = torch.empty(B, N_out, D, dtype=torch.float32, device=x.device)
    y = torch.empty(B, N_out, D, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().tr_gather_layernorm_bf16(
        _dev(x, torch.float32, "x"), _opt(delta, torch.bfloat16, "delta"), _dev(idx, torch.int32, "idx"), _opt(compl, torch.int32, "compl"),
        _opt(scores, torch.float32, "scores"), _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta"),
        x_out.data_ptr(), y.data_ptr(), B, N, K, D, eps, _stream()), "tr_gather_layernorm_bf16")
    return x_out, y


# ---------------------------------------------------------------------------------------- fp32 validation path
def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, epilogue: int = TR_EPI_F32, out: torch.Tensor = None,
             aux: torch.Tensor = None, aux_i: int = 0, split: bool = False) -> torch.Tensor:
    """nn.Linear in the reference's arithmetic (fp32 operands): epilogue TR_EPI_F32 | TR_EPI_GELU_BF16 (GELU, fp32 out) | PATCH.
    split=True: the same Linear as split-bf16 (hi/lo) products on the matrix cores (TR_PREC_BF16X3)."""
    M, K = a.shape
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K:
        raise ValueError(f"gemm: weight is {tuple(w.shape)}, expected [N, {K}] for an activation of {tuple(a.shape)}")
    if bias.numel() != N:
        raise ValueError(f"gemm: bias has {bias.numel()} entries for {N} output columns")
    _same_device(a, w, bias, out, aux)
    if out is not None and epilogue not in (TR_EPI_PATCH_F32,) and tuple(out.shape) != (M, N):
        raise ValueError(f"gemm: out is {tuple(out.shape)}, expected {(M, N)}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    fn = _lib.load().tr_gemm_split if split else _lib.load().tr_gemm_f32
    _lib.check(fn(_dev(a, torch.float32, "a"), _dev(w, torch.float32, "w"), _dev(bias, torch.float32, "bias"),
                  _dev(out, torch.float32, "out"), _opt(aux, torch.float32, "aux"), aux_i, M, N, K, epilogue, _stream()),
               "tr_gemm_split" if split else "tr_gemm_f32")
    return out


def layernorm2(x: torch.Tensor, gamma, beta, eps: float, delta: torch.Tensor, delta2: torch.Tensor = None, write_x: bool = True) -> torch.Tensor:
    """y = LayerNorm((x + delta) + delta2) -> bf16 [M,D]; the sum is written back to x unless write_x is False (tr_layernorm2_bf16:
    the eval executor's norm2 without a stream write, and the norm1 that absorbs both pending residuals after it)."""
    D = x.shape[-1]
    M = x.numel() // D
    y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    xp = _dev(x, torch.float32, "x")
    _lib.check(_lib.load().tr_layernorm2_bf16(xp, D, xp if write_x else None, D, _dev(delta, torch.bfloat16, "delta"), D,
                                              _opt(delta2, torch.bfloat16, "delta2"), D, _dev(gamma, torch.float32, "gamma"),
                                              _dev(beta, torch.float32, "beta"), y.data_ptr(), M, D, eps, _stream()), "tr_layernorm2_bf16")
    return y


def layernorm_f32(x: torch.Tensor, gamma, beta, eps: float, delta: torch.Tensor = None) -> torch.Tensor:
    D = x.shape[-1]
    M = x.numel() // D
    y = torch.empty(M, D, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tr_layernorm_f32(_dev(x, torch.float32, "x"), D, _opt(delta, torch.float32, "delta"), D,
                                            _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta"), y.data_ptr(), M, D,
                                            eps, _stream()), "tr_layernorm_f32")
    return y


def attention_f32(qkv: torch.Tensor, B: int, N: int, H: int, want_cls: bool = False, size: torch.Tensor = None,
                  colsum_part: torch.Tensor = None, split: bool = False):
    out = torch.empty(B * N, H * 64, dtype=torch.float32, device=qkv.device)
    cls_rows = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device) if want_cls else None
    fn = _lib.load().tr_attention_split if split else _lib.load().tr_attention_f32
    _lib.check(fn(_dev(qkv, torch.float32, "qkv"), out.data_ptr(), None if cls_rows is None else cls_rows.data_ptr(),
                  _opt(size, torch.float32, "size"), _opt(colsum_part, torch.float32, "colsum_part"), B, N, H, _stream()),
               "tr_attention_split" if split else "tr_attention_f32")
    return out, cls_rows


# ---------------------------------------------------------------------------------------- ToMe (models/tome.py)
def ats_width(new_mask: torch.Tensor) -> int:
    """ats.py:77-78: the token count the reference keeps after a sampling block = the batch maximum of valid ids (row sums of new_mask)."""
    B, K = new_mask.shape
    w = torch.zeros(1, dtype=torch.int32, device=new_mask.device)
    _lib.check(_lib.load().tr_ats_width(_dev(new_mask, torch.float32, "new_mask"), w.data_ptr(), B, K, _stream(new_mask)), "tr_ats_width")
    return int(w.item())


def ats_narrow(ids: torch.Tensor, new_mask: torch.Tensor, Kw: int):
    """The first Kw columns of ids / new_mask [B,K] as contiguous [B,Kw] tensors (what ats_gather and the next block's key mask take)."""
    B, K = ids.shape
    ids_out = torch.empty(B, Kw, dtype=torch.int32, device=ids.device)
    mask_out = torch.empty(B, Kw, dtype=torch.float32, device=ids.device)
    _lib.check(_lib.load().tr_ats_narrow(_dev(ids, torch.int32, "ids"), _dev(new_mask, torch.float32, "new_mask"), ids_out.data_ptr(),
                                         mask_out.data_ptr(), B, K, int(Kw), _stream(ids)), "tr_ats_narrow")
    return ids_out, mask_out


def tome_match(qkv: torch.Tensor, B: int, N: int, H: int, r: int):
    """bipartite_soft_matching (tome.py:230-277) on metric = k.mean(1): qkv bf16|fp32 [B*N, 3*H*64] ->
    (unm_idx [B, ceil(N/2)-r], src_idx [B,r], dst_idx [B,r]) int32."""
    if qkv.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("qkv must be bf16 or fp32")
    na = (N + 1) // 2
    unm = torch.empty(B, na - r, dtype=torch.int32, device=qkv.device)
    src = torch.empty(B, r, dtype=torch.int32, device=qkv.device)
    dst = torch.empty(B, r, dtype=torch.int32, device=qkv.device)
    _lib.check(_lib.load().tr_tome_match(_dev(qkv, qkv.dtype, "qkv"), int(qkv.dtype == torch.float32), unm.data_ptr(), src.data_ptr(),
                                         dst.data_ptr(), B, N, H, r, _stream()), "tr_tome_match")
    return unm, src, dst


def tome_merge_layernorm(x: torch.Tensor, delta, size, unm, src, dst, gamma, beta, eps: float, f32: bool = False):
    """merge_wavg (tome.py:309-323) of x (+ delta) with norm2 fused: -> (x_out fp32 [B,N-r,D], size_out fp32 [B,N-r], y)."""
    B, N, D = x.shape
    r = src.shape[1]
    x_out = torch.empty(B, N - r, D, dtype=torch.float32, device=x.device)
    size_out = torch.empty(B, N - r, dtype=torch.float32, device=x.device)
    y = torch.empty(B, N - r, D, dtype=torch.float32 if f32 else torch.bfloat16, device=x.device)
    ddt = torch.float32 if f32 else torch.bfloat16
    _lib.check(_lib.load().tr_tome_merge_layernorm(
        _dev(x, torch.float32, "x"), _opt(delta, ddt, "delta"), int(f32), _opt(size, torch.float32, "size"), _dev(unm, torch.int32, "unm"),
        _dev(src, torch.int32, "src"), _dev(dst, torch.int32, "dst"), _dev(gamma, torch.float32, "gamma"),
        _dev(beta, torch.float32, "beta"), x_out.data_ptr(), size_out.data_ptr(), y.data_ptr(), B, N, r, D, eps, _stream()),
        "tr_tome_merge_layernorm")
    return x_out, size_out, y


# ---------------------------------------------------------------------------------------- DyViT / SiT (reduce before the block)
def pool_broadcast(h: torch.Tensor, B: int, N: int, eps: float = 1e-6):
    """PredictorLG global feature (dyvit.py:115-118, policy == 1): in place on h bf16|fp32 [B*N, C]."""
    if h.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("h must be bf16 or fp32")
    C_ = h.shape[-1]
    _lib.check(_lib.load().tr_pool_broadcast(_dev(h, h.dtype, "h"), int(h.dtype == torch.float32), B, N, C_, eps, _stream()),
               "tr_pool_broadcast")
    return h


def dyvit_score(h: torch.Tensor, w: torch.Tensor, b: torch.Tensor):
    """out_conv.4 + LogSoftmax + [..., 0] (dyvit.py:108-109, 231): h bf16|fp32 [M, C], w fp32 [2, C] -> scores fp32 [M]."""
    M, C_ = h.shape
    scores = torch.empty(M, dtype=torch.float32, device=h.device)
    _lib.check(_lib.load().tr_dyvit_score(_dev(h, h.dtype, "h"), int(h.dtype == torch.float32), _dev(w, torch.float32, "w"),
                                          _dev(b, torch.float32, "b"), scores.data_ptr(), M, C_, _stream()), "tr_dyvit_score")
    return scores


def softassign_merge_fast(logits: torch.Tensor, scale: float, x: torch.Tensor, K: int, apply_softmax: bool = True,
                          want_soft: bool = False, src: torch.Tensor = None):
    """MFMA version of sit_merge / weighted_merge (bf16 executor; `logits` is overwritten by the weights when apply_softmax):
    -> (x_out fp32 [B,K+1,D], soft fp32 [B,K,N-1] | None)."""
    B, N, D = x.shape
    ldl = logits.shape[-1]
    x_out = torch.empty(B, K + 1, D, dtype=torch.float32, device=x.device)
    soft = torch.empty(B, K, N - 1, dtype=torch.float32, device=x.device) if (want_soft and apply_softmax) else None
    _lib.check(_lib.load().tr_softassign_merge_fast(_dev(logits, torch.float32, "logits"), ldl, float(scale), int(apply_softmax),
                                                    _dev(x, torch.float32, "x"), _dev(x if src is None else src, torch.float32, "src"),
                                                    x_out.data_ptr(), None if soft is None else soft.data_ptr(), B, N, K, D, _stream()),
               "tr_softassign_merge_fast")
    return x_out, soft


def sit_merge(logits: torch.Tensor, scale: float, x: torch.Tensor, K: int, want_soft: bool = False, src: torch.Tensor = None):
    """TokenSlimmingModule tail (sit.py:38-39): logits fp32 [B,N,ldl] (first K columns), x fp32 [B,N,D] ->
    (x_out fp32 [B,K+1,D], soft fp32 [B,K,N-1] | None)."""
    B, N, D = x.shape
    ldl = logits.shape[-1]
    x_out = torch.empty(B, K + 1, D, dtype=torch.float32, device=x.device)
    soft = torch.empty(B, K, N - 1, dtype=torch.float32, device=x.device) if want_soft else None
    _lib.check(_lib.load().tr_softassign_merge(_dev(logits, torch.float32, "logits"), ldl, float(scale), _dev(x, torch.float32, "x"),
                                               _dev(x if src is None else src, torch.float32, "src"), x_out.data_ptr(),
                                               None if soft is None else soft.data_ptr(), B, N, K, D, _stream()),
               "tr_softassign_merge")
    return x_out, soft


# ---------------------------------------------------------------------------------------- DPC-KNN (models/dpcknn.py)
def dpcknn_cluster(x: torch.Tensor, K: int, noise: torch.Tensor = None, k: int = 5, fast_dist=False):
    """cluster_dpc_knn (dpcknn.py:44-100) on the patch rows of x fp32 [B,N,D] ->
    (centers int32 [B,K] in descending-score order, idx_cluster int32 [B,N-1], scores fp32 [B,N-1]).
    fast_dist: False / 0 = fp32 distances; True / 1 = split-bf16 MFMA distances, in ONE launch where the kernel applies; 2 = the same
    distances through the staged launches."""
    B, N, D = x.shape
    lib = _lib.load()
    ws = torch.empty(lib.tr_dpcknn_workspace_floats(B, N), dtype=torch.float32, device=x.device)
    centers = torch.empty(B, K, dtype=torch.int32, device=x.device)
    idx_cluster = torch.empty(B, N - 1, dtype=torch.int32, device=x.device)
    scores = torch.empty(B, N - 1, dtype=torch.float32, device=x.device)
    _lib.check(lib.tr_dpcknn_cluster(_dev(x, torch.float32, "x"), _opt(noise, torch.float32, "noise"), ws.data_ptr(),
                                     centers.data_ptr(), idx_cluster.data_ptr(), scores.data_ptr(), B, N, D, K, k, int(fast_dist),
                                     _stream()),
               "tr_dpcknn_cluster")
    return centers, idx_cluster, scores


def cluster_merge_layernorm(x: torch.Tensor, idx_cluster: torch.Tensor, K: int, gamma, beta, eps: float, score_w=None, score_b=None,
                            f32: bool = False):
    """merge_tokens (dpcknn.py:103-132) with token_weight = exp(Linear(D,1)(x)) (None = equal weights) + the next LayerNorm:
    x fp32 [B,N,D] -> (x_out fp32 [B,K+1,D], y [B,K+1,D])."""
    B, N, D = x.shape
    x_out = torch.empty(B, K + 1, D, dtype=torch.float32, device=x.device)
    y = torch.empty(B, K + 1, D, dtype=torch.float32 if f32 else torch.bfloat16, device=x.device)
    wws = torch.empty(B, N - 1, dtype=torch.float32, device=x.device) if score_w is not None else None
    _lib.check(_lib.load().tr_cluster_merge_layernorm(
        _dev(x, torch.float32, "x"), _opt(score_w, torch.float32, "score_w"), _opt(score_b, torch.float32, "score_b"),
        None if wws is None else wws.data_ptr(), _dev(idx_cluster, torch.int32, "idx_cluster"), _dev(gamma, torch.float32, "gamma"),
        _dev(beta, torch.float32, "beta"), x_out.data_ptr(), y.data_ptr(), int(f32), B, N, K, D, eps, _stream()),
        "tr_cluster_merge_layernorm")
    return x_out, y


# ---------------------------------------------------------------------------------------- ATS (models/ats.py)
def ats_sample(cls_rows: torch.Tensor, qkv: torch.Tensor, mask, steps: torch.Tensor, K: int, want_cdf: bool = False):
    """AdaptiveTokenSampling (ats.py:52-84): cls_rows fp32 [B,H,N], qkv bf16|fp32 [B*N, 3*H*64], mask fp32 [B,N] of 1/0 or None
    -> (ids int32 [B,K] (CLS 0 first, 1-based ids, 0 padding), new_mask fp32 [B,K], cdf fp32 [B,N-1] | None)."""
    B, H, N = cls_rows.shape
    ids = torch.empty(B, K, dtype=torch.int32, device=qkv.device)
    new_mask = torch.empty(B, K, dtype=torch.float32, device=qkv.device)
    cdf = torch.empty(B, N - 1, dtype=torch.float32, device=qkv.device) if want_cdf else None
    _lib.check(_lib.load().tr_ats_sample(_dev(cls_rows, torch.float32, "cls_rows"), _dev(qkv, qkv.dtype, "qkv"),
                                         int(qkv.dtype == torch.float32), _opt(mask, torch.float32, "mask"),
                                         _dev(steps, torch.float32, "steps"), steps.numel(), ids.data_ptr(), new_mask.data_ptr(),
                                         None if cdf is None else cdf.data_ptr(), B, N, H, K, _stream()), "tr_ats_sample")
    return ids, new_mask, cdf


def ats_gather(x: torch.Tensor, ao: torch.Tensor, ids: torch.Tensor):
    """x fp32 [B,N,D], ao bf16|fp32 [B*N, D], ids int32 [B,K] -> (x_out fp32 [B,K,D], ao_out [B*K, D]) rows ids (ats.py:86,157)."""
    B, N, D = x.shape
    K = ids.shape[1]
    x_out = torch.empty(B, K, D, dtype=torch.float32, device=x.device)
    ao_out = torch.empty(B * K, D, dtype=ao.dtype, device=x.device)
    _lib.check(_lib.load().tr_ats_gather(_dev(x, torch.float32, "x"), _dev(ao, ao.dtype, "ao"), int(ao.dtype == torch.float32),
                                         _dev(ids, torch.int32, "ids"), x_out.data_ptr(), ao_out.data_ptr(), B, N, K, D, _stream()),
               "tr_ats_gather")
    return x_out, ao_out


# ---------------------------------------------------------------------------------------- Sinkhorn (models/sinkhorn.py)
def rownorm(x: torch.Tensor, f32: bool = False):
    """F.normalize(x, dim=-1): x fp32 [M,D] -> (xh fp32 [M,D], xh as GEMM operand bf16|fp32)."""
    M, D = x.shape
    xh = torch.empty_like(x)
    lp = torch.empty(M, D, dtype=torch.float32 if f32 else torch.bfloat16, device=x.device)
    _lib.check(_lib.load().tr_rownorm(_dev(x, torch.float32, "x"), xh.data_ptr(), lp.data_ptr(), int(f32), M, D, _stream()), "tr_rownorm")
    return xh, lp


def sinkhorn(scores: torch.Tensor, K: int, eps: float, iters: int, want_soft: bool = False):
    """log_optimal_transport (sinkhorn.py:41-56): scores fp32 [B,N,ldl] -> (wt fp32 [B,N,ldl] token-major plan, soft [B,K,N-1]|None)."""
    B, N, ldl = scores.shape
    wt = torch.zeros_like(scores)
    soft = torch.empty(B, K, N - 1, dtype=torch.float32, device=scores.device) if want_soft else None
    _lib.check(_lib.load().tr_sinkhorn(_dev(scores, torch.float32, "scores"), ldl, float(eps), int(iters), wt.data_ptr(),
                                       None if soft is None else soft.data_ptr(), B, N, K, _stream()), "tr_sinkhorn")
    return wt, soft


def weighted_merge(wt: torch.Tensor, x: torch.Tensor, src: torch.Tensor, K: int):
    """x_out[b,1+k] = sum_p wt[b,1+p,k] * src[b,1+p]; x_out[b,0] = x[b,0]  (sinkhorn.py:83)."""
    B, N, D = x.shape
    x_out = torch.empty(B, K + 1, D, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tr_weighted_merge(_dev(wt, torch.float32, "wt"), wt.shape[-1], _dev(x, torch.float32, "x"),
                                             _dev(src, torch.float32, "src"), x_out.data_ptr(), B, N, K, D, _stream()),
               "tr_weighted_merge")
    return x_out


# ---------------------------------------------------------------------------------------- K-Medoids (models/kmedoids.py)
def kmedoids(x: torch.Tensor, colsum_part: torch.Tensor, K: int, iters: int, fast_dist: bool = False):
    """k_medoids_fit (kmedoids.py:40-85, weighted branch): x fp32 [B,N,D], colsum_part fp32 [B,H,4,N] (previous block's attention)
    -> (centers int32 [B,K], assignment int32 [B,N-1])."""
    B, N, D = x.shape
    H = colsum_part.shape[1]
    lib = _lib.load()
    ws = torch.empty(lib.tr_dpcknn_workspace_floats(B, N), dtype=torch.float32, device=x.device)
    centers = torch.empty(B, K, dtype=torch.int32, device=x.device)
    assign = torch.empty(B, N - 1, dtype=torch.int32, device=x.device)
    _lib.check(lib.tr_kmedoids(_dev(x, torch.float32, "x"), _dev(colsum_part, torch.float32, "colsum_part"), ws.data_ptr(),
                               centers.data_ptr(), assign.data_ptr(), B, N, D, H, K, iters, int(fast_dist), _stream()), "tr_kmedoids")
    return centers, assign


def kmedoids_equal(x: torch.Tensor, first: int, K: int, iters: int, fast_dist: bool = False):
    """k_medoids_fit with token_weight None (args.equal_weight, kmedoids.py:43-58): `first` = the np.random.choice draw."""
    B, N, D = x.shape
    lib = _lib.load()
    ws = torch.empty(lib.tr_dpcknn_workspace_floats(B, N), dtype=torch.float32, device=x.device)
    centers = torch.empty(B, K, dtype=torch.int32, device=x.device)
    assign = torch.empty(B, N - 1, dtype=torch.int32, device=x.device)
    _lib.check(lib.tr_kmedoids_equal(_dev(x, torch.float32, "x"), int(first), ws.data_ptr(), centers.data_ptr(), assign.data_ptr(), B, N, D, K,
                                     iters, int(fast_dist), _stream()), "tr_kmedoids_equal")
    return centers, assign


# ---------------------------------------------------------------------------------------- DyViT training attention (forward)
def attention_policy(qkv: torch.Tensor, policy: torch.Tensor, B: int, N: int, H: int) -> torch.Tensor:
    """Policy_Attention with softmax_with_policy (dyvit.py:39-67): qkv bf16|fp32 [B*N, 3*H*64], policy fp32 [B,N] of 1/0 ->
    out [B*N, H*64] in qkv's dtype."""
    out = torch.empty(B * N, H * 64, dtype=qkv.dtype, device=qkv.device)
    lib = _lib.load()
    fn = lib.tr_attention_policy_f32 if qkv.dtype == torch.float32 else lib.tr_attention_policy_bf16
    _lib.check(fn(_dev(qkv, qkv.dtype, "qkv"), out.data_ptr(), _dev(policy, torch.float32, "policy"), B, N, H, _stream()),
               "tr_attention_policy")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# backward kernels (csrc/tr_backward.hip, csrc/tr_attention_bwd.hip): gradients of the ops above
def _ws(n_floats: int, dev) -> torch.Tensor:
    return torch.empty(max(int(n_floats), 4), dtype=torch.float32, device=dev)


def wgrad(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor = None, accumulate: bool = False, yskip: int = 0,
          rows: int = None) -> torch.Tensor:
    """nn.Linear weight gradient dW[N,K] (+)= dy[M,N]^T x[M,K] (bf16 operands, fp32 result)."""
    M = rows if rows is not None else x.shape[0]
    N, K = dy.shape[-1], x.shape[-1]
    lib = _lib.load()
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=x.device)
    nws = lib.tr_wgrad_workspace_floats(M, N, K)
    ws = _ws(nws, x.device)
    _lib.check(lib.tr_wgrad_bf16(_dev(dy, torch.bfloat16, "dy"), N, yskip, _dev(x, torch.bfloat16, "x"), K, _dev(out, torch.float32, "out"),
                                 int(accumulate), ws.data_ptr(), ws.numel(), M, N, K, _stream()), "tr_wgrad_bf16")
    return out


def linear_bwd_params(dy: torch.Tensor, x: torch.Tensor, accumulate: bool = False, dw: torch.Tensor = None, db: torch.Tensor = None,
                      yskip: int = 0):
    """Weight and bias gradient of an nn.Linear in one pass (tr_linear_bwd_params): (dW fp32 [N,K], db fp32 [N]).  yskip > 0: dy holds one
    extra leading row per `yskip` rows (the CLS row of [B, P + 1, D]) that the layer never saw (PatchEmbed)."""
    M, N, K = x.shape[0], dy.shape[-1], x.shape[-1]
    lib = _lib.load()
    dw = torch.empty(N, K, dtype=torch.float32, device=x.device) if dw is None else dw
    db = torch.empty(N, dtype=torch.float32, device=x.device) if db is None else db
    ws = _ws(lib.tr_wgrad_workspace_floats(M, N, K), x.device)

    def rows(t, name):      # 2-D operands may be column slices of a wider tensor (unit column stride, any row stride): the C ABI takes ldy / ldx
        if t.dim() == 2 and t.stride(1) == 1 and t.is_cuda and t.dtype == torch.bfloat16:
            return t.data_ptr(), t.stride(0)
        return _dev(t, torch.bfloat16, name), t.shape[-1]
    (py, ldy), (px, ldx) = rows(dy, "dy"), rows(x, "x")
    _lib.check(lib.tr_linear_bwd_params(py, ldy, yskip, px, ldx, _dev(dw, torch.float32, "dw"),
                                        _dev(db, torch.float32, "db"), int(accumulate), ws.data_ptr(), ws.numel(), M, N, K, _stream()),
               "tr_linear_bwd_params")
    return dw, db


def colsum(dy: torch.Tensor, out: torch.Tensor = None, accumulate: bool = False, yskip: int = 0, rows: int = None) -> torch.Tensor:
    """nn.Linear bias gradient db[N] (+)= sum_m dy[m,n]."""
    N = dy.shape[-1]
    M = rows if rows is not None else dy.numel() // N
    lib = _lib.load()
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=dy.device)
    ws = _ws(lib.tr_colsum_workspace_floats(M, N), dy.device)
    _lib.check(lib.tr_colsum_bf16(_dev(dy, torch.bfloat16, "dy"), N, yskip, _dev(out, torch.float32, "out"), int(accumulate),
                                  ws.data_ptr(), ws.numel(), M, N, _stream()), "tr_colsum_bf16")
    return out


def gelu(pre: torch.Tensor) -> torch.Tensor:
    h = torch.empty_like(pre)
    _lib.check(_lib.load().tr_gelu_bf16(_dev(pre, torch.bfloat16, "pre"), h.data_ptr(), pre.numel(), _stream()), "tr_gelu_bf16")
    return h


def gelu_bwd(pre: torch.Tensor, dh: torch.Tensor) -> torch.Tensor:
    """dh := dh * gelu'(pre) in place."""
    _lib.check(_lib.load().tr_gelu_bwd_bf16(_dev(pre, torch.bfloat16, "pre"), _dev(dh, torch.bfloat16, "dh"), pre.numel(), _stream()),
               "tr_gelu_bwd_bf16")
    return dh


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, eps: float, g_in: torch.Tensor = None, idx: torch.Tensor = None,
                  n_out: int = None, fused: bool = False):
    """LayerNorm backward fused with the residual gradient.  dy bf16 [M,D], x fp32 [M,D] (LayerNorm input), g_in fp32 [M,D]|None.
    Plain: returns (g fp32 [M,D], gb bf16 [M,D], dgamma, dbeta).  With idx int32 [B,K] (Top-K gather backward): rows are
    [B, K+1(+1 fused)], results scattered into zero-filled [B, n_out, D]; also returns g_fused [B,D] when fused."""
    M, D = dy.shape
    lib = _lib.load()
    dgamma = torch.empty(D, dtype=torch.float32, device=dy.device)
    dbeta = torch.empty(D, dtype=torch.float32, device=dy.device)
    ws = _ws(lib.tr_layernorm_bwd_workspace_floats(M, D), dy.device)
    g_fused = None
    if idx is None:
        g = torch.empty(M, D, dtype=torch.float32, device=dy.device)
        gb = torch.empty(M, D, dtype=torch.bfloat16, device=dy.device)
        K = n_in = no = 0
    else:
        B, K = idx.shape
        n_in = K + 1 + (1 if fused else 0)
        no = n_out
        g = torch.zeros(B * n_out, D, dtype=torch.float32, device=dy.device)
        gb = torch.zeros(B * n_out, D, dtype=torch.bfloat16, device=dy.device)
        if fused:
            g_fused = torch.empty(B, D, dtype=torch.float32, device=dy.device)
    _lib.check(lib.tr_layernorm_bwd(_dev(dy, torch.bfloat16, "dy"), _dev(x, torch.float32, "x"), D, _dev(gamma, torch.float32, "gamma"),
                                    _opt(g_in, torch.float32, "g_in"), D, g.data_ptr(), D, gb.data_ptr(), _opt(idx, torch.int32, "idx"), K, n_in,
                                    no, None if g_fused is None else g_fused.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), 0,
                                    ws.data_ptr(), ws.numel(), M, D, eps, _stream()), "tr_layernorm_bwd")
    if idx is not None and fused:
        return g, gb, dgamma, dbeta, g_fused
    return g, gb, dgamma, dbeta


def attention_bwd(qkv: torch.Tensor, dout: torch.Tensor, B: int, N: int, H: int, size: torch.Tensor = None, dcls: torch.Tensor = None):
    """d qkv (bf16 [B*N, 3*H*64]) of softmax(q k^T / 8 [+ log size]) v given d out (bf16 [B*N, H*64])."""
    dqkv = torch.empty_like(qkv)
    _lib.check(_lib.load().tr_attention_bwd_bf16(_dev(qkv, torch.bfloat16, "qkv"), _dev(dout, torch.bfloat16, "dout"),
                                                 _opt(size, torch.float32, "size"), _opt(dcls, torch.float32, "dcls"), dqkv.data_ptr(), B, N,
                                                 H, _stream()), "tr_attention_bwd_bf16")
    return dqkv


def head_bwd(dlogits: torch.Tensor, w: torch.Tensor, xn: torch.Tensor):
    B, Cc = dlogits.shape
    D = w.shape[1]
    dxn = torch.empty(B, D, dtype=torch.bfloat16, device=w.device)
    dw = torch.empty(Cc, D, dtype=torch.float32, device=w.device)
    db = torch.empty(Cc, dtype=torch.float32, device=w.device)
    lib = _lib.load()
    dl16 = torch.empty(B, Cc, dtype=torch.bfloat16, device=w.device)
    ws = _ws(lib.tr_wgrad_workspace_floats(B, Cc, D), w.device)
    _lib.check(lib.tr_head_bwd(_dev(dlogits, torch.float32, "dlogits"), _dev(w, torch.bfloat16, "w"), _dev(xn, torch.bfloat16, "xn"),
                               dxn.data_ptr(), dw.data_ptr(), db.data_ptr(), 0, dl16.data_ptr(), ws.data_ptr(), ws.numel(), B, Cc, D, _stream()),
               "tr_head_bwd")
    return dxn, dw, db


def embed_bwd(g: torch.Tensor):
    B, N, D = g.shape
    dpos = torch.empty(N, D, dtype=torch.float32, device=g.device)
    dcls = torch.empty(D, dtype=torch.float32, device=g.device)
    _lib.check(_lib.load().tr_embed_bwd(_dev(g, torch.float32, "g"), dpos.data_ptr(), dcls.data_ptr(), 0, B, N, D, _stream()), "tr_embed_bwd")
    return dpos, dcls


def evit_fuse_bwd(x, delta, compl, scores, g_fused, g_out, gb_out):
    """In place on g_out / gb_out ([B,N,D]); returns dscore fp32 [B,N] (gradient wrt the head-mean CLS attention, entry 1+token)."""
    B, N, D = x.shape
    K = N - 1 - compl.shape[1]
    dscore = torch.zeros(B, N, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tr_evit_fuse_bwd(_dev(x, torch.float32, "x"), _opt(delta, torch.bfloat16, "delta"), _dev(compl, torch.int32, "compl"),
                                            _dev(scores, torch.float32, "scores"), _dev(g_fused, torch.float32, "g_fused"),
                                            _dev(g_out, torch.float32, "g_out"), _dev(gb_out, torch.bfloat16, "gb_out"), dscore.data_ptr(),
                                            B, N, K, D, _stream()), "tr_evit_fuse_bwd")
    return dscore


def tome_merge_bwd(g_merged, size_in, size_out, unm, src, dst, N: int):
    B, N_out, D = g_merged.shape
    r = N - N_out
    g = torch.empty(B, N, D, dtype=torch.float32, device=g_merged.device)
    gb = torch.empty(B, N, D, dtype=torch.bfloat16, device=g_merged.device)
    inv = torch.empty(B, N, dtype=torch.int32, device=g_merged.device)
    _lib.check(_lib.load().tr_tome_merge_bwd(_dev(g_merged, torch.float32, "g_merged"), _opt(size_in, torch.float32, "size_in"),
                                             _dev(size_out, torch.float32, "size_out"), _dev(unm, torch.int32, "unm"), _dev(src, torch.int32, "src"),
                                             _dev(dst, torch.int32, "dst"), inv.data_ptr(), g.data_ptr(), gb.data_ptr(), B, N, r, D, _stream()),
               "tr_tome_merge_bwd")
    return g, gb


def cluster_merge_bwd(g_in, x0, x1, wtok, assign, score_w):
    """DPC-KNN CTM backward.  Returns (g fp32 [B,N,D], gb bf16, d_sw [D]|None, d_sb [1]|None)."""
    B, N, D = x0.shape
    K = x1.shape[1] - 1
    lib = _lib.load()
    g = torch.empty(B, N, D, dtype=torch.float32, device=x0.device)
    gb = torch.empty(B, N, D, dtype=torch.bfloat16, device=x0.device)
    dsw = torch.empty(D, dtype=torch.float32, device=x0.device) if score_w is not None else None
    dsb = torch.empty(1, dtype=torch.float32, device=x0.device) if score_w is not None else None
    ws = _ws((8 * B + 1) * (D + 4), x0.device)
    _lib.check(lib.tr_cluster_merge_bwd(_dev(g_in, torch.float32, "g_in"), _dev(x0, torch.float32, "x0"), _dev(x1, torch.float32, "x1"),
                                        _opt(wtok, torch.float32, "wtok"), _dev(assign, torch.int32, "assign"), _opt(score_w, torch.float32, "score_w"),
                                        g.data_ptr(), gb.data_ptr(), None if dsw is None else dsw.data_ptr(), None if dsb is None else dsb.data_ptr(),
                                        0, ws.data_ptr(), ws.numel(), B, N, K, D, _stream()), "tr_cluster_merge_bwd")
    return g, gb, dsw, dsb


def ats_scatter(g, dao_s, ids, N: int):
    B, Ks, D = g.shape
    g_full = torch.zeros(B, N, D, dtype=torch.float32, device=g.device)
    dao_full = torch.zeros(B, N, D, dtype=torch.bfloat16, device=g.device)
    _lib.check(_lib.load().tr_ats_scatter(_dev(g, torch.float32, "g"), _dev(dao_s, torch.bfloat16, "dao_s"), _dev(ids, torch.int32, "ids"),
                                          g_full.data_ptr(), dao_full.data_ptr(), B, N, Ks, D, _stream()), "tr_ats_scatter")
    return g_full, dao_full


# ---------------------------------------------------------------------------------------- soft-assignment reducers, backward (csrc/tr_soft_bwd.hip)
def soft_merge_bwd(g: torch.Tensor, wt: torch.Tensor, src: torch.Tensor):
    """out[b,1+k] = sum_p wt[b,1+p,k] src[b,1+p] backwards: g fp32 [B,K+1,D], wt fp32 [B,N,ldl], src fp32 [B,N,D] ->
    (dwt fp32 [B,N,ldl] (CLS rows / columns >= K zero), dsrc fp32 [B,N,D] (CLS rows zero))."""
    B, N, D = src.shape
    K, ldl = g.shape[1] - 1, wt.shape[2]
    lib = _lib.load()
    dwt = torch.zeros(B, N, ldl, dtype=torch.float32, device=src.device)
    dsrc = torch.zeros(B, N, D, dtype=torch.float32, device=src.device)
    _lib.check(lib.tr_soft_dweights(_dev(g, torch.float32, "g"), _dev(src, torch.float32, "src"), dwt.data_ptr(), ldl, B, N, K, D, _stream()),
               "tr_soft_dweights")
    _lib.check(lib.tr_soft_dsrc(_dev(g, torch.float32, "g"), _dev(wt, torch.float32, "wt"), ldl, dsrc.data_ptr(), B, N, K, D, _stream()), "tr_soft_dsrc")
    return dwt, dsrc


def token_softmax_bwd(wt: torch.Tensor, dwt: torch.Tensor, logits: torch.Tensor, scale: float, K: int, want_dscale: bool = False):
    """-> (ds bf16 [B,N,ld64], dscale fp32[1] | None)."""
    B, N, ldl = wt.shape
    ldo = (K + 63) // 64 * 64
    lib = _lib.load()
    ds = torch.zeros(B, N, ldo, dtype=torch.bfloat16, device=wt.device)
    dscale = torch.zeros(1, dtype=torch.float32, device=wt.device) if want_dscale else None
    ws = _ws(lib.tr_token_softmax_bwd_workspace_floats(B, K), wt.device)
    _lib.check(lib.tr_token_softmax_bwd(_dev(wt, torch.float32, "wt"), _dev(dwt, torch.float32, "dwt"), _dev(logits, torch.float32, "logits"), ldl,
                                        float(scale), ds.data_ptr(), ldo, None if dscale is None else dscale.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                                        B, N, K, _stream()), "tr_token_softmax_bwd")
    return ds, dscale


def sinkhorn_bwd(scores: torch.Tensor, dplan: torch.Tensor, K: int, eps: float, iters: int) -> torch.Tensor:
    B, N, ldl = scores.shape
    ldo = (K + 63) // 64 * 64
    ds = torch.zeros(B, N, ldo, dtype=torch.bfloat16, device=scores.device)
    _lib.check(_lib.load().tr_sinkhorn_bwd(_dev(scores, torch.float32, "scores"), _dev(dplan, torch.float32, "dplan"), ldl, float(eps), iters,
                                           ds.data_ptr(), ldo, B, N, K, _stream()), "tr_sinkhorn_bwd")
    return ds


def rownorm_bwd(x: torch.Tensor, da: torch.Tensor, db: torch.Tensor = None) -> torch.Tensor:
    M, D = x.shape
    dx = torch.empty(M, D, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tr_rownorm_bwd(_dev(x, torch.float32, "x"), _dev(da, torch.float32, "da"), _opt(db, torch.bfloat16, "db"), dx.data_ptr(), M, D,
                                          _stream()), "tr_rownorm_bwd")
    return dx


def add_into_bf16(a: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    _lib.check(_lib.load().tr_add_into_bf16(_dev(a, torch.float32, "a"), _dev(y, torch.bfloat16, "y"), a.numel(), _stream()), "tr_add_into_bf16")
    return y


def attention_policy_bwd(qkv: torch.Tensor, dout: torch.Tensor, policy: torch.Tensor, B: int, N: int, H: int):
    """Backward of attention_policy (bf16): -> (dqkv bf16 [B*N, 3*H*64], d policy fp32 [B,H,N] per head).  Beyond 224 tokens the
    key-blocked kernels (tr_attention_policy_bwd_long_bf16), as in the training executor."""
    lib = _lib.load()
    dqkv = torch.empty_like(qkv)
    dpart = torch.zeros(B, H, N, dtype=torch.float32, device=qkv.device)
    args = (_dev(qkv, torch.bfloat16, "qkv"), _dev(dout, torch.bfloat16, "dout"), _dev(policy, torch.float32, "policy"), dqkv.data_ptr(), dpart.data_ptr())
    if N > 224:
        ws = _ws(lib.tr_attention_bwd_long_workspace_floats(B, N, H), qkv.device)
        _lib.check(lib.tr_attention_policy_bwd_long_bf16(*args, ws.data_ptr(), ws.numel(), B, N, H, _stream()), "tr_attention_policy_bwd_long_bf16")
    else:
        _lib.check(lib.tr_attention_policy_bwd_bf16(*args, B, N, H, _stream()), "tr_attention_policy_bwd_bf16")
    return dqkv, dpart


def attention_bwd_long(qkv: torch.Tensor, dout: torch.Tensor, B: int, N: int, H: int, size: torch.Tensor = None, dcls: torch.Tensor = None):
    """tr_attention_bwd_bf16's gradient for any sequence length (key-blocked kernels; what the training executor runs beyond 224 tokens)."""
    lib = _lib.load()
    dqkv = torch.empty_like(qkv)
    ws = _ws(lib.tr_attention_bwd_long_workspace_floats(B, N, H), qkv.device)
    _lib.check(lib.tr_attention_bwd_long_bf16(_dev(qkv, torch.bfloat16, "qkv"), _dev(dout, torch.bfloat16, "dout"), _opt(size, torch.float32, "size"),
                                              _opt(dcls, torch.float32, "dcls"), dqkv.data_ptr(), ws.data_ptr(), ws.numel(), B, N, H, _stream()),
               "tr_attention_bwd_long_bf16")
    return dqkv


def gemm_gelu_keep(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor):
    """Training forward of Linear + GELU in one launch: (pre bf16 [M,N], h = gelu(pre) bf16 [M,N])."""
    M, K = a.shape
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or bias.numel() != N:
        raise ValueError(f"gemm_gelu_keep: a {tuple(a.shape)}, w {tuple(w.shape)}, bias {tuple(bias.shape)} do not form a Linear")
    _same_device(a, w, bias)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    h = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    _lib.check(_lib.load().tr_gemm_gelu_keep_bf16(_dev(a, torch.bfloat16, "a"), _dev(w, torch.bfloat16, "w"), _dev(bias, torch.float32, "bias"),
                                                  pre.data_ptr(), h.data_ptr(), M, N, K, _stream(a)), "tr_gemm_gelu_keep_bf16")
    return pre, h


def gemm_dgelu(a: torch.Tensor, w: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    """Data gradient of a Linear with the GELU backward of the layer below folded in: bf16(a w^T) * gelu'(pre), bf16 [M,N]."""
    M, K = a.shape
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or tuple(pre.shape) != (M, N):
        raise ValueError(f"gemm_dgelu: a {tuple(a.shape)}, w {tuple(w.shape)}, pre {tuple(pre.shape)} do not fit")
    _same_device(a, w, pre)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    _lib.check(_lib.load().tr_gemm_dgelu_bf16(_dev(a, torch.bfloat16, "a"), _dev(w, torch.bfloat16, "w"), _dev(pre, torch.bfloat16, "pre"),
                                              out.data_ptr(), M, N, K, _stream(a)), "tr_gemm_dgelu_bf16")
    return out


def linear_bwd_params2(dy0: torch.Tensor, x0: torch.Tensor, dy1: torch.Tensor, x1: torch.Tensor, accumulate: bool = False, outs=None):
    """Weight and bias gradients of TWO Linear layers in one weight-gradient launch + one reduce (tr_linear_bwd_params2):
    ((dW0, db0), (dW1, db1)).  outs: the four destination tensors when accumulating."""
    lib = _lib.load()
    (M0, N0), K0 = dy0.shape, x0.shape[-1]
    (M1, N1), K1 = dy1.shape, x1.shape[-1]
    dev = x0.device
    if outs is None:
        outs = (torch.empty(N0, K0, device=dev), torch.empty(N0, device=dev), torch.empty(N1, K1, device=dev), torch.empty(N1, device=dev))
    ws = _ws(lib.tr_linear_bwd_params2_workspace_floats(M0, N0, K0, M1, N1, K1), dev)
    _lib.check(lib.tr_linear_bwd_params2(_dev(dy0, torch.bfloat16, "dy0"), N0, _dev(x0, torch.bfloat16, "x0"), K0, _dev(outs[0], torch.float32, "dw0"),
                                         _dev(outs[1], torch.float32, "db0"), M0, N0, K0, _dev(dy1, torch.bfloat16, "dy1"), N1,
                                         _dev(x1, torch.bfloat16, "x1"), K1, _dev(outs[2], torch.float32, "dw1"), _dev(outs[3], torch.float32, "db1"),
                                         M1, N1, K1, int(accumulate), ws.data_ptr(), ws.numel(), _stream()), "tr_linear_bwd_params2")
    return (outs[0], outs[1]), (outs[2], outs[3])


def linear_bwd_group(layers, accumulate: bool = False, outs=None):
    """Weight and bias gradients of up to four Linear layers in ONE weight-gradient launch + one reduce (tr_linear_bwd_group).
    layers: [(dy bf16 [M_i, N_i], x bf16 [M_i, K_i]), ...]; returns [(dW_i fp32 [N_i, K_i], db_i fp32 [N_i]), ...]; outs: the same list
    of destination tensors when accumulating."""
    import ctypes
    lib = _lib.load()
    n = len(layers)
    if not 1 <= n <= 4:
        raise ValueError("linear_bwd_group: 1..4 layers")
    dev = layers[0][0].device
    if outs is None:
        outs = [(torch.empty(dy.shape[1], x.shape[1], device=dev), torch.empty(dy.shape[1], device=dev)) for dy, x in layers]
    arr = (_lib.TrLinearGrad * n)()
    for i, ((dy, x), (dw, db)) in enumerate(zip(layers, outs)):
        if dy.shape[0] != x.shape[0] or tuple(dw.shape) != (dy.shape[1], x.shape[1]) or db.numel() != dy.shape[1]:
            raise ValueError(f"linear_bwd_group: layer {i}: dy {tuple(dy.shape)}, x {tuple(x.shape)}, dw {tuple(dw.shape)}, db {tuple(db.shape)} do not fit")
        arr[i] = _lib.TrLinearGrad(_dev(dy, torch.bfloat16, "dy"), dy.shape[1], _dev(x, torch.bfloat16, "x"), x.shape[1], _dev(dw, torch.float32, "dw"),
                                   _dev(db, torch.float32, "db"), dy.shape[0], dy.shape[1], x.shape[1])
    ws = _ws(lib.tr_linear_bwd_group_workspace_floats(ctypes.cast(arr, ctypes.c_void_p), n), dev)
    _lib.check(lib.tr_linear_bwd_group(ctypes.cast(arr, ctypes.c_void_p), n, int(accumulate), ws.data_ptr(), ws.numel(), _stream()), "tr_linear_bwd_group")
    return outs
