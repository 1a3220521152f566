"""Oracle: Adaptive Token Sampling (models/ats.py).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
torch-CPU fp32, functional; reference = /root/reference (read-only):

  models/ats.py   AdaptiveTokenSampling :44-89, ATSAttention.forward :110-134, ATSBlock.forward :152-161,
                  ATSVisionTransformer.__init__ :199-215 / forward :232-271

No extra parameters.  Token counts are data dependent in the reference (pad to the batch maximum of unique samples,
ats.py:78); `static_pad=True` pads every sampling block to its bound K instead -- what the HIP path does.  Padded rows
are masked keys with exactly zero softmax weight (masked_fill(-finfo.max) underflows to 0), so valid rows and the logits
are identical in both layouts; tests check that on the CPU too.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pad_sequence

from .vit import dropout_site, VitConfig, _r, embed_tokens, head, layer_norm, mlp, patch_embed, round_bf16

Tensor = torch.Tensor


def ats_sample_counts(cfg: VitConfig) -> Dict[int, int]:
    """ats.py:204-205: one keep_rate -> int(kr**(i+1) * P0) + 1 per stage; several -> ABSOLUTE counts verbatim."""
    counts = list(cfg.keep_rate)
    loc = list(cfg.reduction_loc)
    if len(counts) == 1:
        counts = [int(counts[0] ** (i + 1) * cfg.num_patches) + 1 for i in range(len(loc))]
    assert len(counts) == len(loc), "keep_rate / reduction_loc length mismatch"
    return {int(l): int(c) for c, l in zip(counts, loc)}


def ats_token_bounds(cfg: VitConfig) -> Dict[int, int]:
    """Most tokens (CLS included) a sampling block can keep: one per grid point + the CLS token.  The grid of ats.py:48 is a float
    `torch.arange` with an exclusive end; for 42 of the sample counts up to 197 (7, 12, 14, 19, 27, ..., 126, ...) rounding lets the
    end point in and the grid has K points instead of K - 1, so the bound is K + 1 there (the reference's dynamic shapes do not care)."""
    return {blk: int(ats_sample_steps(c).numel()) + 1 for blk, c in ats_sample_counts(cfg).items() if c}


def ats_sample_steps(sample_count: int) -> Tensor:
    """ats.py:48 verbatim: K-1 points of the inverse-CDF grid."""
    return torch.arange(1 / (2 * sample_count), (2 * sample_count - 1) / (2 * sample_count), 2 / (2 * sample_count))


def ats_scores(attn_cls: Tensor, v: Tensor, eps: float = 1e-6) -> Tensor:
    """ats.py:53-67: significance score of the patch tokens, normalised to sum 1.  attn_cls [B,H,N] (CLS query row),
    v [B,H,N,dh] -> [B,N-1]."""
    value_norms = v[:, :, 1:, :].norm(dim=-1)
    sig = torch.sum(attn_cls[:, :, 1:] * value_norms, dim=1)
    return sig / (sig.sum(dim=-1, keepdim=True) + eps)


def ats_cdf(normed: Tensor, mask: Tensor) -> Tensor:
    """ats.py:69-70: running sum of the normalised scores; masked (padded) positions are pushed away by +0.1."""
    cdf = normed.cumsum(dim=1)
    cdf[mask[:, 1:] == False] += 0.1          # noqa: E712
    return cdf


def ats_ids_from_cdf(cdf: Tensor, steps: Tensor, pad_to: Optional[int] = None):
    """ats.py:73-84: nearest cdf entry per grid point (torch.cdist -- its matmul form |a|^2+|b|^2-2ab since P > 25, whose
    rounding decides among candidates closer than ~3e-4 to a grid point), per-image sorted unique, zero padding, CLS id 0 in
    front.  Returns (ids [B,K'], new_mask [B,K'])."""
    dist = torch.cdist(steps.unsqueeze(0).unsqueeze(2), cdf.unsqueeze(2))
    sampled = torch.argmin(dist, dim=-1) + 1
    uniq = [torch.unique(t, sorted=True) for t in torch.unbind(sampled)]
    ids = pad_sequence(uniq, batch_first=True)
    if pad_to is not None:
        ids = F.pad(ids, (0, pad_to - 1 - ids.shape[1]), value=0)
    new_mask = F.pad(ids != 0, (1, 0), value=True)
    return F.pad(ids, (1, 0), value=0), new_mask


def ats_sample_ids(normed: Tensor, mask: Tensor, steps: Tensor, pad_to: Optional[int] = None):
    return ats_ids_from_cdf(ats_cdf(normed, mask), steps, pad_to)


def ats_block_forward(x: Tensor, mask: Tensor, p: Dict[str, Tensor], i: int, cfg: VitConfig, sample_count: int,
                      precision: str = "fp32", static_pad: bool = False, forced_ids: Optional[Tensor] = None):
    """ATSBlock.forward ats.py:152-161 over ATSAttention.forward ats.py:110-134.  Returns (x, mask, ids|None, cdf|None).
    forced_ids (tests only): use these sampled ids [B,K'] (CLS id 0 first, 0 = pad) instead of the block's own."""
    pre = f"blocks.{i}."
    B, N, D = x.shape
    H = cfg.num_heads
    dh = D // H
    xn = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps, precision)
    qkv = _r(xn @ _r(p[pre + "attn.qkv.weight"], precision).t() + p[pre + "attn.qkv.bias"], precision)
    q, k, v = qkv.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4).unbind(0)
    dots = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    dots_mask = mask.unsqueeze(1).unsqueeze(3) * mask.unsqueeze(1).unsqueeze(2)
    dots = dots.masked_fill(~dots_mask, -torch.finfo(dots.dtype).max)
    if precision == "bf16":                   # un-normalised probabilities rounded to bf16 before P.V, like the HIP kernel
        e = torch.exp(dots - dots.amax(dim=-1, keepdim=True))
        attn = e / e.sum(dim=-1, keepdim=True)
        pv = lambda rows: (round_bf16(e[:, :, rows] if rows is not None else e) @ v)          # noqa: E731
    else:
        attn = dots.softmax(dim=-1)
    ids = cdf = None
    if sample_count:
        with torch.no_grad():      # the sampled ids are integers (argmin, ats.py:74): nothing of this reaches a gradient
            cdf = ats_cdf(ats_scores(attn[:, :, 0, :].detach(), v.detach()), mask)
        if forced_ids is None:
            steps = ats_sample_steps(sample_count)
            ids, mask = ats_ids_from_cdf(cdf, steps, int(steps.numel()) + 1 if static_pad else None)
        else:
            ids, mask = forced_ids, F.pad(forced_ids[:, 1:] != 0, (1, 0), value=True)
        gi = ids[:, None, :, None].expand(B, H, ids.shape[1], N)
        attn_rows = torch.gather(attn, 2, gi)                 # batched_index_select(attn, ids, dim=2), ats.py:86
        if precision == "bf16":
            e_rows = torch.gather(e, 2, gi)
            o = (round_bf16(e_rows) @ v) / e_rows.sum(dim=-1, keepdim=True)
        else:
            o = attn_rows @ v
        x = torch.gather(x, 1, ids[:, :, None].expand(B, ids.shape[1], D))                     # ats.py:157
    else:
        o = (round_bf16(e) @ v) / e.sum(dim=-1, keepdim=True) if precision == "bf16" else attn @ v
    n_out = o.shape[2]
    o = _r(o.transpose(1, 2).reshape(B, n_out, D), precision)
    x = x + dropout_site(_r(o @ _r(p[pre + "attn.proj.weight"], precision).t() + p[pre + "attn.proj.bias"], precision), precision)
    xn2 = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps, precision)
    x = x + mlp(xn2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"],
                precision)
    return x, mask, ids, cdf


@torch.no_grad()
def ats_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32", return_viz: bool = False,
                static_pad: bool = False, forced: Optional[Dict[int, Tensor]] = None):
    """ATSVisionTransformer.forward ats.py:232-271, eval mode."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    counts = ats_sample_counts(cfg)
    mask = torch.ones(h.shape[0], h.shape[1], dtype=torch.bool)
    viz = {"Kept_Tokens": {}, "Tokens": {}, "Masks": {}, "Cdf": {}}
    for i in range(cfg.depth):
        h, mask, ids, cdf = ats_block_forward(h, mask, p, i, cfg, counts.get(i, 0), precision, static_pad,
                                              None if forced is None else forced.get(i))
        viz["Tokens"][i] = h.shape[1]
        if ids is not None:
            viz["Kept_Tokens"][i] = (ids[:, 1:] - 1).numpy()          # ats.py:253
            viz["Cdf"][i] = cdf
            viz["Masks"][i] = mask
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits
