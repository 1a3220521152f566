"""Oracle: DeiT trunk + Top-K / EViT token reduction, torch-CPU fp32, functional.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Reference = /root/reference (read-only):
  models/deit_viz.py   -- in-tree copy of timm-0.4.12 Attention/Block/VisionTransformer
  models/topk.py       -- Top-K pruning inside the block
  models/evit.py       -- Top-K + fused "extra" token
  models_act.py        -- factory dims (192/3, 384/6, 768/12; depth 12; LN eps 1e-6)

`params` is a flat dict of tensors with the reference's state-dict key names
(SURVEY.md section 8b): cls_token, pos_embed, patch_embed.proj.{weight,bias},
blocks.{i}.{norm1,norm2}.{weight,bias}, blocks.{i}.attn.{qkv,proj}.{weight,bias},
blocks.{i}.mlp.{fc1,fc2}.{weight,bias}, norm.{weight,bias}, head.{weight,bias}.

precision="fp32" is the reference arithmetic.  precision="bf16" re-states the SAME
algorithm with the rounding points of the HIP pipeline (bf16 GEMM operands, fp32
accumulate, fp32 residual stream; DESIGN.md "Numerics") so GPU results can be
compared tightly; it is still a CPU checker, not a product path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def round_bf16(t: Tensor) -> Tensor:
    """Round-to-nearest-even to bf16 and back to fp32 (what `v_cvt_pk_bf16_f32` does)."""
    return t.to(torch.bfloat16).to(torch.float32)


def _r(t: Tensor, precision: str) -> Tensor:
    return round_bf16(t) if precision == "bf16" else t


@dataclass
class VitConfig:
    family: str = "topk"                 # "deit" | "topk" | "evit"
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_classes: int = 1000
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    mlp_ratio: float = 4.0
    keep_rate: List[float] = field(default_factory=lambda: [1.0])
    reduction_loc: List[int] = field(default_factory=list)
    ln_eps: float = 1e-6                 # models_act.py:1121  partial(nn.LayerNorm, eps=1e-6)

    @property
    def num_patches(self) -> int:
        g = self.img_size // self.patch_size
        return g * g


def stage_keep_counts(cfg: VitConfig) -> Dict[int, int]:
    """block index -> number of patch tokens kept by that block's Top-K.

    topk.py:141-150 / evit.py:171-180: one keep_rate -> geometric kr**(i+1) per stage,
    else used verbatim; topk.py:40,56: K = int(ratio * 14*14) with 196 HARD-CODED
    (independent of img_size).  Ratios of exactly 1 mean "no reduction" (topk.py:55).
    """
    ratios = list(cfg.keep_rate)
    loc = list(cfg.reduction_loc)
    if len(ratios) == 1:
        ratios = [ratios[0] ** (i + 1) for i in range(len(loc))]
    assert len(ratios) == len(loc), "keep_rate / reduction_loc length mismatch"
    out = {}
    for r, l in zip(ratios, loc):
        assert 0 < r <= 1
        if r < 1:
            out[int(l)] = int(r * 196)
    return out


# --------------------------------------------------------------------------- trunk
def patch_embed(x: Tensor, w: Tensor, b: Tensor, patch: int, precision: str = "fp32") -> Tensor:
    """timm PatchEmbed (call site topk.py:181): Conv2d(k=s=patch) -> flatten(2).transpose(1,2).

    Written as im2col + GEMM (identical arithmetic): row (b,py,px), col (c,iy,ix).
    """
    B, C, H, W = x.shape
    gh, gw = H // patch, W // patch
    cols = x.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * patch * patch)
    wm = w.reshape(w.shape[0], -1)
    return _r(cols, precision) @ _r(wm, precision).t() + b


# Training-mode nn.Dropout (timm's drop_rate: pos_drop topk.py:186, proj_drop :53, the Mlp's two nn.Dropout) REPLAYED with recorded keep
# masks: inside `with dropout_replay(masks, p):` every dropout site of the trunk takes the next mask (in the reference module's call
# order) and scales the survivors by 1 / (1 - p), like F.dropout.  Outside the context the sites are the identity (eval).
_DROPOUT = None


class dropout_replay:
    def __init__(self, masks, p: float):
        self.masks, self.p = list(masks), float(p)

    def __enter__(self):
        global _DROPOUT
        _DROPOUT = (iter(self.masks), self.p, self)
        self.used = 0
        return self

    def __exit__(self, *exc):
        global _DROPOUT
        _DROPOUT = None
        if exc[0] is None and self.used != len(self.masks):
            raise AssertionError(f"dropout_replay: {self.used} of {len(self.masks)} recorded masks consumed")


def dropout_site(x: Tensor, precision: str = "fp32") -> Tensor:
    if _DROPOUT is None:
        return x
    it, p, ctx = _DROPOUT
    m = torch.as_tensor(next(it)).to(x.dtype).reshape(x.shape)
    ctx.used += 1
    return _r(x * m * (1.0 / (1.0 - p)), precision)


def embed_tokens(tok: Tensor, cls_token: Tensor, pos_embed: Tensor) -> Tensor:
    """topk.py:183-186: cat(cls.expand, x) + pos_embed; pos_drop is identity in eval (training: dropout_replay)."""
    B = tok.shape[0]
    return dropout_site(torch.cat((cls_token.expand(B, -1, -1), tok), dim=1) + pos_embed)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float, precision: str = "fp32") -> Tensor:
    """nn.LayerNorm(D, eps=1e-6): biased variance, fp32 statistics."""
    return _r(F.layer_norm(x, (x.shape[-1],), w, b, eps), precision)


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default (exact erf form) used by timm Mlp."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def attention(xn: Tensor, qkv_w: Tensor, qkv_b: Tensor, proj_w: Tensor, proj_b: Tensor,
              num_heads: int, precision: str = "fp32") -> Tuple[Tensor, Tensor]:
    """deit_viz.py:41-53 == topk.py:42-53 == evit.py:64-75.

    Returns (proj(attn @ v) [B,N,D], cls_rows [B,H,N]) where cls_rows[b,h,:] is the
    softmax row of the CLS query -- the only part of `attn` the reduction reads
    (topk.py:59).  QKV weight rows are [q(all heads); k; v], head-major inside each
    (reshape [B,N,3,H,dh] at topk.py:44).
    """
    B, N, D = xn.shape
    dh = D // num_heads
    scale = dh ** -0.5
    qkv = _r(xn @ _r(qkv_w, precision).t() + qkv_b, precision)
    qkv = qkv.reshape(B, N, 3, num_heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-2, -1)) * scale
    if precision == "bf16":
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        o = (round_bf16(p) @ v) / l
        attn = p / l
    else:
        attn = s.softmax(dim=-1)
        o = attn @ v
    o = _r(o.transpose(1, 2).reshape(B, N, D), precision)
    out = _r(o @ _r(proj_w, precision).t() + proj_b, precision)   # bf16 mode: the Linear output is stored in bf16
    return dropout_site(out, precision), attn[:, :, 0, :]


def mlp(xn: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, precision: str = "fp32") -> Tensor:
    """timm Mlp: fc1 -> GELU(erf) -> fc2 (dropouts are identity in eval)."""
    h = dropout_site(_r(gelu_erf(xn @ _r(w1, precision).t() + b1), precision), precision)
    return dropout_site(_r(h @ _r(w2, precision).t() + b2, precision), precision)


def head(x: Tensor, norm_w: Tensor, norm_b: Tensor, head_w: Tensor, head_b: Tensor,
         eps: float, precision: str = "fp32") -> Tensor:
    """topk.py:201-203: norm -> x[:,0] -> pre_logits(Identity) -> head.  LN is per-row,
    so only row 0 needs normalising."""
    c = layer_norm(x[:, 0], norm_w, norm_b, eps, precision)
    return c @ _r(head_w, precision).t() + head_b


# --------------------------------------------------------------------------- reduction ops
def cls_scores_from_heads(cls_rows: Tensor) -> Tensor:
    """topk.py:59-60: cls_attn = attn[:, :, 0, 1:].mean(dim=1) -> [B, N-1] fp32."""
    return cls_rows[:, :, 1:].mean(dim=1)


def cls_topk_select(scores: Tensor, k: int) -> Tensor:
    """topk.py:61 torch.topk(cls_attn, K, dim=1, largest=True, sorted=True) -> idx [B,K] int64,
    in DESCENDING-score order.  Tie rule of this build (torch's CPU tie order is
    unspecified): equal scores -> lowest index first.  Golden
    vectors are tie-free, so the rule never decides a pinned case."""
    order = torch.sort(scores, dim=1, descending=True, stable=True).indices
    return order[:, :k].contiguous()


def gather_compact(x: Tensor, idx: Tensor) -> Tensor:
    """topk.py:89-93: x = cat(x[:,0:1], gather(x[:,1:], 1, idx)) -- token order = score order."""
    B, N, D = x.shape
    others = torch.gather(x[:, 1:], 1, idx.unsqueeze(-1).expand(-1, -1, D))
    return torch.cat([x[:, 0:1], others], dim=1)


def complement_idx(idx: Tensor, dim: int) -> Tensor:
    """evit.py:25-46: indices of range(dim) NOT in idx, ascending -> [B, dim-K] int64."""
    B, K = idx.shape
    keep = torch.ones(B, dim, dtype=torch.bool)
    keep.scatter_(1, idx, False)
    ar = torch.arange(dim).expand(B, dim)
    return ar[keep].reshape(B, dim - K)


def evit_fuse(x: Tensor, idx: Tensor, scores: Tensor) -> Tuple[Tensor, Tensor]:
    """evit.py:111-123: extra = sum_j x_nonTopK[j] * cls_attn[j] (un-normalised weights);
    x = cat(cls, x_topk, extra).  Returns (x [B,K+2,D], compl [B,P-K])."""
    B, N, D = x.shape
    non_cls = x[:, 1:]
    compl = complement_idx(idx, N - 1)
    non_topk = torch.gather(non_cls, 1, compl.unsqueeze(-1).expand(-1, -1, D))
    w = torch.gather(scores, 1, compl)
    extra = torch.sum(non_topk * w.unsqueeze(-1), dim=1, keepdim=True)
    return torch.cat([gather_compact(x, idx), extra], dim=1), compl


# --------------------------------------------------------------------------- block / model
def block_forward(x: Tensor, p: Dict[str, Tensor], i: int, cfg: VitConfig, keep: Optional[int],
                  precision: str = "fp32", forced_idx: Optional[Tensor] = None, drop: Optional[Tensor] = None):
    """Block_TopK.forward topk.py:83-99 / Block_EVIT.forward evit.py:105-129 /
    deit_viz.Block.forward :69-72 (keep=None).  Returns (x, idx|None, compl|None).

    forced_idx (tests only): use this selection instead of the block's own top-k ("teacher forcing"): isolates the
    continuous arithmetic from the discrete decisions when a bf16 pipeline is compared with this restatement."""
    pre = f"blocks.{i}."
    xn = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps, precision)
    a, cls_rows = attention(xn, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"],
                            p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"],
                            cfg.num_heads, precision)
    # drop [2,B] (training only): timm DropPath's per-image scale (0 or 1/keep_prob) of the attention / MLP branch, topk.py:87,95
    x = x + (a if drop is None else _r(a * drop[0][:, None, None], precision))
    idx = compl = None
    N = x.shape[1]
    # topk.py:55-58: keep_rate<1 and left_tokens != N-1, else the block is a plain block
    if keep is not None and keep != N - 1:
        assert 1 <= keep < N - 1
        scores = cls_scores_from_heads(cls_rows)
        idx = cls_topk_select(scores, keep) if forced_idx is None else forced_idx
        assert idx.shape == (x.shape[0], keep)
        if cfg.family == "evit":
            x, compl = evit_fuse(x, idx, scores)
        else:
            x = gather_compact(x, idx)
    xn2 = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps, precision)
    m = mlp(xn2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"], precision)
    x = x + (m if drop is None else _r(m * drop[1][:, None, None], precision))
    return x, idx, compl


@torch.no_grad()
def vit_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32",
                return_viz: bool = False, forced: Optional[Dict[int, Tensor]] = None, drop: Optional[Tensor] = None):
    """TopKVisionTransformer.forward topk.py:179-212 / EfficientVisionTransformer.forward
    evit.py:209-244 / deit_viz.VisionTransformer.forward :186-212 (eval mode).

    viz (when asked) mirrors the reference's viz_data index contract:
      Kept_Tokens[blk]   = idx [B,K] (evit: trailing -1 appended, evit.py:123)
      Fusion_Assign[blk] = compl [B,P-K] (evit only)
    plus Tokens[blk] = token count after the block (for shape checks).
    """
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    keeps = stage_keep_counts(cfg) if cfg.family in ("topk", "evit") else {}
    viz = {"Kept_Tokens": {}, "Fusion_Assign": {}, "Tokens": {}}
    for i in range(cfg.depth):
        h, idx, compl = block_forward(h, p, i, cfg, keeps.get(i), precision, None if forced is None else forced.get(i),
                                      None if drop is None else drop[2 * i: 2 * i + 2])           # drop [2*depth, B]: DropPath scales
        viz["Tokens"][i] = h.shape[1]
        if idx is not None:
            if cfg.family == "evit":
                idx = torch.cat([idx, torch.full((idx.shape[0], 1), -1, dtype=idx.dtype)], dim=1)
                viz["Fusion_Assign"][i] = compl.numpy()
            viz["Kept_Tokens"][i] = idx.numpy()
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits


# =========================================================================== ToMe (models/tome.py)
def tome_schedule(cfg: VitConfig) -> Dict[int, int]:
    """block index -> r (tokens to remove in that block), tome.py:145-156.

    One keep_rate -> targets int(P0 * kr**(i+1)); several -> ABSOLUTE patch-token counts used verbatim (cast to int: the
    CLI delivers floats, which crash the reference's slicing -- SURVEY App. A.4).  r_i = previous target - target_i; the
    per-call clamp min(r, (N-1)//2) (tome.py:253) is applied where the token count is known (tome_block_r)."""
    targets = list(cfg.keep_rate)
    loc = list(cfg.reduction_loc)
    if len(targets) == 1:
        targets = [int(cfg.num_patches * targets[0] ** (i + 1)) for i in range(len(loc))]
    assert len(targets) == len(loc), "keep_rate / reduction_loc length mismatch"
    out, prev = {}, cfg.num_patches
    for t, l in zip(targets, loc):
        out[int(l)] = prev - int(t)
        prev = int(t)
    return out


def tome_block_r(r: int, n_tokens: int) -> int:
    """tome.py:252-253: at most 50 % of the non-CLS tokens can be merged in one call."""
    return max(0, min(int(r), (n_tokens - 1) // 2))


def tome_attention(xn: Tensor, qkv_w: Tensor, qkv_b: Tensor, proj_w: Tensor, proj_b: Tensor, num_heads: int,
                   size: Optional[Tensor], precision: str = "fp32") -> Tuple[Tensor, Tensor]:
    """Attention_ToMe.forward tome.py:41-58: attention + log(size) added to the logits of every KEY (proportional attention,
    only once a merge has happened), returns (proj output, metric = k.mean(1) [B,N,dh])."""
    B, N, D = xn.shape
    dh = D // num_heads
    qkv = _r(xn @ _r(qkv_w, precision).t() + qkv_b, precision)
    qkv = qkv.reshape(B, N, 3, num_heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    if size is not None:
        s = s + size.log()[:, None, None, :, 0]
    if precision == "bf16":
        m = s.amax(dim=-1, keepdim=True)
        p = torch.exp(s - m)
        o = (round_bf16(p) @ v) / p.sum(dim=-1, keepdim=True)
    else:
        o = s.softmax(dim=-1) @ v
    o = _r(o.transpose(1, 2).reshape(B, N, D), precision)
    out = dropout_site(_r(o @ _r(proj_w, precision).t() + proj_b, precision), precision)
    return out, k.mean(1)


def tome_match(metric: Tensor, r: int) -> Tuple[Tensor, Tensor, Tensor]:
    """bipartite_soft_matching tome.py:230-277 with class_token=True: returns (unm_idx [B,na-r], src_idx [B,r], dst_idx [B,r]),
    indices into the EVEN-position set A (unm, src) / the ODD-position set B (dst).  unm is sorted ascending so CLS (A[0],
    whose scores are -inf) stays first.  Tie rules of this build: row maximum -> first index (as torch's CPU max); the
    descending argsort of the row maxima -> lowest index first (torch's order is unspecified; fixtures are tie-free)."""
    metric = metric / metric.norm(dim=-1, keepdim=True)
    a, b = metric[..., ::2, :], metric[..., 1::2, :]
    scores = a @ b.transpose(-1, -2)
    scores[..., 0, :] = -math.inf
    node_max, node_idx = scores.max(dim=-1)
    edge = torch.sort(node_max, dim=-1, descending=True, stable=True).indices
    unm = edge[..., r:].sort(dim=1)[0]
    src = edge[..., :r]
    dst = node_idx.gather(dim=-1, index=src)
    return unm, src, dst


def tome_merge(x: Tensor, size: Optional[Tensor], unm: Tensor, src: Tensor, dst: Tensor) -> Tuple[Tensor, Tensor]:
    """merge_wavg tome.py:309-323 over the `merge` closure tome.py:279-289: size-weighted sums scattered into the odd tokens,
    divided by the summed sizes.  Output order: unmerged even tokens (ascending), then ALL odd tokens."""
    if size is None:
        size = torch.ones_like(x[..., 0, None])

    def merge(t):
        s_, d_ = t[..., ::2, :], t[..., 1::2, :]
        n, t1, c = s_.shape
        u = s_.gather(dim=-2, index=unm[..., None].expand(n, unm.shape[1], c))
        sr = s_.gather(dim=-2, index=src[..., None].expand(n, src.shape[1], c))
        d_ = d_.scatter_add(-2, dst[..., None].expand(n, dst.shape[1], c), sr)
        return torch.cat([u, d_], dim=1)

    xs = merge(x * size)
    sz = merge(size)
    return xs / sz, sz


def tome_assignment(unm: Tensor, src: Tensor, dst: Tensor, n_tokens: int) -> Tensor:
    """Assignment_Maps entry of tome.py:91-99 (what merge_source + amax computes through a B*N*N eye): for every input PATCH
    token (CLS dropped), the index of its output token minus 1."""
    B = unm.shape[0]
    na = (n_tokens + 1) // 2
    n_unm = unm.shape[1]
    out = torch.empty(B, n_tokens, dtype=torch.int64)
    pos_unm = torch.zeros(B, na, dtype=torch.int64)
    pos_unm.scatter_(1, unm, torch.arange(n_unm).expand(B, n_unm))
    pos_unm.scatter_(1, src, n_unm + dst)
    out[:, 0::2] = pos_unm
    out[:, 1::2] = n_unm + torch.arange(n_tokens // 2)
    return (out - 1)[:, 1:]


def tome_block_forward(x: Tensor, size: Optional[Tensor], p: Dict[str, Tensor], i: int, cfg: VitConfig, r: int,
                       precision: str = "fp32", forced=None):
    """Block_ToMe.forward tome.py:83-104.  Returns (x, size, assignment|None).  forced = (unm, src, dst) replaces the block's
    own matching (tests only, see block_forward)."""
    pre = f"blocks.{i}."
    xn = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps, precision)
    a, metric = tome_attention(xn, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"], p[pre + "attn.proj.weight"],
                               p[pre + "attn.proj.bias"], cfg.num_heads, size, precision)
    x = x + a
    assign = None
    r = tome_block_r(r, x.shape[1])
    if r > 0:
        if forced is None:
            with torch.no_grad():                      # tome.py:258: the matching runs under no_grad; only merge_wavg is differentiable
                unm, src, dst = tome_match(metric.detach(), r)
        else:
            unm, src, dst = forced
        assert src.shape == (x.shape[0], r)
        assign = tome_assignment(unm, src, dst, x.shape[1])
        x, size = tome_merge(x, size, unm, src, dst)
    xn2 = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps, precision)
    x = x + mlp(xn2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"],
                precision)
    return x, size, assign


@torch.no_grad()
def tome_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32", return_viz: bool = False,
                 forced=None):
    """ToMeVisionTransformer.forward tome.py:183-223 (eval)."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    sched = tome_schedule(cfg)
    size = None
    viz = {"Assignment_Maps": {}, "Tokens": {}}
    for i in range(cfg.depth):
        h, size, assign = tome_block_forward(h, size, p, i, cfg, sched.get(i, 0), precision,
                                             None if forced is None else forced.get(i))
        viz["Tokens"][i] = h.shape[1]
        if assign is not None and i in sched:
            viz["Assignment_Maps"][i] = assign.numpy()
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits
