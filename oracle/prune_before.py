"""Oracle: the two families that reduce tokens BEFORE a block runs -- DyViT (eval) and SiT.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  torch-CPU fp32, functional; reference = /root/reference (read-only):

  models/dyvit.py  PredictorLG :90-119, eval branch of DynamicVisionTransformer.forward :230-243, batch_index_select :340-356
  models/sit.py    TokenSlimmingModule :25-40, SelfSlimmedVisionTransformer.forward :99-150

Extra state-dict keys:
  score_predictor.{j}.in_conv.0.{weight,bias}   LayerNorm(D)  (nn.LayerNorm default eps 1e-5)
  score_predictor.{j}.in_conv.1.{weight,bias}   Linear(D, D)
  score_predictor.{j}.out_conv.{0,2,4}.*        Linear(D, D/2), Linear(D/2, D/4), Linear(D/4, 2)
  cluster_layers.{j}.weight.0.{weight,bias}     LayerNorm(D)  (eps 1e-5)
  cluster_layers.{j}.weight.1.{weight,bias}     Linear(D, D/2)
  cluster_layers.{j}.weight.3.{weight,bias}     Linear(D/2, K_j)
  cluster_layers.{j}.scale                      [1,1,1]

precision="bf16" places the rounding points where the HIP pipeline has them (bf16 GEMM operands and bf16 activations
between GEMMs, fp32 accumulation, fp32 residual stream, fp32 scores / soft assignment).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .vit import dropout_site, VitConfig, _r, block_forward, embed_tokens, gelu_erf, head, layer_norm, patch_embed, round_bf16

Tensor = torch.Tensor
LN_EPS_DEFAULT = 1e-5          # nn.LayerNorm(embed_dim) without eps: dyvit.py:97, sit.py:30


# =========================================================================================== DyViT (eval)
def dyvit_keep_counts(cfg: VitConfig) -> Dict[int, int]:
    """dyvit.py:175-176 + :232: one keep_rate -> kr**(i+1) per stage, else verbatim RATIOS; K = int(num_patches * ratio) of the
    INITIAL patch count at every stage."""
    ratios = list(cfg.keep_rate)
    loc = list(cfg.reduction_loc)
    if len(ratios) == 1:
        ratios = [ratios[0] ** (i + 1) for i in range(len(loc))]
    assert len(ratios) == len(loc), "keep_rate / reduction_loc length mismatch"
    return {int(l): int(cfg.num_patches * r) for r, l in zip(ratios, loc)}


def dyvit_predictor_scores(x_sp: Tensor, p: Dict[str, Tensor], j: int, precision: str = "fp32", eps: float = 1e-6) -> Tensor:
    """PredictorLG.forward dyvit.py:113-119 with policy == 1 (the eval path never changes prev_decision from ones, :216,:237):
    returns log_softmax(...)[:, :, 0]  [B, P] -- the score dyvit.py:231 ranks."""
    pre = f"score_predictor.{j}."
    h = layer_norm(x_sp, p[pre + "in_conv.0.weight"], p[pre + "in_conv.0.bias"], LN_EPS_DEFAULT, precision)
    h = _r(gelu_erf(h @ _r(p[pre + "in_conv.1.weight"], precision).t() + p[pre + "in_conv.1.bias"]), precision)
    C = h.shape[-1]
    local_x = h[:, :, :C // 2]
    # (x * policy).sum(1) / policy.sum(1) + eps  -- eps OUTSIDE the fraction (dyvit.py:117)
    global_x = _r(h[:, :, C // 2:].sum(dim=1, keepdim=True) / float(h.shape[1]) + eps, precision)
    h = torch.cat([local_x, global_x.expand(-1, h.shape[1], -1)], dim=-1)
    h = _r(gelu_erf(h @ _r(p[pre + "out_conv.0.weight"], precision).t() + p[pre + "out_conv.0.bias"]), precision)
    h = _r(gelu_erf(h @ _r(p[pre + "out_conv.2.weight"], precision).t() + p[pre + "out_conv.2.bias"]), precision)
    logits = h @ p[pre + "out_conv.4.weight"].t() + p[pre + "out_conv.4.bias"]
    return torch.log_softmax(logits, dim=-1)[:, :, 0]


def dyvit_select(scores: Tensor, k: int) -> Tensor:
    """dyvit.py:233: argsort(score, descending)[:, :K].  torch's unstable sort leaves tie order unspecified; this build takes
    the lowest index first (fixtures are tie-free)."""
    return torch.sort(scores, dim=1, descending=True, stable=True).indices[:, :k]


@torch.no_grad()
def dyvit_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32", return_viz: bool = False,
                  forced: Optional[Dict[int, Tensor]] = None):
    """DynamicVisionTransformer.forward dyvit.py:203-263, eval mode."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    keeps = dyvit_keep_counts(cfg)
    viz = {"Kept_Tokens": {}, "Scores": {}, "Tokens": {}}
    j = 0
    for i in range(cfg.depth):
        if i in keeps:
            scores = dyvit_predictor_scores(h[:, 1:], p, j, precision)
            idx = dyvit_select(scores, keeps[i]) if forced is None else forced[i]
            assert idx.shape == (h.shape[0], keeps[i])
            # batch_index_select(x, cat(0, keep+1)) dyvit.py:234-236
            h = torch.cat([h[:, :1], torch.gather(h[:, 1:], 1, idx.unsqueeze(-1).expand(-1, -1, h.shape[-1]))], dim=1)
            viz["Kept_Tokens"][i] = idx.numpy()
            viz["Scores"][i] = scores
            j += 1
        h, _, _ = block_forward(h, p, i, cfg, None, precision)
        viz["Tokens"][i] = h.shape[1]
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits


def dyvit_softmax_with_policy(attn: Tensor, policy: Tensor, eps: float = 1e-6) -> Tensor:
    """Policy_Attention.softmax_with_policy dyvit.py:39-51 (training forward): attn [B,H,N,N] scaled logits, policy [B,N,1]."""
    B, N, _ = policy.size()
    attn_policy = policy.reshape(B, 1, 1, N)
    eye = torch.eye(N, dtype=attn_policy.dtype).view(1, 1, N, N)
    attn_policy = attn_policy + (1.0 - attn_policy) * eye
    max_att = torch.max(attn, dim=-1, keepdim=True)[0]
    attn = attn - max_att
    attn = attn.to(torch.float32).exp_() * attn_policy.to(torch.float32)
    attn = (attn + eps / N) / (attn.sum(dim=-1, keepdim=True) + eps)
    return attn.type_as(max_att)


def dyvit_predictor_logprob(x_sp: Tensor, policy: Tensor, p: Dict[str, Tensor], j: int, precision: str = "fp32", eps: float = 1e-6) -> Tensor:
    """PredictorLG.forward dyvit.py:113-119 with a real policy [B,P,1] (training): log_softmax over the two classes, [B,P,2]."""
    pre = f"score_predictor.{j}."
    h = layer_norm(x_sp, p[pre + "in_conv.0.weight"], p[pre + "in_conv.0.bias"], LN_EPS_DEFAULT, precision)
    h = _r(gelu_erf(_r(h @ _r(p[pre + "in_conv.1.weight"], precision).t() + p[pre + "in_conv.1.bias"], precision)), precision)
    C = h.shape[-1]
    local_x = h[:, :, :C // 2]
    global_x = _r((h[:, :, C // 2:] * policy).sum(dim=1, keepdim=True) / torch.sum(policy, dim=1, keepdim=True) + eps, precision)
    h = torch.cat([local_x, global_x.expand(-1, h.shape[1], -1)], dim=-1)
    h = _r(gelu_erf(_r(h @ _r(p[pre + "out_conv.0.weight"], precision).t() + p[pre + "out_conv.0.bias"], precision)), precision)
    h = _r(gelu_erf(_r(h @ _r(p[pre + "out_conv.2.weight"], precision).t() + p[pre + "out_conv.2.bias"], precision)), precision)
    return torch.log_softmax(h @ p[pre + "out_conv.4.weight"].t() + p[pre + "out_conv.4.bias"], dim=-1)


def dyvit_policy_block(x: Tensor, policy: Tensor, p: Dict[str, Tensor], i: int, cfg: VitConfig, precision: str = "fp32") -> Tensor:
    """Block_DyVIT.forward dyvit.py:85-88 over Policy_Attention.forward :53-67 (policy [B,N,1])."""
    pre = f"blocks.{i}."
    B, N, D = x.shape
    H = cfg.num_heads
    xn = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps, precision)
    qkv = _r(xn @ _r(p[pre + "attn.qkv.weight"], precision).t() + p[pre + "attn.qkv.bias"], precision)
    q, k, v = qkv.reshape(B, N, 3, H, D // H).permute(2, 0, 3, 1, 4).unbind(0)
    attn = dyvit_softmax_with_policy((q @ k.transpose(-2, -1)) * ((D // H) ** -0.5), policy)
    o = _r((_r(attn, precision) @ v).transpose(1, 2).reshape(B, N, D), precision)
    x = x + dropout_site(_r(o @ _r(p[pre + "attn.proj.weight"], precision).t() + p[pre + "attn.proj.bias"], precision), precision)
    xn2 = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps, precision)
    from .vit import mlp
    return x + mlp(xn2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"], precision)


def dyvit_train_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, gumbels: Dict[int, Tensor], precision: str = "fp32",
                        forced_hard: Optional[Dict[int, Tensor]] = None):
    """DynamicVisionTransformer.forward dyvit.py:203-261 in TRAINING mode: nothing is pruned; at each stage the predictor's
    log-probabilities go through F.gumbel_softmax(hard=True) (torch/nn/functional.py: y_soft = softmax(logits + gumbels);
    y_hard = one_hot(argmax); ret = y_hard - y_soft.detach() + y_soft) with the Gumbel noise `gumbels[stage]` [B,P,2] given (the
    reference draws it), the first class times prev_decision is the keep decision, [1, decision] the policy of this and every later
    block.  forced_hard[stage] [B,P] (tests): take this one-hot value instead of the argmax (the straight-through gradient stays).
    Returns (logits, features [B,P,D], prev_decision [B,P,1], out_pred_prob list of [B,P]).  Differentiable (not under no_grad)."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    B, P = h.shape[0], cfg.num_patches
    stages = sorted(dyvit_keep_counts(cfg))
    prev = torch.ones(B, P, 1)
    policy = torch.ones(B, P + 1, 1)
    out_pred = []
    for i in range(cfg.depth):
        if i in stages:
            j = stages.index(i)
            score = dyvit_predictor_logprob(h[:, 1:], prev, p, j, precision)
            y_soft = torch.softmax(score + gumbels[j], dim=-1)
            if forced_hard is None:
                hard0 = (torch.argmax(y_soft, dim=-1) == 0).to(y_soft.dtype)
            else:
                hard0 = forced_hard[j].to(y_soft.dtype)
            y0 = (hard0 - y_soft[..., 0].detach() + y_soft[..., 0]).unsqueeze(-1)          # straight-through
            keep = y0 * prev
            out_pred.append(keep.reshape(B, P))
            policy = torch.cat([torch.ones(B, 1, 1), keep], dim=1)
            prev = keep
        h = dyvit_policy_block(h, policy, p, i, cfg, precision)
    hn = layer_norm(h, p["norm.weight"], p["norm.bias"], cfg.ln_eps)
    logits = _r(hn[:, 0], precision) @ _r(p["head.weight"], precision).t() + p["head.bias"]
    return logits, hn[:, 1:], prev.detach(), out_pred


# =========================================================================================== SiT
def sit_cluster_counts(cfg: VitConfig) -> Dict[int, int]:
    """sit.py:77-83: one keep_rate -> int(P0 * kr**(i+1)); several -> ABSOLUTE counts used verbatim."""
    counts = list(cfg.keep_rate)
    loc = list(cfg.reduction_loc)
    if len(counts) == 1:
        counts = [int(cfg.num_patches * counts[0] ** (i + 1)) for i in range(len(loc))]
    assert len(counts) == len(loc), "keep_rate / reduction_loc length mismatch"
    return {int(l): int(c) for c, l in zip(counts, loc)}


def sit_slim(x_sp: Tensor, p: Dict[str, Tensor], j: int, precision: str = "fp32"):
    """TokenSlimmingModule.forward sit.py:36-40: soft assignment = softmax over the TOKEN axis of an MLP's K logits,
    out = W^T x.  Returns (x [B,K,D], weight [B,K,P])."""
    pre = f"cluster_layers.{j}."
    h = layer_norm(x_sp, p[pre + "weight.0.weight"], p[pre + "weight.0.bias"], LN_EPS_DEFAULT, precision)
    h = _r(gelu_erf(h @ _r(p[pre + "weight.1.weight"], precision).t() + p[pre + "weight.1.bias"]), precision)
    w = h @ _r(p[pre + "weight.3.weight"], precision).t() + p[pre + "weight.3.bias"]
    w = torch.softmax(w * p[pre + "scale"], dim=1).transpose(2, 1)
    return torch.bmm(w, x_sp), w


@torch.no_grad()
def sit_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32", return_viz: bool = False):
    """SelfSlimmedVisionTransformer.forward sit.py:99-150, eval mode."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    counts = sit_cluster_counts(cfg)
    viz = {"Assignment_Maps": {}, "Soft_Assignment_Maps": {}, "Tokens": {}}
    j = 0
    for i in range(cfg.depth):
        if i in counts:
            xs, w = sit_slim(h[:, 1:], p, j, precision)
            h = torch.cat([h[:, :1], xs], dim=1)
            viz["Soft_Assignment_Maps"][i] = w.detach().numpy()
            viz["Assignment_Maps"][i] = torch.argmax(w, dim=-2).numpy()        # sit.py:122
            j += 1
        h, _, _ = block_forward(h, p, i, cfg, None, precision)
        viz["Tokens"][i] = h.shape[1]
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits


# =========================================================================================== PatchMerger
def patchmerger_merge(x_sp: Tensor, p: Dict[str, Tensor], j: int, precision: str = "fp32", scale: float = 1.0):
    """PatchMerger.forward patchmerger.py:35-39: LayerNorm (eps 1e-5), similarity of K learned queries to every token, softmax
    over the TOKENS, output = attention-weighted sum of the NORMALISED tokens.  Returns (x [B,K,D], attn [B,K,P])."""
    pre = f"cluster_layers.{j}."
    xn = torch.nn.functional.layer_norm(x_sp, (x_sp.shape[-1],), p[pre + "norm.weight"], p[pre + "norm.bias"], LN_EPS_DEFAULT)
    sim = torch.matmul(_r(p[pre + "queries"], precision), _r(xn, precision).transpose(-1, -2)) * scale
    attn = sim.softmax(dim=-1)
    return torch.matmul(attn, xn), attn


@torch.no_grad()
def patchmerger_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32", return_viz: bool = False):
    """PatchMergerVisionTransformer.forward patchmerger.py:98-150, eval mode (cluster counts as SiT, patchmerger.py:78-79)."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    counts = sit_cluster_counts(cfg)
    viz = {"Assignment_Maps": {}, "Soft_Assignment_Maps": {}, "Tokens": {}}
    j = 0
    for i in range(cfg.depth):
        if i in counts:
            xs, w = patchmerger_merge(h[:, 1:], p, j, precision)
            h = torch.cat([h[:, :1], xs], dim=1)
            viz["Soft_Assignment_Maps"][i] = w.detach().numpy()
            viz["Assignment_Maps"][i] = torch.argmax(w, dim=-2).numpy()        # patchmerger.py:122
            j += 1
        h, _, _ = block_forward(h, p, i, cfg, None, precision)
        viz["Tokens"][i] = h.shape[1]
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits
