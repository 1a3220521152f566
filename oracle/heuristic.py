"""Oracle: fixed spatial pruning patterns (models/heuristic.py).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  models/heuristic.py   prep_pattern :157-181, prep_pattern_stage_subset :184-224, MaskedHeuristicAttention :25-57,
                        HeuristicVisionTransformer.forward :233-277

Nothing is learned and nothing depends on the image: every block in the reduction range masks the patch tokens whose distance
(L1 / L2 / Linf on the patch grid) from the image centre exceeds that block's radius.  Tokens are never removed -- they are
masked as attention keys (and queries) from that block on, so the token count stays N.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .ats import ats_block_forward
from .vit import VitConfig, embed_tokens, head, patch_embed

Tensor = torch.Tensor


def _distances(num_patches: int, pattern: str):
    """heuristic.py:158-169: distance of every patch from the grid centre."""
    P = int(num_patches ** 0.5)
    xs = torch.linspace(-P // 2, P // 2, steps=P)
    ys = torch.linspace(-P // 2, P // 2, steps=P)
    x, y = torch.meshgrid(xs, ys, indexing="ij")
    pattern = pattern.lower()
    if pattern == "l1":
        z = torch.abs(x) + torch.abs(y)
    elif pattern == "l2":
        z = torch.sqrt(x * x + y * y)
    elif pattern == "linf":
        z = torch.max(torch.abs(x), torch.abs(y))
    else:
        raise ValueError(pattern)
    return z, P


def heuristic_masks(cfg: VitConfig, pattern: str, not_contiguous: bool, min_radius: Optional[float] = None) -> Dict[int, Tensor]:
    """block index -> bool mask [P0] of the patch tokens that stay visible from that block on (blocks outside the reduction
    range are absent).  not_contiguous: radii chosen so the visible count is closest to int(P0 * kr**(i+1))
    (prep_pattern_stage_subset); else a linear radius schedule between the first and last reduction block (prep_pattern)."""
    z, P = _distances(cfg.num_patches, pattern)
    depth = cfg.depth
    if not_contiguous:
        loc = [int(l) for l in cfg.reduction_loc]
        assert len(cfg.keep_rate) == 1, "the reference only defines num_tokens for a single keep_rate (heuristic.py:127-128)"
        num_tokens = [int(cfg.num_patches * cfg.keep_rate[0] ** (i + 1)) for i in range(len(loc))]
        unique = torch.unique(z)
        within = [torch.sum(z <= u).item() for u in unique]
        closest = []
        for nt in num_tokens:
            best, thr = np.inf, None
            for idx, t in enumerate(within):
                if np.abs(nt - t) < best:
                    best, thr = np.abs(nt - t), unique[idx].item()
            closest.append(thr)
        closest = [unique[-1].item()] + closest
        out, counter = {}, 0
        for idx in range(depth):
            if idx in loc:
                counter += 1
                out[idx] = (z <= torch.ones((P, P)) * closest[counter]).reshape(P * P)
        return out
    start, end = int(min(cfg.reduction_loc)), int(max(cfg.reduction_loc))
    if min_radius is None or min_radius <= 0:
        min_radius = z[P // 2, P // 2]
    steps = end - start + 3
    threshold = torch.linspace(float(z[0, 0]), float(min_radius), steps)
    threshold = F.pad(threshold, (max(start - 1, 0), 0), value=float(z[0, 0]))
    threshold = F.pad(threshold, (0, max(depth - end - 1, 0)), value=float(threshold[-1]))
    return {idx: (z <= threshold[idx]).reshape(P * P) for idx in range(start, end + 1)}


@torch.no_grad()
def heuristic_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, pattern: str, not_contiguous: bool,
                      min_radius: Optional[float] = None, precision: str = "fp32", return_viz: bool = False):
    """HeuristicVisionTransformer.forward heuristic.py:233-277, eval mode."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    masks = heuristic_masks(cfg, pattern, not_contiguous, min_radius)
    B = h.shape[0]
    mask = None
    viz = {"Kept_Tokens_Abs": {}, "Tokens": {}}
    for i in range(cfg.depth):
        if i in masks:
            viz["Kept_Tokens_Abs"][i] = masks[i].nonzero(as_tuple=True)[0].unsqueeze(0).expand(B, -1).numpy()
            mask = F.pad(masks[i], (1, 0), value=True).unsqueeze(0).expand(B, -1)
        m = mask if mask is not None else torch.ones(B, h.shape[1], dtype=torch.bool)
        h, _, _, _ = ats_block_forward(h, m, p, i, cfg, 0, precision)
        viz["Tokens"][i] = h.shape[1]
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits
