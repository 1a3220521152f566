"""Oracle: clustering-based token merging before a block -- DPC-KNN (models/dpcknn.py).  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  torch-CPU fp32, functional; reference = /root/reference (read-only):

  models/dpcknn.py   cluster_dpc_knn :44-100, merge_tokens :103-140, CTM.forward :153-172,
                     DPCKNNVisionTransformer.forward :229-290

Extra state-dict keys: cluster_layers.{j}.score.{weight [1,D], bias [1]} (absent with args.equal_weight).

The reference perturbs the densities with `torch.rand(...) * 1e-6` (dpcknn.py:71-72), i.e. it is not a function of its
inputs.  The restatement takes that noise as an INPUT (`noise[blk]`, uniform [0,1) of shape [B,P_in]); fixtures record the
draws the reference made (tests/golden/gen_golden.py spies torch.rand).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import torch.nn.functional as F

from .vit import dropout_site, VitConfig, _r, block_forward, embed_tokens, head, patch_embed

Tensor = torch.Tensor


def dpcknn_cluster_counts(cfg: VitConfig) -> Dict[int, int]:
    """dpcknn.py:214-215: one keep_rate -> int(P0 * kr**(i+1)); several -> ABSOLUTE counts verbatim."""
    counts = list(cfg.keep_rate)
    loc = list(cfg.reduction_loc)
    if len(counts) == 1:
        counts = [int(cfg.num_patches * counts[0] ** (i + 1)) for i in range(len(loc))]
    assert len(counts) == len(loc), "keep_rate / reduction_loc length mismatch"
    return {int(l): int(c) for c, l in zip(counts, loc)}


def dpcknn_distances(x: Tensor) -> Tensor:
    """dpcknn.py:59: torch.cdist(x, x) / sqrt(C).  cdist takes its matmul path for P > 25 (|a|^2 + |b|^2 - 2ab, clamped at
    1e-30, sqrt), so the diagonal is rounding residue, not 0 -- kept as is: the k nearest neighbours include the token itself."""
    return torch.cdist(x, x) / (x.shape[-1] ** 0.5)


def dpcknn_scores(dist: Tensor, noise: Tensor, k: int = 5):
    """dpcknn.py:68-88 from a distance matrix: (density [B,P], parent distance [B,P], score [B,P])."""
    dist_nearest, _ = torch.topk(dist, k=k, dim=-1, largest=False)
    density = (-(dist_nearest ** 2).mean(dim=-1)).exp()
    density = density + noise * 1e-6
    mask = (density[:, None, :] > density[:, :, None]).type(dist.dtype)
    dist_max = dist.flatten(1).max(dim=-1)[0][:, None, None]
    parent, _ = (dist * mask + dist_max * (1 - mask)).min(dim=-1)
    return density, parent, parent * density


def dpcknn_assign(dist: Tensor, centers: Tensor) -> Tensor:
    """dpcknn.py:90-98: every token goes to its nearest centre (first on ties); centres go to themselves."""
    B, K = centers.shape
    d = torch.gather(dist, 1, centers[:, :, None].expand(B, K, dist.shape[-1]))      # index_points(dist_matrix, index_down)
    idx_cluster = d.argmin(dim=1)
    idx_cluster.scatter_(1, centers, torch.arange(K).expand(B, K))
    return idx_cluster


def dpcknn_cluster(x: Tensor, cluster_num: int, noise: Tensor, k: int = 5, forced_centers: Optional[Tensor] = None):
    """cluster_dpc_knn dpcknn.py:44-100 (token_mask=None).  Returns (idx_cluster [B,P], index_down [B,K], score [B,P])."""
    dist = dpcknn_distances(x)
    _, _, score = dpcknn_scores(dist, noise, k)
    if isinstance(forced_centers, (tuple, list)):          # tests only: centres AND assignment map taken from the device
        centers, idx_cluster = forced_centers
        return idx_cluster, centers, score
    centers = torch.sort(score, dim=-1, descending=True, stable=True).indices[:, :cluster_num] if forced_centers is None \
        else forced_centers
    return dpcknn_assign(dist, centers), centers, score


def dpcknn_merge(x: Tensor, idx_cluster: Tensor, cluster_num: int, token_weight: Optional[Tensor]) -> Tensor:
    """merge_tokens dpcknn.py:103-132 (the idx_token / agg_weight bookkeeping never reaches an output): weighted mean of
    each cluster's tokens, weights normalised by (cluster sum + 1e-6)."""
    B, N, C = x.shape
    if token_weight is None:
        token_weight = x.new_ones(B, N, 1)
    idx = (idx_cluster + torch.arange(B)[:, None] * cluster_num).reshape(B * N)
    all_weight = token_weight.new_zeros(B * cluster_num, 1)
    all_weight.index_add_(0, idx, token_weight.reshape(B * N, 1))
    all_weight = all_weight + 1e-6
    norm_weight = token_weight / all_weight[idx].reshape(B, N, 1)
    merged = x.new_zeros(B * cluster_num, C)
    merged.index_add_(0, idx, (x * norm_weight).reshape(B * N, C))
    return merged.reshape(B, cluster_num, C)


def dpcknn_ctm(x_sp: Tensor, p: Dict[str, Tensor], j: int, cluster_num: int, noise: Tensor, k: int = 5,
               forced_centers: Optional[Tensor] = None):
    """CTM.forward dpcknn.py:153-172.  Returns (x [B,K,D], idx_centers, idx_cluster, score)."""
    key = f"cluster_layers.{j}.score.weight"
    token_weight = None
    if key in p:                                                                         # not equal_weight
        token_weight = (x_sp @ p[key].t() + p[f"cluster_layers.{j}.score.bias"]).exp()
    with torch.no_grad():                                  # dpcknn.py:56: the clustering runs under no_grad
        idx_cluster, centers, score = dpcknn_cluster(x_sp.detach(), cluster_num, noise, k, forced_centers)
    return dpcknn_merge(x_sp, idx_cluster, cluster_num, token_weight), centers, idx_cluster, score


@torch.no_grad()
def dpcknn_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, noise: Dict[int, Tensor], precision: str = "fp32",
                   return_viz: bool = False, forced: Optional[Dict[int, Tensor]] = None, k: int = 5):
    """DPCKNNVisionTransformer.forward dpcknn.py:229-290, eval mode.  The clustering itself is fp32 in both precisions (the
    HIP path clusters on the fp32 residual stream); `precision` only moves the trunk's rounding points."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    counts = dpcknn_cluster_counts(cfg)
    viz = {"Kept_Tokens": {}, "Assignment_Maps": {}, "Scores": {}, "Tokens": {}}
    j = 0
    for i in range(cfg.depth):
        if i in counts:
            xs, centers, idx_cluster, score = dpcknn_ctm(h[:, 1:], p, j, counts[i], noise[i], k,
                                                         None if forced is None else forced[i])
            h = torch.cat([h[:, :1], xs], dim=1)
            viz["Kept_Tokens"][i] = centers.numpy()
            viz["Assignment_Maps"][i] = idx_cluster.numpy()
            viz["Scores"][i] = score
            j += 1
        h, _, _ = block_forward(h, p, i, cfg, None, precision)
        viz["Tokens"][i] = h.shape[1]
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits


# =========================================================================================== Sinkhorn (models/sinkhorn.py)
def sinkhorn_log_iterations(Z: Tensor, log_mu: Tensor, log_nu: Tensor, iters: int) -> Tensor:
    """log_sinkhorn_iterations sinkhorn.py:25-38."""
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def sinkhorn_transport(scores: Tensor, eps: float, iters: int) -> Tensor:
    """log_optimal_transport sinkhorn.py:41-56: scores [B,K,P] -> transport plan [B,K,P] scaled by (K+P)."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    norm = -((m * one) + (n * one)).log()
    log_mu = norm.expand(m)[None].expand(b, -1)
    log_nu = norm.expand(n)[None].expand(b, -1)
    Z = sinkhorn_log_iterations(scores / eps, log_mu, log_nu, iters)
    return (Z - norm).exp()


def sinkhorn_layer(x_sp: Tensor, centers: Tensor, eps: float, iters: int, precision: str = "fp32"):
    """Sinkhorn.forward sinkhorn.py:66-86: unit-norm tokens against unit-norm centres, Sinkhorn-normalised soft assignment,
    output = assignment-weighted sum of the NORMALISED tokens.  Returns (x [B,K,D], soft [B,K,P])."""
    xh = F.normalize(x_sp, p=2, dim=-1)
    # sinkhorn.py:72-76 re-normalises self.v IN PLACE under no_grad and then uses the parameter itself: the value is the unit
    # vector, the gradient reaches v without the normalisation's Jacobian
    w = centers + (F.normalize(centers.detach(), p=2, dim=-1) - centers.detach())
    scores = torch.bmm(_r(xh, precision), _r(w, precision)[None].expand(x_sp.shape[0], -1, -1).transpose(1, 2))
    weights = sinkhorn_transport(scores.transpose(1, 2), eps, iters).transpose(1, 2)
    out = torch.bmm(xh.transpose(1, 2), weights).transpose(1, 2)
    return out, weights.transpose(1, 2)


@torch.no_grad()
def sinkhorn_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32", return_viz: bool = False,
                     eps: float = 1.0, iters: int = 3):
    """SinkhornVisionTransformer.forward sinkhorn.py:147-200, eval mode (cluster counts as sit/dpcknn: sinkhorn.py:128-129)."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    counts = dpcknn_cluster_counts(cfg)
    viz = {"Assignment_Maps": {}, "Soft_Assignment_Maps": {}, "Tokens": {}}
    j = 0
    for i in range(cfg.depth):
        if i in counts:
            xs, soft = sinkhorn_layer(h[:, 1:], p[f"cluster_layers.{j}.v"], eps, iters, precision)
            h = torch.cat([h[:, :1], xs], dim=1)
            viz["Soft_Assignment_Maps"][i] = soft.detach().numpy()
            viz["Assignment_Maps"][i] = torch.argmax(soft, dim=-2).numpy()            # sinkhorn.py:173
            j += 1
        h, _, _ = block_forward(h, p, i, cfg, None, precision)
        viz["Tokens"][i] = h.shape[1]
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits


# =========================================================================================== K-Medoids (models/kmedoids.py)
def kmedoids_token_weights(attn: Tensor) -> Tensor:
    """kmedoids.py:240: sum over heads, then over queries, of the previous block's softmax matrix; patch columns. -> [B,P,1]"""
    return torch.sum(torch.sum(attn, dim=1), dim=1)[:, 1:].unsqueeze(2)


def kmedoids_init_equal(x: Tensor, cluster_num: int, first: int) -> Tensor:
    """k_medoids_fit kmedoids.py:43-56, token_weight None (args.equal_weight): `first` is the reference's
    np.random.choice(np.arange(N), 1) draw, shared by the whole batch; then cluster_num-1 times the token whose LARGEST distance
    to the medoids chosen so far is largest joins (rows of chosen tokens are zeroed first; torch.max -> first index)."""
    B, N, C = x.shape
    cluster_idx = torch.full((B, 1), int(first), dtype=torch.long)
    for k in range(1, cluster_num):
        centers = torch.gather(x, 1, cluster_idx[:, :, None].expand(B, k, C))
        inter = torch.cdist(x, centers)
        inter.scatter_(1, cluster_idx[:, :, None].expand(B, k, k), 0.0)           # inter[b, cluster_idx[b, j], :] = 0
        new = inter.max(dim=-1).values.max(dim=-1).indices
        cluster_idx = torch.cat([cluster_idx, new.reshape(B, 1)], dim=-1)
    return cluster_idx


def kmedoids_fit(x: Tensor, cluster_num: int, iterations: int, token_weight: Tensor, forced_init: Optional[Tensor] = None):
    """k_medoids_fit kmedoids.py:40-85, weighted branch (token_weight given).  Returns (centres [B,K,D], cluster_idx [B,K],
    assignment [B,P]).  The per-cluster loop of :74-79 is restated without the B*P*P clone: a row outside cluster k sums to
    P*1e6, a row inside to sum_j dist_ij*w_i, argmin takes the first minimum (index 0 for an empty cluster)."""
    B, N, C = x.shape
    if forced_init is None:
        cluster_idx = torch.sort(token_weight.squeeze(2), dim=1, descending=True, stable=True).indices[:, :cluster_num]
    elif isinstance(forced_init, int):                    # equal_weight branch: the host's first-medoid draw, unit weights
        cluster_idx = kmedoids_init_equal(x, cluster_num, forced_init)
        token_weight = x.new_ones(B, N, 1)
    else:
        cluster_idx = forced_init
    cluster_idx = cluster_idx.clone()
    dist = torch.cdist(x, x)
    row_cost = torch.sum(dist * token_weight, dim=-1)                                   # [B,P]
    masked = torch.sum(torch.full((N,), 1000000.0))
    for _ in range(iterations):
        center_matrix = torch.gather(dist, 2, cluster_idx[:, None, :].expand(B, N, cluster_num))
        assignment = torch.argmin(center_matrix, dim=-1)
        for k in range(cluster_num):
            total = torch.where(assignment == k, row_cost, masked)
            cluster_idx[:, k] = torch.argmin(total, dim=1)
    center_matrix = torch.gather(dist, 2, cluster_idx[:, None, :].expand(B, N, cluster_num))
    assignment = torch.argmin(center_matrix, dim=-1)
    centers = torch.gather(x, 1, cluster_idx[:, :, None].expand(B, cluster_num, C))
    return centers, cluster_idx, assignment


def kmedoids_block_forward(x: Tensor, p: Dict[str, Tensor], i: int, cfg: VitConfig, precision: str = "fp32"):
    """kmedoids.Block.forward :125-132: a plain block that also returns the softmax matrix [B,H,N,N]."""
    from .vit import layer_norm, mlp, round_bf16
    pre = f"blocks.{i}."
    B, N, D = x.shape
    H = cfg.num_heads
    dh = D // H
    xn = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps, precision)
    qkv = _r(xn @ _r(p[pre + "attn.qkv.weight"], precision).t() + p[pre + "attn.qkv.bias"], precision)
    q, k, v = qkv.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    if precision == "bf16":
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        attn = e / e.sum(dim=-1, keepdim=True)
        o = (round_bf16(e) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        attn = s.softmax(dim=-1)
        o = attn @ v
    o = _r(o.transpose(1, 2).reshape(B, N, D), precision)
    x = x + dropout_site(_r(o @ _r(p[pre + "attn.proj.weight"], precision).t() + p[pre + "attn.proj.bias"], precision), precision)
    xn2 = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps, precision)
    x = x + mlp(xn2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"],
                precision)
    return x, attn


@torch.no_grad()
def kmedoids_forward(params: Dict[str, Tensor], x: Tensor, cfg: VitConfig, precision: str = "fp32", return_viz: bool = False,
                     iters: int = 3, forced: Optional[Dict[int, Tensor]] = None, equal_first: Optional[Dict[int, int]] = None):
    """KMedoidsVisionTransformer.forward kmedoids.py:219-272, eval mode, args.equal_weight False.  forced[blk] (tests only)
    replaces the whole clustering result of that block by the given medoid ids [B,K]."""
    p = params
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.patch_size, precision)
    h = embed_tokens(tok, p["cls_token"], p["pos_embed"])
    counts = dpcknn_cluster_counts(cfg)
    viz = {"Kept_Tokens": {}, "Assignment_Maps": {}, "Weights": {}, "Tokens": {}}
    attn = None
    for i in range(cfg.depth):
        if i in counts:
            w = kmedoids_token_weights(attn)
            # equal_first[blk] (args.equal_weight): the reference's np.random.choice draw for that stage
            xs, centers, assign = kmedoids_fit(h[:, 1:], counts[i], iters, w, None if equal_first is None else int(equal_first[i]))
            if forced is not None:
                centers = forced[i]
                xs = torch.gather(h[:, 1:], 1, centers[:, :, None].expand(-1, -1, h.shape[-1]))
            h = torch.cat([h[:, :1], xs], dim=1)
            viz["Kept_Tokens"][i] = centers.numpy()
            viz["Assignment_Maps"][i] = assign.numpy()
            viz["Weights"][i] = w.squeeze(2)
        h, attn = kmedoids_block_forward(h, p, i, cfg, precision)
        viz["Tokens"][i] = h.shape[1]
    logits = head(h, p["norm.weight"], p["norm.bias"], p["head.weight"], p["head.bias"], cfg.ln_eps, precision)
    if return_viz:
        viz["Final_Tokens"] = h
        return logits, viz
    return logits
