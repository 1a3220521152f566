"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

Fixtures store only seeds + expected outputs; the weights are re-created from the seed
(numpy PCG64 `standard_normal` is stable across numpy versions), so the .npz stay small.
Key names are the reference's state-dict names.
"""
from __future__ import annotations

import types

import numpy as np
import torch


def _normal(rng, shape, std):
    return torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32))


def make_params(cfg, seed: int, qkv_gain: float = 1.0):
    """cfg: anything with embed_dim, depth, num_heads, mlp_ratio, num_classes, img_size,
    patch_size, in_chans.  All biases and LN affine terms are non-trivial on purpose so every
    epilogue path is exercised.  qkv_gain>1 gives 'peaky' attention (realistic score spread)."""
    rng = np.random.default_rng(seed)
    D = cfg.embed_dim
    Hd = int(D * cfg.mlp_ratio)
    P = (cfg.img_size // cfg.patch_size) ** 2
    p = {}
    p["cls_token"] = _normal(rng, (1, 1, D), 0.02)
    p["pos_embed"] = _normal(rng, (1, P + 1, D), 0.02)
    p["patch_embed.proj.weight"] = _normal(rng, (D, cfg.in_chans, cfg.patch_size, cfg.patch_size), 0.02)
    p["patch_embed.proj.bias"] = _normal(rng, (D,), 0.02)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        p[b + "norm1.weight"] = 1.0 + _normal(rng, (D,), 0.1)
        p[b + "norm1.bias"] = _normal(rng, (D,), 0.05)
        p[b + "attn.qkv.weight"] = _normal(rng, (3 * D, D), 0.02 * qkv_gain)
        p[b + "attn.qkv.bias"] = _normal(rng, (3 * D,), 0.02)
        p[b + "attn.proj.weight"] = _normal(rng, (D, D), 0.02)
        p[b + "attn.proj.bias"] = _normal(rng, (D,), 0.02)
        p[b + "norm2.weight"] = 1.0 + _normal(rng, (D,), 0.1)
        p[b + "norm2.bias"] = _normal(rng, (D,), 0.05)
        p[b + "mlp.fc1.weight"] = _normal(rng, (Hd, D), 0.02)
        p[b + "mlp.fc1.bias"] = _normal(rng, (Hd,), 0.02)
        p[b + "mlp.fc2.weight"] = _normal(rng, (D, Hd), 0.02)
        p[b + "mlp.fc2.bias"] = _normal(rng, (D,), 0.02)
    p["norm.weight"] = 1.0 + _normal(rng, (D,), 0.1)
    p["norm.bias"] = _normal(rng, (D,), 0.05)
    p["head.weight"] = _normal(rng, (cfg.num_classes, D), 0.02)
    p["head.bias"] = _normal(rng, (cfg.num_classes,), 0.02)
    return p


def make_stage_params(cfg, case: dict):
    """Extra per-stage weights of the families that own learned reduction modules (state-dict names of the reference):
    DyViT score_predictor.{j}.* (dyvit.py:96-110), SiT cluster_layers.{j}.* (sit.py:29-34).  Separate generator
    (wseed + 1000), so the trunk weights of a case do not depend on its family."""
    fam = case["family"]
    rng = np.random.default_rng(case["wseed"] + 1000)
    D = cfg.embed_dim
    p = {}
    n_stages = len(case["reduction_loc"])
    if fam == "dyvit":
        for j in range(n_stages):
            b = f"score_predictor.{j}."
            p[b + "in_conv.0.weight"] = 1.0 + _normal(rng, (D,), 0.1)
            p[b + "in_conv.0.bias"] = _normal(rng, (D,), 0.05)
            for name, (o, i) in (("in_conv.1", (D, D)), ("out_conv.0", (D // 2, D)), ("out_conv.2", (D // 4, D // 2)),
                                 ("out_conv.4", (2, D // 4))):
                p[b + name + ".weight"] = _normal(rng, (o, i), 0.08)
                p[b + name + ".bias"] = _normal(rng, (o,), 0.02)
    elif fam == "dpcknn" and not case.get("equal_weight", False):
        for j in range(n_stages):
            p[f"cluster_layers.{j}.score.weight"] = _normal(rng, (1, D), 0.05)
            p[f"cluster_layers.{j}.score.bias"] = _normal(rng, (1,), 0.02)
    elif fam == "patchmerger":
        from oracle.prune_before import sit_cluster_counts
        counts = sit_cluster_counts(cfg)
        for j, loc in enumerate(case["reduction_loc"]):
            b = f"cluster_layers.{j}."
            p[b + "norm.weight"] = 1.0 + _normal(rng, (D,), 0.1)
            p[b + "norm.bias"] = _normal(rng, (D,), 0.05)
            p[b + "queries"] = _normal(rng, (counts[loc], D), 0.15)
    elif fam == "sinkhorn":
        from oracle.cluster import dpcknn_cluster_counts
        counts = dpcknn_cluster_counts(cfg)
        for j, loc in enumerate(case["reduction_loc"]):
            p[f"cluster_layers.{j}.v"] = _normal(rng, (counts[loc], D), 1.0)          # sinkhorn.py:62 randn
    elif fam == "sit":
        from oracle.prune_before import sit_cluster_counts
        counts = sit_cluster_counts(cfg)
        for j, loc in enumerate(case["reduction_loc"]):
            b = f"cluster_layers.{j}."
            p[b + "weight.0.weight"] = 1.0 + _normal(rng, (D,), 0.1)
            p[b + "weight.0.bias"] = _normal(rng, (D,), 0.05)
            p[b + "weight.1.weight"] = _normal(rng, (D // 2, D), 0.08)
            p[b + "weight.1.bias"] = _normal(rng, (D // 2,), 0.02)
            p[b + "weight.3.weight"] = _normal(rng, (counts[loc], D // 2), 0.08)
            p[b + "weight.3.bias"] = _normal(rng, (counts[loc],), 0.02)
            p[b + "scale"] = torch.full((1, 1, 1), 1.5 + 0.25 * j, dtype=torch.float32)
    return p


def make_images(batch: int, img_size: int, seed: int, in_chans: int = 3):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal((batch, in_chans, img_size, img_size)).astype(np.float32))


# The golden cases (name -> kwargs).  Used by tests/golden/gen_golden.py to drive the reference
# and by tests to drive the oracle / HIP path on identical inputs.
GOLDEN_CASES = {
    # micro models built straight from the reference classes (dh = 64 like every DeiT size)
    "topk_micro": dict(family="topk", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                       keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=11, xseed=12, qkv_gain=6.0),
    "evit_micro": dict(family="evit", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                       keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=21, xseed=22, qkv_gain=6.0),
    "deit_micro": dict(family="deit", embed_dim=128, depth=3, num_heads=2, num_classes=16,
                       keep_rate=[1.0], reduction_loc=[], batch=2, wseed=31, xseed=32, qkv_gain=6.0),
    # explicit per-stage ratios incl. a repeated one: evit.py:79-80 early-out (K == N-1) at blk 2
    "evit_micro_explicit": dict(family="evit", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                                keep_rate=[0.6, 0.6, 0.3], reduction_loc=[0, 2, 3], batch=2,
                                wseed=41, xseed=42, qkv_gain=6.0),
    # BASELINE.json configs[0] / configs[1] at full DeiT-S size through the factory names
    "topk_small_kr09": dict(family="topk", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                            keep_rate=[0.9], reduction_loc=[3, 6, 9], batch=2, wseed=51, xseed=52,
                            qkv_gain=4.0, factory="topk_small_patch16_224"),
    "topk_small_kr07": dict(family="topk", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                            keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=61, xseed=62,
                            qkv_gain=4.0, factory="topk_small_patch16_224"),
    "evit_small_kr07": dict(family="evit", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                            keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=71, xseed=72,
                            qkv_gain=4.0, factory="evit_small_patch16_224"),
    # BASELINE.json north_star's own target line: DeiT-S at keep_rate 0.5 (stages of 99 / 50 / 25 tokens: topk.py:55-56, :141-150)
    "topk_small_kr05": dict(family="topk", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                            keep_rate=[0.5], reduction_loc=[3, 6, 9], batch=2, wseed=63, xseed=64,
                            qkv_gain=4.0, factory="topk_small_patch16_224"),
    "evit_small_kr05": dict(family="evit", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                            keep_rate=[0.5], reduction_loc=[3, 6, 9], batch=2, wseed=73, xseed=74,
                            qkv_gain=4.0, factory="evit_small_patch16_224"),
    # ToMe (models/tome.py): geometric keep_rate, and BASELINE configs[2] "r=16 every block" = explicit absolute counts
    "tome_micro": dict(family="tome", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                       keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=91, xseed=92, qkv_gain=6.0),
    "tome_small_kr07": dict(family="tome", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                            keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=93, xseed=94,
                            qkv_gain=4.0, factory="tome_small_patch16_224"),
    "tome_small_r16": dict(family="tome", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                           keep_rate=[196 - 16 * (i + 1) for i in range(12)], reduction_loc=list(range(12)), batch=2,
                           wseed=95, xseed=96, qkv_gain=4.0, factory="tome_small_patch16_224"),
    # DyViT eval path (models/dyvit.py): predictor MLP scores -> argsort -> gather BEFORE the block
    "dyvit_micro": dict(family="dyvit", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                        keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=101, xseed=102, qkv_gain=6.0),
    "dyvit_small_kr07": dict(family="dyvit", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                             keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=103, xseed=104,
                             qkv_gain=4.0, factory="dyvit_small_patch16_224"),
    # DropPath in training (timm drop_path, topk.py:78,87,95; train.py's default --drop-path 0.1): gradient fixture only
    "topk_micro_droppath": dict(family="topk", embed_dim=128, depth=4, num_heads=2, num_classes=16, drop_path=0.3, train_only=True,
                                keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=6, wseed=13, xseed=14, qkv_gain=6.0),
    # Dropout in training (timm's drop_rate, train.py:46 --drop: pos_drop topk.py:186, proj_drop :53, the Mlp's two nn.Dropout), the
    # first one together with DropPath.  Gradient fixtures only; the reference's keep masks are recorded with them (bit-packed) and
    # replayed by the oracle and by the HIP executor.  (Not ATS: its reference masks have the data-dependent batch-max row count,
    # ats.py:77-78, the executor's the static bound -- ATS with dropout is covered by the every-family smoke test only.)
    "topk_micro_dropout": dict(family="topk", embed_dim=128, depth=4, num_heads=2, num_classes=16, drop_rate=0.1, drop_path=0.2, train_only=True,
                               keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=4, wseed=15, xseed=16, qkv_gain=6.0),
    "evit_micro_dropout": dict(family="evit", embed_dim=128, depth=4, num_heads=2, num_classes=16, drop_rate=0.15, train_only=True,
                               keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=17, xseed=18, qkv_gain=6.0),
    # DyViT TRAINING (dyvit.py:221-229, 257-261) with the distillation outputs: gradient fixtures only (grad_<name>.npz)
    "dyvit_micro_train": dict(family="dyvit", embed_dim=128, depth=4, num_heads=2, num_classes=16, dyvit_distill=True, train_only=True,
                              keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=105, xseed=106, qkv_gain=6.0),
    # ... and at 384 x 384 (577 tokens under the keep policy: the online-softmax policy attention and its key-blocked backward)
    "dyvit_micro_train_384": dict(family="dyvit", embed_dim=128, depth=4, num_heads=2, num_classes=16, dyvit_distill=True, train_only=True,
                                  img_size=384, keep_rate=[0.6], reduction_loc=[1, 2, 3], batch=2, wseed=125, xseed=126, qkv_gain=6.0),
    "dyvit_small_train": dict(family="dyvit", embed_dim=384, depth=12, num_heads=6, num_classes=1000, dyvit_distill=True, train_only=True,
                              keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=107, xseed=108,
                              qkv_gain=4.0, factory="dyvit_small_patch16_224"),
    # DeiT-T width (D = 192, 3 heads: the *_tiny_* factory names): D/2 = 96 is not a multiple of the GEMMs' 64-deep K-step, so the
    # predictor / slimming-module hidden layers run zero-padded to 128 -- in eval since round 2, through the tape and the backward
    # since round 3 (models_act.py:280, :1375)
    "dyvit_tiny_train": dict(family="dyvit", embed_dim=192, depth=4, num_heads=3, num_classes=16, dyvit_distill=True, train_only=True,
                             keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=115, xseed=116, qkv_gain=5.0),
    "sit_tiny": dict(family="sit", embed_dim=192, depth=4, num_heads=3, num_classes=16,
                     keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=117, xseed=118, qkv_gain=5.0),
    # DPC-KNN (models/dpcknn.py): density-peak clustering + weighted merge BEFORE the block (noise recorded in the fixture)
    "dpcknn_micro": dict(family="dpcknn", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                         keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=121, xseed=122, qkv_gain=6.0),
    "dpcknn_small_kr07": dict(family="dpcknn", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                              keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=123, xseed=124,
                              qkv_gain=4.0, factory="dpcknn_small_patch16_224"),
    "dpcknn_micro_equal": dict(family="dpcknn", embed_dim=128, depth=4, num_heads=2, num_classes=16, equal_weight=True,
                               keep_rate=[0.5], reduction_loc=[0, 2], batch=2, wseed=125, xseed=126, qkv_gain=6.0),
    # ATS (models/ats.py): inverse-CDF sampling on CLS attention x |v| inside the attention
    "ats_micro": dict(family="ats", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                      keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=131, xseed=132, qkv_gain=6.0),
    "ats_small_kr07": dict(family="ats", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                           keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=133, xseed=134,
                           qkv_gain=4.0, factory="ats_small_patch16_224"),
    "ats_small_kr05": dict(family="ats", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                           keep_rate=[0.5], reduction_loc=[3, 6, 9], batch=3, wseed=135, xseed=136,
                           qkv_gain=4.0, factory="ats_small_patch16_224"),
    # K-Medoids (models/kmedoids.py): weighted K-Medoids seeded by the previous block's attention column sums
    "kmedoids_micro": dict(family="kmedoids", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                           keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=151, xseed=152, qkv_gain=6.0),
    "kmedoids_small_kr07": dict(family="kmedoids", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                                keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=153, xseed=154,
                                qkv_gain=4.0, factory="kmedoids_small_patch16_224"),
    # args.equal_weight (kmedoids.py:43-58): first medoid from numpy's global RNG (seeded with xseed by the generator), farthest-point init
    "kmedoids_micro_equal": dict(family="kmedoids", embed_dim=128, depth=4, num_heads=2, num_classes=16, equal_weight=True,
                                 keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=155, xseed=156, qkv_gain=6.0),
    # Sinkhorn (models/sinkhorn.py): optimal-transport soft assignment to learned centres BEFORE the block
    "sinkhorn_micro": dict(family="sinkhorn", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                           keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=141, xseed=142, qkv_gain=6.0),
    "sinkhorn_small_kr07": dict(family="sinkhorn", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                                keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=143, xseed=144,
                                qkv_gain=4.0, factory="sinkhorn_small_patch16_224"),
    # PatchMerger (models/patchmerger.py): learned queries attend over the normalised tokens BEFORE the block
    "patchmerger_micro": dict(family="patchmerger", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                              keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=171, xseed=172, qkv_gain=6.0),
    "patchmerger_small_kr07": dict(family="patchmerger", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                                   keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=173, xseed=174,
                                   qkv_gain=4.0, factory="patchmerger_small_patch16_224"),
    # Heuristic (models/heuristic.py): fixed spatial masks, no token removal
    "heuristic_micro_l2": dict(family="heuristic", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                               keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=2, wseed=181, xseed=182, qkv_gain=6.0,
                               heuristic_pattern="l2", not_contiguous=True),
    "heuristic_small_linf": dict(family="heuristic", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                                 keep_rate=[0.7], reduction_loc=[3, 9], batch=2, wseed=183, xseed=184, qkv_gain=4.0,
                                 factory="heuristic_small_patch16_224", heuristic_pattern="linf", not_contiguous=False,
                                 min_radius=2.0),
    # SiT (models/sit.py): soft assignment (softmax over tokens) BEFORE the block
    "sit_micro": dict(family="sit", embed_dim=128, depth=4, num_heads=2, num_classes=16,
                      keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=3, wseed=111, xseed=112, qkv_gain=6.0),
    "sit_small_kr07": dict(family="sit", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                           keep_rate=[0.7], reduction_loc=[3, 6, 9], batch=2, wseed=113, xseed=114,
                           qkv_gain=4.0, factory="sit_small_patch16_224"),
    # 384x384 inputs (BASELINE configs[4] geometry, N = 577): chunked attention, Sinkhorn with K*P beyond the LDS, K-Medoids / ATS
    # side outputs of the long kernel.  Top-K keeps int(ratio*196) tokens even here (196 is hard-coded, topk.py:56).
    "topk_micro_384": dict(family="topk", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                           keep_rate=[0.5], reduction_loc=[1, 2, 3], batch=2, wseed=161, xseed=162, qkv_gain=6.0),
    "sinkhorn_micro_384": dict(family="sinkhorn", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                               keep_rate=[0.25], reduction_loc=[1, 2, 3], batch=2, wseed=163, xseed=164, qkv_gain=6.0),
    "kmedoids_micro_384": dict(family="kmedoids", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                               keep_rate=[0.25], reduction_loc=[1, 2, 3], batch=2, wseed=165, xseed=166, qkv_gain=6.0),
    "ats_micro_384": dict(family="ats", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                          keep_rate=[0.25], reduction_loc=[1, 2, 3], batch=2, wseed=167, xseed=168, qkv_gain=6.0),
    # 384x384 with keep_rate 0.9: 518 / 466 / 419 clusters (Sinkhorn's whole-image kernel beyond 256 centres, DPC-KNN's merge backward
    # beyond 256 clusters) and soft assignments wider than the 192 columns the fused merge holds in registers (SiT, PatchMerger)
    "sinkhorn_micro_384_kr09": dict(family="sinkhorn", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                                    keep_rate=[0.9], reduction_loc=[1, 2, 3], batch=2, wseed=171, xseed=172, qkv_gain=6.0),
    "dpcknn_micro_384_kr09": dict(family="dpcknn", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                                  keep_rate=[0.9], reduction_loc=[1, 2, 3], batch=2, wseed=173, xseed=174, qkv_gain=6.0),
    "sit_micro_384_kr07": dict(family="sit", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                               keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=2, wseed=175, xseed=176, qkv_gain=6.0),
    "patchmerger_micro_384_kr07": dict(family="patchmerger", embed_dim=128, depth=4, num_heads=2, num_classes=16, img_size=384,
                                       keep_rate=[0.7], reduction_loc=[1, 2, 3], batch=2, wseed=177, xseed=178, qkv_gain=6.0),
    # BASELINE.json configs[3] families at DeiT-B width (D = 768, H = 12): ATS + DPC-KNN keep_rate 0.5.  qkv gain 2 gives the
    # attention logits the same spread (std ~1.2-2.5) as gain 4 does at DeiT-S width: q.k/8 scales with D
    "dpcknn_base_kr05": dict(family="dpcknn", embed_dim=768, depth=12, num_heads=12, num_classes=1000,
                             keep_rate=[0.5], reduction_loc=[3, 6, 9], batch=2, wseed=201, xseed=202,
                             qkv_gain=2.0, factory="dpcknn_base_patch16_224"),
    "ats_base_kr05": dict(family="ats", embed_dim=768, depth=12, num_heads=12, num_classes=1000,
                          keep_rate=[0.5], reduction_loc=[3, 6, 9], batch=2, wseed=203, xseed=204,
                          qkv_gain=2.0, factory="ats_base_patch16_224"),
    # BASELINE.json configs[4] at full size: DeiT-B at 384 x 384 (577 tokens), Sinkhorn / K-Medoids keep_rate 0.25
    "sinkhorn_base_384_kr025": dict(family="sinkhorn", embed_dim=768, depth=12, num_heads=12, num_classes=1000, img_size=384,
                                    keep_rate=[0.25], reduction_loc=[3, 6, 9], batch=2, wseed=207, xseed=208,
                                    qkv_gain=2.0, factory="sinkhorn_base_patch16_224"),
    "kmedoids_base_384_kr025": dict(family="kmedoids", embed_dim=768, depth=12, num_heads=12, num_classes=1000, img_size=384,
                                    keep_rate=[0.25], reduction_loc=[3, 6, 9], batch=2, wseed=209, xseed=210,
                                    qkv_gain=2.0, factory="kmedoids_base_patch16_224"),
    # dense DeiT-B: the trunk alone at D = 768 / H = 12 / depth 12 (no discrete decision anywhere)
    "deit_base": dict(family="deit", embed_dim=768, depth=12, num_heads=12, num_classes=1000,
                      keep_rate=[1.0], reduction_loc=[], batch=2, wseed=205, xseed=206,
                      qkv_gain=2.0, factory="deit_base_patch16_224_local"),
    "deit_small": dict(family="deit", embed_dim=384, depth=12, num_heads=6, num_classes=1000,
                       keep_rate=[1.0], reduction_loc=[], batch=2, wseed=81, xseed=82,
                       qkv_gain=4.0, factory="deit_small_patch16_224_local"),
}


# Gradient fixtures (tests/golden/grad_<case>.npz): the reference's own `loss.backward()` (engine.py:60-76 with a plain
# cross-entropy criterion) in train mode on the golden case's weights / images, labels from grad_labels().  Too large to store
# whole (22 M values at DeiT-S), so per parameter: the L2 norm and <= 512 evenly strided entries (grad_sample_index).
GRAD_CASES = ["deit_micro", "topk_micro", "evit_micro", "tome_micro", "dpcknn_micro", "dpcknn_micro_equal", "ats_micro", "topk_small_kr07",
              "evit_small_kr07", "topk_small_kr05", "evit_small_kr05", "tome_small_r16", "deit_base", "dpcknn_base_kr05", "ats_base_kr05", "dyvit_micro_train", "dyvit_small_train", "kmedoids_micro", "heuristic_micro_l2", "topk_micro_droppath",
              "sit_micro", "patchmerger_micro", "sinkhorn_micro", "sit_small_kr07", "patchmerger_small_kr07", "sinkhorn_small_kr07",
              "topk_micro_384", "kmedoids_micro_384", "ats_micro_384", "sinkhorn_micro_384",
              "sinkhorn_micro_384_kr09", "dpcknn_micro_384_kr09", "sit_micro_384_kr07", "patchmerger_micro_384_kr07",
              "dyvit_tiny_train", "sit_tiny", "topk_micro_dropout", "evit_micro_dropout", "dyvit_micro_train_384"]


def finetune_ingest_setup():
    """(224 config, 384 config, synthetic DeiT-layout 224 x 224 checkpoint state dict) of the f3 ingest fixture: a micro trunk trained at
    224 with 16 classes, to be fine-tuned at 384 with 10 classes (the head must be dropped, the position embedding resized 14 -> 24)."""
    cfg224 = types.SimpleNamespace(embed_dim=64, depth=3, num_heads=1, mlp_ratio=4, num_classes=16, img_size=224, patch_size=16, in_chans=3)
    cfg384 = types.SimpleNamespace(embed_dim=64, depth=3, num_heads=1, mlp_ratio=4, num_classes=10, img_size=384, patch_size=16, in_chans=3)
    return cfg224, cfg384, make_params(cfg224, 777, 1.0)


def dyvit_token_ratio(case: dict):
    r = list(case["keep_rate"])
    return [r[0] ** (i + 1) for i in range(len(case["reduction_loc"]))] if len(r) == 1 else r


def dyvit_train_loss(outputs, labels, case: dict):
    """The loss of the DyViT gradient fixtures: cross-entropy + 2 x the keep-ratio loss of losses.py:113-118 + 0.5 x a token MSE
    against seeded pseudo-teacher tokens on the kept positions (the shape of losses.py:134-151 with mse_token) -- so that all
    four training outputs (logits, features, prev_decision mask, out_pred_prob) carry gradient."""
    pred, token_pred, mask, out_pred = outputs
    loss = torch.nn.functional.cross_entropy(pred, labels)
    ratio = dyvit_token_ratio(case)
    pl = 0.0
    for i, score in enumerate(out_pred):
        pl = pl + ((score.mean(1) - ratio[i]) ** 2).mean()
    loss = loss + 2.0 * pl / len(out_pred)
    B, N, C = token_pred.shape
    rng = np.random.default_rng(case["xseed"] + 11)
    target = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(token_pred.device)
    keep = (mask.reshape(B * N) > 0.5)
    if keep.any():
        loss = loss + 0.5 * torch.pow(token_pred.reshape(B * N, C)[keep] - target.reshape(B * N, C)[keep], 2).mean()
    return loss


def grad_labels(case: dict):
    rng = np.random.default_rng(case["xseed"] + 7)
    return torch.from_numpy(rng.integers(0, case["num_classes"], size=(case["batch"],)).astype(np.int64))


def grad_sample_index(numel: int):
    return np.unique(np.linspace(0, numel - 1, min(numel, 512)).astype(np.int64))


def drop_path_draws(case: dict, rand_calls):
    """[2*depth, B] uniform draws for tokenreduction_amd's DropPath from the reference's recorded torch.rand calls: block 0 has
    drop probability 0 (an nn.Identity: no draw, topk.py:78,157), every later block draws twice (attention branch, MLP branch)."""
    B, depth = case["batch"], case["depth"]
    assert len(rand_calls) == 2 * (depth - 1), len(rand_calls)
    u = torch.zeros(2 * depth, B)
    for n, r in enumerate(rand_calls):
        u[2 + n] = torch.as_tensor(r).reshape(B)
    return u


def drop_path_scale(case: dict, draws):
    dpr = torch.linspace(0, case["drop_path"], case["depth"])
    keep = (1.0 - dpr).repeat_interleave(2).unsqueeze(1)
    return (keep + draws).floor() / keep


def dropout_masks(g):
    """The reference's nn.Dropout keep masks of a gradient fixture, in call order: list of uint8 arrays (flat), or None."""
    if "dropkeep" not in g.files:
        return None
    sizes = [int(v) for v in g["dropkeep_sizes"]]
    bits = np.unpackbits(g["dropkeep"])[: sum(sizes)]
    out, o = [], 0
    for n in sizes:
        out.append(bits[o: o + n].copy())
        o += n
    return out


def oracle_param_grads(case: dict, forced=None, precision: str = "fp32", noise=None, dropout=None):
    """dropout: the recorded keep masks (dropout_masks) of a case with drop_rate."""
    if dropout is not None:
        import oracle.vit as ovit
        with ovit.dropout_replay([torch.from_numpy(m) for m in dropout], case["drop_rate"]):
            return _oracle_param_grads(case, forced, precision, noise)
    return _oracle_param_grads(case, forced, precision, noise)


def _oracle_param_grads(case: dict, forced=None, precision: str = "fp32", noise=None):
    """Parameter gradients of cross-entropy(oracle logits, grad_labels) by torch.autograd over the oracle's functional forward
    (the reference's backward IS torch.autograd over its eager forward, engine.py:60-76).  Returns (loss, logits, {name: grad}).
    noise: DPC-KNN's density draws {blk: [B,P_in]} (the reference's torch.rand calls, recorded with the fixture)."""
    import oracle
    cfg, params = case_params(case)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    x = make_images(case["batch"], case.get("img_size", 224), case["xseed"])
    fam = case["family"]
    with torch.enable_grad():          # the undecorated functions: the public ones run under torch.no_grad()
        if fam == "dyvit":             # training forward: noise = {stage: gumbel [B,P,2]}, forced = {stage: hard decision [B,P]}
            outs = oracle.dyvit_train_forward(leaves, x, cfg, noise, precision, forced)
            loss = dyvit_train_loss(outs, grad_labels(case), case)
            loss.backward()
            return loss.item(), outs[0].detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
        if fam == "kmedoids":
            logits = oracle.kmedoids_forward.__wrapped__(leaves, x, cfg, precision, False, 3, forced)
        elif fam == "heuristic":
            logits = oracle.heuristic_forward.__wrapped__(leaves, x, cfg, case["heuristic_pattern"], case["not_contiguous"], case.get("min_radius"),
                                                          precision, False)
        elif fam == "tome":
            logits = oracle.tome_forward.__wrapped__(leaves, x, cfg, precision, False, forced)
        elif fam == "dpcknn":
            logits = oracle.dpcknn_forward.__wrapped__(leaves, x, cfg, noise, precision, False, forced)
        elif fam == "ats":
            logits = oracle.ats_forward.__wrapped__(leaves, x, cfg, precision, False, False, forced)
        elif fam == "sit":
            logits = oracle.sit_forward.__wrapped__(leaves, x, cfg, precision, False)
        elif fam == "patchmerger":
            logits = oracle.patchmerger_forward.__wrapped__(leaves, x, cfg, precision, False)
        elif fam == "sinkhorn":
            logits = oracle.sinkhorn_forward.__wrapped__(leaves, x, cfg, precision, False)
        else:
            drop = drop_path_scale(case, noise) if case.get("drop_path") else None       # noise = the [2*depth, B] uniform draws
            logits = oracle.vit_forward.__wrapped__(leaves, x, cfg, precision, False, forced, drop)
        loss = torch.nn.functional.cross_entropy(logits, grad_labels(case))
        loss.backward()
    return loss.item(), logits.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}


def case_params(case: dict):
    """(cfg, params) of a golden case: trunk + the family's stage modules."""
    cfg = case_config(case)
    params = make_params(cfg, case["wseed"], case.get("qkv_gain", 1.0))
    params.update(make_stage_params(cfg, case))
    return cfg, params


def case_config(case: dict):
    from oracle import VitConfig
    return VitConfig(family=case["family"], img_size=case.get("img_size", 224), embed_dim=case["embed_dim"], depth=case["depth"],
                     num_heads=case["num_heads"], num_classes=case["num_classes"],
                     keep_rate=list(case["keep_rate"]), reduction_loc=list(case["reduction_loc"]))


def assert_valid_ranking(idx, ref_scores, tol):
    """idx [B,K] is a correct descending top-K ordering of ref_scores [B,P] up to fp noise `tol`: consecutive picks never
    increase by more than tol, and nothing left out beats the last pick by more than tol.  (Exact equality with the
    reference's argsort is only defined where its own fp32 rounding does not decide: adjacent gaps of ~1e-6 are natural
    among 196 scores, SURVEY App. D.)"""
    idx = np.asarray(idx)
    ref_scores = np.asarray(ref_scores)
    B, K = idx.shape
    for b in range(B):
        assert len(set(idx[b].tolist())) == K, "duplicate token index"
        picked = ref_scores[b][idx[b]]
        assert (picked[:-1] - picked[1:] >= -tol).all(), "order violates the reference scores beyond fp noise"
        rest = np.delete(ref_scores[b], idx[b])
        if rest.size:
            assert rest.max() <= picked.min() + tol, "a dropped token outranks a kept one beyond fp noise"


def assert_valid_sampling(kept, cdf, steps, tol):
    """ATS (ats.py:73-77): `kept` [B,W] (0-based patch ids, -1 = pad) is a correct set of inverse-CDF samples of `cdf` [B,P]
    up to `tol` in the distance |step - cdf|: every grid point has a kept id within tol of its nearest cdf entry, and every
    kept id is within tol of nearest for some grid point.  (The reference's cdist takes its matmul form, whose fp32 rounding
    decides among candidates closer than ~3e-4 to a grid point.)"""
    kept, cdf, steps = np.asarray(kept), np.asarray(cdf, dtype=np.float64), np.asarray(steps, dtype=np.float64)
    for b in range(kept.shape[0]):
        ids = kept[b][kept[b] >= 0]
        assert (np.diff(ids) > 0).all(), "ids must be sorted and unique"
        d = np.abs(steps[:, None] - cdf[b][None, :])                    # [steps, P]
        ok = d <= d.min(axis=1, keepdims=True) + tol
        assert ok[:, ids].any(axis=1).all(), "a grid point has no kept id among its nearest cdf entries"
        assert ok[:, ids].any(axis=0).all(), "a kept id is nearest (within tol) to no grid point"
