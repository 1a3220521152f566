"""CPU oracle for the ViT-with-token-reduction hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain torch-CPU fp32 (functional, no nn.Module, no timm),
the algorithm of the reference's `model.forward()` for the families on the hot path
(SURVEY.md section 8a).  Every function cites the reference file:line it follows.

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` -- as the checker / reported CPU baseline, never as the thing shipped.
The product package `tokenreduction_amd` must not import anything from here and has
no CPU fallback: it raises when the HIP library is missing.

Pinning: the reference has no tests/golden vectors of its own, so
the oracle is pinned against outputs of the reference itself, run in the build
container through a test-only timm stand-in (`tests/golden/gen_golden.py`, vectors
committed under `tests/golden/*.npz`, checked by `tests/test_oracle_golden.py`).
"""
from .vit import (  # noqa: F401
    VitConfig, vit_forward, block_forward, attention, layer_norm, mlp, gelu_erf,
    patch_embed, embed_tokens, head, cls_topk_select, cls_scores_from_heads,
    gather_compact, complement_idx, evit_fuse, stage_keep_counts, round_bf16,
    tome_schedule, tome_block_r, tome_attention, tome_match, tome_merge, tome_assignment, tome_block_forward, tome_forward,
)
from .prune_before import (  # noqa: F401
    dyvit_keep_counts, dyvit_predictor_scores, dyvit_select, dyvit_forward, dyvit_softmax_with_policy, dyvit_train_forward,
    dyvit_predictor_logprob, dyvit_policy_block, sit_cluster_counts, sit_slim, sit_forward, patchmerger_merge, patchmerger_forward,
)
from .cluster import (  # noqa: F401
    dpcknn_cluster_counts, dpcknn_distances, dpcknn_scores, dpcknn_assign, dpcknn_cluster, dpcknn_merge, dpcknn_ctm,
    dpcknn_forward, kmedoids_token_weights, kmedoids_fit, kmedoids_block_forward, kmedoids_forward, sinkhorn_log_iterations, sinkhorn_transport, sinkhorn_layer, sinkhorn_forward,
)
from .ats import (  # noqa: F401
    ats_sample_counts, ats_token_bounds, ats_sample_steps, ats_scores, ats_cdf, ats_ids_from_cdf, ats_sample_ids, ats_block_forward, ats_forward,
)
from .heuristic import heuristic_masks, heuristic_forward  # noqa: F401


def forward(params, x, cfg, precision="fp32", return_viz=False, forced=None, noise=None, extra=None):
    """Family dispatch used by the tests."""
    if cfg.family == "heuristic":
        assert forced is None
        return heuristic_forward(params, x, cfg, extra["heuristic_pattern"], extra["not_contiguous"], extra.get("min_radius"),
                                 precision, return_viz)
    if cfg.family == "patchmerger":
        assert forced is None
        return patchmerger_forward(params, x, cfg, precision, return_viz)
    if cfg.family == "kmedoids":
        return kmedoids_forward(params, x, cfg, precision, return_viz, forced=forced, equal_first=(extra or {}).get("equal_first"))
    if cfg.family == "sinkhorn":
        assert forced is None
        return sinkhorn_forward(params, x, cfg, precision, return_viz)
    if cfg.family == "ats":
        return ats_forward(params, x, cfg, precision, return_viz, static_pad=False, forced=forced)
    if cfg.family == "dpcknn":
        return dpcknn_forward(params, x, cfg, noise, precision, return_viz, forced)
    if cfg.family == "tome":
        return tome_forward(params, x, cfg, precision, return_viz, forced)
    if cfg.family == "dyvit":
        return dyvit_forward(params, x, cfg, precision, return_viz, forced)
    if cfg.family == "sit":
        assert forced is None
        return sit_forward(params, x, cfg, precision, return_viz)
    return vit_forward(params, x, cfg, precision, return_viz, forced)
