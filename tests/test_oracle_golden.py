"""Pin the oracle (oracle/vit.py) against vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import oracle
from tests._params import GOLDEN_CASES, assert_valid_ranking, assert_valid_sampling, case_config, case_params, make_images, make_params

FP_TOL = 2e-5   # fp32 CPU: different op order (im2col GEMM vs conv, fused softmax) only


def fp_tol(case):
    """DeiT-B width (D = 768, depth 12, qkv gain 4): the same op-order noise is amplified to ~2e-4 on logits of magnitude ~2."""
    return 5e-4 if case["embed_dim"] >= 768 else FP_TOL


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", [n for n, c in GOLDEN_CASES.items() if not c.get("train_only")])
def test_model_matches_reference(golden_dir, name):
    case = GOLDEN_CASES[name]
    g = _load(golden_dir, name)
    cfg = case_config(case)
    params = make_params(cfg, case["wseed"], case.get("qkv_gain", 1.0))
    x = make_images(case["batch"], case.get("img_size", 224), case["xseed"])
    if case["family"] == "tome":
        logits, viz = oracle.tome_forward(params, x, cfg, return_viz=True)
        akeys = sorted(k for k in g.files if k.startswith("assign_"))
        assert len(akeys) == len(viz["Assignment_Maps"]) > 0
        for k in akeys:                                   # Assignment_Maps (tome.py:91-99): bit-exact
            np.testing.assert_array_equal(viz["Assignment_Maps"][int(k.split("_")[1])], g[k])
        np.testing.assert_allclose(logits.numpy(), g["logits"], atol=FP_TOL, rtol=0)
        np.testing.assert_allclose(viz["Final_Tokens"][:, :8].numpy(), g["final_tokens"], atol=1e-4, rtol=0)
        for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
            assert viz["Tokens"][int(blk)] == int(n)
        return
    if case["family"] in ("dyvit", "sit", "sinkhorn", "patchmerger"):
        return _check_prune_before(case, g, x)
    if case["family"] == "dpcknn":
        return _check_dpcknn(case, g, x)
    if case["family"] == "ats":
        return _check_ats(case, g, x)
    if case["family"] == "kmedoids":
        return _check_kmedoids(case, g, x)
    if case["family"] == "heuristic":
        return _check_heuristic(case, g, x)
    logits, viz = oracle.vit_forward(params, x, cfg, return_viz=True)
    # integer outputs: bit-exact
    kept_keys = sorted(k for k in g.files if k.startswith("kept_"))
    assert len(kept_keys) == len(viz["Kept_Tokens"])
    for k in kept_keys:
        blk = int(k.split("_")[1])
        np.testing.assert_array_equal(viz["Kept_Tokens"][blk], g[k])
    for k in (k for k in g.files if k.startswith("compl_")):
        blk = int(k.split("_")[1])
        np.testing.assert_array_equal(viz["Fusion_Assign"][blk], g[k])
    # floating point outputs
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=FP_TOL, rtol=0)
    if "final_tokens" in g.files:
        np.testing.assert_allclose(viz["Final_Tokens"][:, :8].numpy(), g["final_tokens"], atol=1e-4, rtol=0)
    if "token_counts" in g.files:
        for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
            assert viz["Tokens"][int(blk)] == int(n)


def _check_ats(case, g, x):
    cfg, params = case_params(case)
    counts = oracle.ats_sample_counts(cfg)
    kept_keys = sorted((k for k in g.files if k.startswith("kept_")), key=lambda k: int(k.split("_")[1]))
    # (1) the sampling op on the reference's own cdf: ids incl. pads and batch-max width, bit-exact
    for k in kept_keys:
        blk = int(k.split("_")[1])
        ids, _ = oracle.ats_ids_from_cdf(torch.from_numpy(g[f"cdf_{blk}"]), oracle.ats_sample_steps(counts[blk]))
        np.testing.assert_array_equal((ids[:, 1:] - 1).numpy(), g[k])
    # (2) teacher-forced with the reference's ids: every stage's cdf and the logits match the reference to fp32 noise
    forced = {int(k.split("_")[1]): torch.from_numpy(np.concatenate([np.zeros((g[k].shape[0], 1), np.int64), g[k] + 1], axis=1))
              for k in kept_keys}
    logits, viz = oracle.ats_forward(params, x, cfg, return_viz=True, forced=forced)
    for k in kept_keys:
        blk = int(k.split("_")[1])
        np.testing.assert_allclose(viz["Cdf"][blk].numpy(), g[f"cdf_{blk}"], atol=2e-4 if case["embed_dim"] >= 768 else 2e-6, rtol=0)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=fp_tol(case), rtol=0)
    np.testing.assert_allclose(viz["Final_Tokens"][:, :8].numpy(), g["final_tokens"], atol=1e-2 if case["embed_dim"] >= 768 else 1e-4, rtol=0)
    for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
        assert viz["Tokens"][int(blk)] == int(n)
    # (3) free-running: the first stage samples the same token set, so its ids must be valid samples of the REFERENCE's cdf
    # up to the rounding plateau of the matmul-form cdist; later stages are printed (their token sets may differ by a neighbour)
    l_free, v_free = oracle.ats_forward(params, x, cfg, return_viz=True)
    blk0 = int(kept_keys[0].split("_")[1])
    assert_valid_sampling(v_free["Kept_Tokens"][blk0], g[f"cdf_{blk0}"], oracle.ats_sample_steps(counts[blk0]).numpy(), tol=5e-4)
    agree = [float(np.mean([len(set(a[a >= 0].tolist()) & set(b[b >= 0].tolist())) / max(1, (b >= 0).sum())
                            for a, b in zip(v_free["Kept_Tokens"][int(k.split("_")[1])], g[k])])) for k in kept_keys]
    assert agree[0] > 0.95, agree       # later stages index into a token list that may already differ by a neighbour
    # (4) static padding to the bound K (the HIP layout) changes no valid row: same ids where valid, same logits
    l2, v2 = oracle.ats_forward(params, x, cfg, return_viz=True, static_pad=True)
    np.testing.assert_allclose(l2.numpy(), l_free.numpy(), atol=2e-6, rtol=0)
    for blk, kt in v_free["Kept_Tokens"].items():
        w = kt.shape[1]
        assert v2["Kept_Tokens"][blk].shape[1] == counts[blk] - 1
        np.testing.assert_array_equal(v2["Kept_Tokens"][blk][:, :w], kt)
        assert (v2["Kept_Tokens"][blk][:, w:] == -1).all()


def _check_heuristic(case, g, x):
    cfg, params = case_params(case)
    logits, viz = oracle.forward(params, x, cfg, return_viz=True, extra=case)
    keys = [k for k in g.files if k.startswith("keptabs_")]
    assert len(keys) == len(viz["Kept_Tokens_Abs"]) > 0
    for k in keys:                                         # the visible-token sets of every block in the range, bit-exact
        np.testing.assert_array_equal(viz["Kept_Tokens_Abs"][int(k.split("_")[1])], g[k])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(viz["Final_Tokens"][:, :1].numpy(), g["final_tokens"][:, :1], atol=1e-4, rtol=0)


def _check_kmedoids(case, g, x):
    cfg, params = case_params(case)
    equal_first = None
    if case.get("equal_weight"):          # the reference's np.random.choice draws, one per stage in forward order
        equal_first = dict(zip(sorted(oracle.dpcknn_cluster_counts(cfg)), g["first_medoid"].tolist()))
    logits, viz = oracle.kmedoids_forward(params, x, cfg, return_viz=True, equal_first=equal_first)
    kept_keys = [k for k in g.files if k.startswith("kept_")]
    assert len(kept_keys) == len(viz["Kept_Tokens"]) > 0
    for k in kept_keys:                                   # medoid ids and assignments of every stage, bit-exact
        blk = int(k.split("_")[1])
        if equal_first is None:
            np.testing.assert_allclose(viz["Weights"][blk].numpy(), g[f"weights_{blk}"], atol=1e-5, rtol=1e-5)
        np.testing.assert_array_equal(viz["Kept_Tokens"][blk], g[k])
        np.testing.assert_array_equal(viz["Assignment_Maps"][blk], g[f"assign_{blk}"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(viz["Final_Tokens"][:, :8].numpy(), g["final_tokens"], atol=1e-4, rtol=0)
    for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
        assert viz["Tokens"][int(blk)] == int(n)


def golden_noise(g):
    return {int(k.split("_")[1]): torch.from_numpy(g[k]) for k in g.files if k.startswith("noise_")}


def _check_dpcknn(case, g, x):
    cfg, params = case_params(case)
    logits, viz = oracle.forward(params, x, cfg, return_viz=True, noise=golden_noise(g))
    kept_keys = [k for k in g.files if k.startswith("kept_")]
    assert len(kept_keys) == len(viz["Kept_Tokens"]) > 0
    for k in kept_keys:                                   # same cdist, same noise -> same centres and assignment, bit-exact
        blk = int(k.split("_")[1])
        wide = case["embed_dim"] >= 768
        np.testing.assert_allclose(viz["Scores"][blk].numpy(), g[f"scores_{blk}"], atol=1e-5 if wide else 1e-6, rtol=1e-3 if wide else 1e-5)
        np.testing.assert_array_equal(viz["Kept_Tokens"][blk], g[k])
        np.testing.assert_array_equal(viz["Assignment_Maps"][blk], g[f"assign_{blk}"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=fp_tol(case), rtol=0)
    np.testing.assert_allclose(viz["Final_Tokens"][:, :8].numpy(), g["final_tokens"], atol=1e-2 if case["embed_dim"] >= 768 else 1e-4, rtol=0)
    for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
        assert viz["Tokens"][int(blk)] == int(n)


def _check_prune_before(case, g, x):
    cfg, params = case_params(case)
    logits, viz = oracle.forward(params, x, cfg, return_viz=True)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(viz["Final_Tokens"][:, :8].numpy(), g["final_tokens"], atol=1e-4, rtol=0)
    for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
        assert viz["Tokens"][int(blk)] == int(n)
    if case["family"] == "dyvit":
        kept_keys = [k for k in g.files if k.startswith("kept_")]
        assert len(kept_keys) == len(viz["Kept_Tokens"]) > 0
        for k in kept_keys:
            blk = int(k.split("_")[1])
            ref_scores = g[f"scores_{blk}"]
            np.testing.assert_allclose(viz["Scores"][blk].numpy(), ref_scores, atol=1e-5, rtol=0)
            # same scores in -> same indices out, bit exact (op-boundary pin)
            np.testing.assert_array_equal(oracle.dyvit_select(torch.from_numpy(ref_scores), g[k].shape[1]).numpy(), g[k])
            # end to end: the oracle's own ordering is the reference's up to fp noise in the scores; identical kept SETS
            assert_valid_ranking(viz["Kept_Tokens"][blk], ref_scores, tol=2e-5)
            np.testing.assert_array_equal(np.sort(viz["Kept_Tokens"][blk], axis=1), np.sort(g[k], axis=1))
    else:
        akeys = [k for k in g.files if k.startswith("assign_")]
        assert len(akeys) == len(viz["Assignment_Maps"]) > 0
        for k in akeys:
            blk = int(k.split("_")[1])
            np.testing.assert_allclose(viz["Soft_Assignment_Maps"][blk][:, :8], g[f"soft_{blk}"], atol=1e-6, rtol=1e-4)
            got, want = viz["Assignment_Maps"][blk], g[k]
            assert got.shape == want.shape
            if float(g[f"soft_margin_{blk}"]) > 1e-6:      # hard assignment = argmax over clusters: exact when no near-tie
                np.testing.assert_array_equal(got, want)
            # else: some token's two best clusters are closer than fp32 noise -- the argmax is not defined by the arithmetic


def test_attention_and_select_op(golden_dir):
    g = _load(golden_dir, "ops")
    x = torch.from_numpy(g["att_x"])
    out, cls_rows = oracle.attention(x, torch.from_numpy(g["att_qkv_weight"]), torch.from_numpy(g["att_qkv_bias"]),
                                     torch.from_numpy(g["att_proj_weight"]), torch.from_numpy(g["att_proj_bias"]), 2)
    np.testing.assert_allclose(out.numpy(), g["att_out"], atol=FP_TOL, rtol=0)
    scores = oracle.cls_scores_from_heads(cls_rows)
    np.testing.assert_allclose(scores.numpy(), g["att_scores"], atol=1e-7, rtol=1e-5)
    # same scores in -> same indices out, bit exact (op-boundary pin, SURVEY section 7 "hard parts")
    idx = oracle.cls_topk_select(torch.from_numpy(g["att_scores"]), g["att_idx"].shape[1])
    np.testing.assert_array_equal(idx.numpy(), g["att_idx"])
    np.testing.assert_array_equal(oracle.cls_topk_select(scores, g["att_idx"].shape[1]).numpy(), g["att_idx"])


def test_complement_idx(golden_dir):
    g = _load(golden_dir, "ops")
    for j in range(5):
        out = oracle.complement_idx(torch.from_numpy(g[f"compl{j}_idx"]), int(g[f"compl{j}_P"]))
        np.testing.assert_array_equal(out.numpy(), g[f"compl{j}_out"])


def test_evit_block(golden_dir):
    g = _load(golden_dir, "ops")
    from types import SimpleNamespace
    cfgp = SimpleNamespace(embed_dim=128, depth=1, num_heads=2, mlp_ratio=4, num_classes=4, img_size=224,
                           patch_size=16, in_chans=3)
    p = make_params(cfgp, 4321, qkv_gain=6.0)
    cfg = oracle.VitConfig(family="evit", embed_dim=128, depth=1, num_heads=2, keep_rate=[0.5], reduction_loc=[0])
    xo, idx, compl = oracle.block_forward(torch.from_numpy(g["evitblk_x"]), p, 0, cfg, keep=98)
    idx = torch.cat([idx, torch.full((idx.shape[0], 1), -1)], dim=1)
    np.testing.assert_array_equal(idx.numpy(), g["evitblk_idx"])
    np.testing.assert_array_equal(compl.numpy(), g["evitblk_compl"])
    np.testing.assert_allclose(xo.numpy(), g["evitblk_out"], atol=FP_TOL, rtol=0)


def test_tome_schedule_and_clamp():
    cfg = oracle.VitConfig(family="tome", keep_rate=[0.7], reduction_loc=[3, 6, 9])
    assert oracle.tome_schedule(cfg) == {3: 196 - 137, 6: 137 - 96, 9: 96 - 67}
    cfg = oracle.VitConfig(family="tome", keep_rate=[196 - 16 * (i + 1) for i in range(12)], reduction_loc=list(range(12)))
    n, seen = 197, []
    for i in range(12):                                   # BASELINE configs[2]: r=16 per block; SURVEY App. B token table
        n -= oracle.tome_block_r(oracle.tome_schedule(cfg)[i], n)
        seen.append(n)
    assert seen == [181, 165, 149, 133, 117, 101, 85, 69, 53, 37, 21, 11]
    cfg = oracle.VitConfig(family="tome", keep_rate=[0.25], reduction_loc=[3, 6, 9])     # 50 % cap: 99/62/53, not 50/13/4
    n, seen = 197, []
    for i in (3, 6, 9):
        n -= oracle.tome_block_r(oracle.tome_schedule(cfg)[i], n)
        seen.append(n)
    assert seen == [99, 62, 53]


def test_tie_rule_lowest_index_first():
    s = torch.tensor([[0.1, 0.5, 0.5, 0.2, 0.5]])
    assert oracle.cls_topk_select(s, 3).tolist() == [[1, 2, 4]]
    assert oracle.cls_topk_select(s, 4).tolist() == [[1, 2, 4, 3]]


def test_stage_keep_counts():
    cfg = oracle.VitConfig(family="topk", keep_rate=[0.7], reduction_loc=[3, 6, 9])
    assert oracle.stage_keep_counts(cfg) == {3: 137, 6: 96, 9: 67}
    cfg = oracle.VitConfig(family="topk", keep_rate=[0.9], reduction_loc=[3, 6, 9])
    assert oracle.stage_keep_counts(cfg) == {3: 176, 6: 158, 9: 142}
    cfg = oracle.VitConfig(family="topk", keep_rate=[0.5], reduction_loc=[3, 6, 9])
    assert oracle.stage_keep_counts(cfg) == {3: 98, 6: 49, 9: 24}


def test_bf16_mode_close_to_fp32():
    case = GOLDEN_CASES["deit_micro"]
    cfg = case_config(case)
    params = make_params(cfg, case["wseed"], case["qkv_gain"])
    x = make_images(2, 224, case["xseed"])
    a = oracle.vit_forward(params, x, cfg)
    b = oracle.vit_forward(params, x, cfg, precision="bf16")
    assert (a - b).abs().max().item() < 2e-2
