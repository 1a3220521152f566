"""GPU parity tests, model level: the drop-in modules (C executor) vs the oracle and the reference's golden vectors.

What is asserted, and why these tolerances:
  * executor == stepwise op sequence, bit for bit (same kernels, same order);
  * op-boundary pin at every reduction block: same scores in -> same indices out, BIT EXACT vs the oracle
    (SURVEY.md section 7 "hard parts": indices are exact only given identical scores);
  * vs the oracle run with the HIP pipeline's rounding points (precision="bf16") and vs the reference's fp32
    golden logits: the synthetic golden models use qkv_gain 4-6 ("peaky" attention) and are ill-conditioned --
    the fp32 reference and its own bf16-rounded restatement already differ by ~3e-2 on |logit| ~ 1.3, and the
    K-th/K+1-th score gap (~1e-3 relative) is below bf16 resolution (4e-3), so tokens near the K boundary flip.
    Measured max-abs on MI355X (round 1): dense DeiT-S 1.8e-2 / 2.8e-2, Top-K-S kr .7 3.0e-2 / 6.9e-2, EViT-S 1.7e-1 (one
    flipped token changes the fused token).  Asserted: relative L2 error of the logit vector <= 5 % (micro) / 12 % (DeiT-S
    size); kept-set overlap >= 0.70 (first stage >= 0.95).
    End-to-end index EXACTNESS is asserted at the op boundary in (2); the fp32 executor mode planned in DESIGN.md
    is what will pin end-to-end indices bit-exactly against the fp32 golden vectors.
The achieved numbers are printed so the log shows the margin.
"""
import os
import types

import numpy as np
import pytest
import torch

import oracle
from tests._params import GOLDEN_CASES, case_config, case_params, make_images, make_params

pytestmark = pytest.mark.gpu

FAM = {"deit": "VisionTransformer", "topk": "TopKVisionTransformer", "evit": "EfficientVisionTransformer",
       "tome": "ToMeVisionTransformer", "dyvit": "DynamicVisionTransformer", "sit": "SelfSlimmedVisionTransformer", "dpcknn": "DPCKNNVisionTransformer", "ats": "ATSVisionTransformer", "sinkhorn": "SinkhornVisionTransformer", "kmedoids": "KMedoidsVisionTransformer", "patchmerger": "PatchMergerVisionTransformer", "heuristic": "HeuristicVisionTransformer"}


def build_model(case):
    import tokenreduction_amd as tra
    args = types.SimpleNamespace(keep_rate=list(case["keep_rate"]), reduction_loc=list(case["reduction_loc"]), viz_mode=True,
                                 dyvit_distill=bool(case.get("dyvit_distill", False)), k_neighbors=5, equal_weight=bool(case.get("equal_weight", False)),
                                 sinkhorn_eps=1.0, cluster_iters=3, heuristic_pattern=case.get("heuristic_pattern", "l2"),
                                 not_contiguous=bool(case.get("not_contiguous", False)), min_radius=case.get("min_radius"))
    if "factory" in case:
        m = tra.create_model(case["factory"].replace("_local", "_local_viz") if case["family"] == "deit" else case["factory"],
                             pretrained=False, num_classes=case["num_classes"], drop_rate=0.0,
                             drop_path_rate=0.0, drop_block_rate=None, img_size=case.get("img_size", 224), args=args)
    else:
        cls = getattr(tra, FAM[case["family"]])
        if case.get("dyvit_distill"):
            from functools import partial
            cls = partial(cls, dyvit_distillation=True)
        m = cls(img_size=case.get("img_size", 224), patch_size=16, embed_dim=case["embed_dim"], depth=case["depth"], num_heads=case["num_heads"], mlp_ratio=4,
                qkv_bias=True, num_classes=case["num_classes"], drop_path_rate=float(case.get("drop_path", 0.0)),
                drop_rate=float(case.get("drop_rate", 0.0)), args=args)
    cfg, params = case_params(case)
    m.load_state_dict(params, strict=True)
    m.viz_mode = True
    return m.cuda().eval(), params, cfg


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


# relative L2 of the logits, bf16 HIP path vs the oracle with the same rounding points and the same discrete decisions
FORCED_TOL = 3e-2   # measured 0.6e-2 .. 1.6e-2 over the golden cases (DeiT-S depth 12 without any decision: 1.2e-2)


def _overlap(a, b):
    return float(np.mean([len(set(x.tolist()) & set(y.tolist())) / len(x) for x, y in zip(a, b)]))


@pytest.mark.parametrize("name", [n for n, c in GOLDEN_CASES.items() if not c.get("train_only")])
def test_model_parity(golden_dir, name):
    from tests._stepwise import forward_stepwise
    case = GOLDEN_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model, params, cfg = build_model(case)
    x = make_images(case["batch"], case.get("img_size", 224), case["xseed"])
    noise = {int(k.split("_")[1]): torch.from_numpy(g[k]) for k in g.files if k.startswith("noise_")}
    if noise:
        model.density_noise = noise                      # DPC-KNN: the reference's own torch.rand draws (dpcknn.py:71-72)
    np.random.seed(case["xseed"])          # K-Medoids equal_weight draws its first medoids from numpy's global generator, like the reference
    out = model(x.cuda())
    logits, viz = out
    logits = logits.cpu()
    assert logits.shape == g["logits"].shape and torch.isfinite(logits).all()

    # (1) executor vs stepwise op sequence: bit identical
    from tests._stepwise import Trace
    trace = Trace(keep=True)      # keep=True also makes the ATS leg return the cdf it sampled on
    l2, info = forward_stepwise(model, x.cuda(), trace)
    info["trace"] = trace
    assert torch.equal(l2.cpu(), logits)
    if case["family"] != "ats":
        for blk, idx in info["kept"].items():
            np.testing.assert_array_equal(viz["Kept_Tokens"][blk][:, :idx.shape[1]], idx.cpu().numpy())

    if case["family"] == "tome":
        return _tome_parity(name, case, g, model, params, cfg, x, logits, viz, info)
    if case["family"] == "heuristic":
        keys = sorted((k for k in g.files if k.startswith("keptabs_")), key=lambda k: int(k.split("_")[1]))
        assert sorted(viz["Kept_Tokens_Abs"].keys()) == [int(k.split("_")[1]) for k in keys]
        for k in keys:                                                   # image-independent masks: bit-exact vs the reference
            np.testing.assert_array_equal(viz["Kept_Tokens_Abs"][int(k.split("_")[1])], g[k])
        lb = oracle.forward(params, x, cfg, precision="bf16", extra=case)
        ref = torch.from_numpy(g["logits"])
        rel_bf, rel_ref = ((logits - lb).norm() / lb.norm()).item(), ((logits - ref).norm() / ref.norm()).item()
        print(f"\n[{name}] relative L2 of logits: vs oracle_bf16 {rel_bf:.3e}, vs reference fp32 {rel_ref:.3e}")
        assert rel_bf < FORCED_TOL and rel_ref < 0.05, (rel_bf, rel_ref)   # no data-dependent decision anywhere
        assert model._last_tokens == [197] * cfg.depth
        return
    if case["family"] in ("sit", "sinkhorn", "patchmerger"):
        return _sit_parity(name, case, g, model, params, cfg, x, logits, viz, info)
    if case["family"] in ("dpcknn", "kmedoids"):
        return _dpcknn_parity(name, case, g, model, params, cfg, x, logits, viz, info, noise)
    if case["family"] == "ats":
        return _ats_parity(name, case, g, model, params, cfg, x, logits, viz, info)

    # (2) op-boundary pin: the device's own scores -> oracle selection == device selection, bit exact
    for blk, idx in info["kept"].items():
        sc = info["scores"][blk].cpu()
        want = oracle.cls_topk_select(sc, idx.shape[1])
        np.testing.assert_array_equal(idx.cpu().numpy(), want.numpy())
        if info["compl"][blk] is not None:
            np.testing.assert_array_equal(info["compl"][blk].cpu().numpy(), oracle.complement_idx(want, sc.shape[1]).numpy())

    # (3) shapes / viz contract identical to the reference's
    kept_keys = sorted(k for k in g.files if k.startswith("kept_"))
    assert sorted(viz.get("Kept_Tokens", {}).keys()) == [int(k.split("_")[1]) for k in kept_keys]
    for k in kept_keys:
        blk = int(k.split("_")[1])
        assert viz["Kept_Tokens"][blk].shape == g[k].shape and viz["Kept_Tokens"][blk].dtype == np.int64
        if case["family"] == "evit":
            assert (viz["Kept_Tokens"][blk][:, -1] == -1).all()
            assert viz["Fusion_Assign"][blk].shape == g[f"compl_{blk}"].shape
    if "token_counts" in g.files and case["family"] != "deit":
        for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
            assert model._last_tokens[int(blk)] == int(n)

    # (4) vs oracle with the HIP rounding points, and vs the reference's fp32 golden
    lb, vb = oracle.forward(params, x, cfg, precision="bf16", return_viz=True)
    d_bf = (logits - lb).abs().max().item()
    d_ref = (logits - torch.from_numpy(g["logits"])).abs().max().item()
    ov_bf = [_overlap(viz["Kept_Tokens"][b], vb["Kept_Tokens"][b]) for b in vb["Kept_Tokens"]]
    ov_ref = [_overlap(viz["Kept_Tokens"][int(k.split("_")[1])], g[k]) for k in kept_keys]
    exact_ref = [bool((viz["Kept_Tokens"][int(k.split("_")[1])] == g[k]).all()) for k in kept_keys]
    print(f"\n[{name}] max|logit - oracle_bf16| = {d_bf:.2e}   max|logit - reference_fp32| = {d_ref:.2e}   "
          f"kept-set overlap vs oracle_bf16 {ov_bf} vs reference {ov_ref}  index-exact vs reference {exact_ref}")
    ref = torch.from_numpy(g["logits"])
    rel_bf = ((logits - lb).norm() / lb.norm()).item()
    rel_ref = ((logits - ref).norm() / ref.norm()).item()
    print(f"   relative L2 error of the logits: vs oracle_bf16 {rel_bf:.3e}, vs reference fp32 {rel_ref:.3e}")
    # bf16 pipeline vs fp32 reference on ill-conditioned synthetic models (see module docstring): relative L2 of the whole
    # logit vector is the stable statistic; max-abs over 2000 logits is printed above for information only
    # (5) teacher-forced: the oracle (HIP rounding points) given the DEVICE's selections -- (2) already pinned those bit-exact
    # to the device's own scores -- so what is left is continuous arithmetic only: accumulation order + bf16 rounding flips
    forced = {blk: idx.cpu().long() for blk, idx in info["kept"].items()}
    lf = oracle.forward(params, x, cfg, precision="bf16", forced=forced)
    rel_forced = ((logits - lf).norm() / lf.norm()).item()
    print(f"   teacher-forced (device selections into the oracle_bf16): relative L2 {rel_forced:.3e}, "
          f"max abs {(logits - lf).abs().max().item():.2e}")
    assert rel_forced < FORCED_TOL, rel_forced
    tol = 0.05 if case["embed_dim"] <= 128 and case.get("img_size", 224) == 224 else 0.12
    if case["family"] == "dyvit":
        # the predictor ranks 196 MLP scores whose neighbours are ~1e-6 apart: many more bf16-level flips than CLS-attention
        # top-k, and every flip changes the token set of all later blocks (free-running numbers are informational; the
        # teacher-forced one above and the fp32 path's exact kept sets are the pins)
        # measured 0.16-0.22 (micro), 0.2-0.3 (small) across kernel revisions: the flips move with every rounding change.  0.35 is a
        # coarse REGRESSION bound (it fails when the token sets fork from the first stage on, ~0.6-1.0), not a parity statement
        tol = 0.35
    ov_floor = 0.60
    if min(case["keep_rate"]) <= 0.5 and case["embed_dim"] > 128:
        # north_star's own schedule (DeiT-S keep_rate 0.5: K = 98 / 49 / 24 of 196): the K-th score sits in the dense middle of the CLS-attention
        # distribution -- the best of eight seed pairs has a minimum relative gap of 2e-2 at the boundary (gen_golden.py prints it), the bf16
        # scores carry ~1e-2 -- so a boundary token flips at the first stage on some image and the later stages see another token set.
        # Measured free-running: 0.12 / 0.18 relative L2 vs oracle_bf16 / reference, overlaps 0.99 / 0.69 / 0.63.  The pins of this case are
        # (2) selections bit-exact on the device's own scores, (5) teacher-forced 1.3e-2 < FORCED_TOL, and the fp32 / bf16x3 executors'
        # reference-identical sets (tests/test_hip_fp32.py, tests/test_hip_split.py); the free-running bound is informational.
        tol, ov_floor = 0.25, 0.50
    assert rel_bf < tol, rel_bf
    assert rel_ref < tol, rel_ref
    assert all(o >= ov_floor for o in ov_bf + ov_ref), (ov_bf, ov_ref)
    assert all(o[0] >= 0.95 for o in (ov_bf, ov_ref) if o), (ov_bf, ov_ref)


@pytest.mark.parametrize("name", ["topk_small_kr07", "evit_small_kr05", "tome_small_r16", "deit_small"])
def test_fused_mlp_executor_is_bit_identical(name):
    """The eval executor runs a block's Mlp either as fc1+GELU / fc2 (two GEMM launches) or as ONE launch that keeps the hidden activation on
    the CU (csrc/tr_mlp_fused.hip) -- by default wherever the fused kernel's block schedule fills the chip.  The two are the same arithmetic in
    the same order: logits and every decision must agree BIT FOR BIT with the fused kernel forced on, forced off and on auto -- also after
    an optimizer step has rewritten the bf16 operand copies behind the packed Mlp matrices' back (optim.FusedAdamW)."""
    from tokenreduction_amd import ops
    from tokenreduction_amd.optim import FusedAdamW
    case = GOLDEN_CASES[name]
    model, _, _ = build_model(case)
    model.viz_mode = False
    x = make_images(9, 224, 5).cuda()
    prev = ops.set_mlp_fused(0)
    prev_rl = ops.set_mlp_resid_ln(False)          # (the fused block TAIL is a different rounding sequence: held to a tolerance below)
    try:
        def run():
            out = {}
            for mode in (0, 1, -1):
                ops.set_mlp_fused(mode)
                model._ws = {}                                  # drop captured graphs: the mode is read when the launches are enqueued
                out[mode] = model(x).clone()
            return out
        a = run()
        assert torch.isfinite(a[0]).all()
        assert torch.equal(a[0], a[1]) and torch.equal(a[0], a[-1]), float((a[0] - a[1]).abs().max())
        # one training step (the optimizer refreshes the bf16 copies itself), then eval again: the packed Mlp copies must have followed
        model.train()
        opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.05, model=model)
        loss = model(x).square().mean()
        loss.backward()
        opt.step()
        model.eval()
        b = run()
        assert not torch.equal(a[0], b[0]), "the step did not change the weights?"
        assert torch.equal(b[0], b[1]) and torch.equal(b[0], b[-1]), float((b[0] - b[1]).abs().max())
        # the fused block tail (Mlp + residual add + next norm1 in one launch, tr_mlp_fused_resid_ln_bf16): the fc2 output enters the stream
        # WITHOUT its bf16 rounding and the norm's statistics are summed in another order -- not the same bits, the same function: within the
        # bf16 executor's own noise floor of the two-launch form (FORCED_TOL is that floor against the oracle), and itself deterministic
        ops.set_mlp_resid_ln(True)
        ops.set_mlp_fused(1)
        model._ws = {}
        c1 = model(x).clone()
        model._ws = {}
        c2 = model(x).clone()
        assert torch.equal(c1, c2), "the fused block tail is not deterministic"
        assert torch.isfinite(c1).all()
        rel = float((c1 - b[0]).norm() / b[0].norm())
        assert rel < 2e-2, f"fused block tail: logits differ from the two-launch form by rel L2 {rel:.3g}"
    finally:
        ops.set_mlp_fused(prev)
        ops.set_mlp_resid_ln(prev_rl)


def test_fused_mlp_follows_reloaded_weights():
    """The packed Mlp copies (tr_mlp_pack_bf16) are derived state: after load_state_dict on a model that has already run, after an in-place edit
    announced by weights_changed(), and in a deep copy (harness.ModelEma) the fused launch must compute with the CURRENT weights -- each time
    equal, bit for bit, to the two-GEMM form on the same weights."""
    import copy
    from tokenreduction_amd import ops
    case = GOLDEN_CASES["topk_small_kr07"]
    model, params, _ = build_model(case)
    model.viz_mode = False
    x = make_images(5, 224, 3).cuda()
    prev = ops.set_mlp_fused(1)
    try:
        def both(m):
            ops.set_mlp_fused(1)
            m._ws = {}
            a = m(x).clone()
            ops.set_mlp_fused(0)
            m._ws = {}
            b = m(x).clone()
            ops.set_mlp_fused(1)
            return a, b
        a0, b0 = both(model)
        assert torch.equal(a0, b0)
        other = {k: (v + 0.01 * torch.randn(v.shape, generator=torch.Generator().manual_seed(9)).to(v.dtype) if v.is_floating_point() and "mlp" in k else v)
                 for k, v in params.items()}
        model.load_state_dict(other, strict=True)
        a1, b1 = both(model)
        assert torch.equal(a1, b1) and not torch.equal(a1, a0), "the fused launch kept the old Mlp weights after load_state_dict"
        with torch.no_grad():
            model.blocks[0].mlp.fc2.weight.mul_(1.5)
        model.weights_changed()
        a2, b2 = both(model)
        assert torch.equal(a2, b2) and not torch.equal(a2, a1), "the fused launch kept the old Mlp weights after weights_changed()"
        twin = copy.deepcopy(model)
        a3, b3 = both(twin)
        assert torch.equal(a3, b3) and torch.equal(a3, a2), "a deep copy does not compute with its own packed Mlp weights"
    finally:
        ops.set_mlp_fused(prev)


@pytest.mark.parametrize("name", ["ats_micro", "ats_small_kr07", "ats_small_kr05"])
def test_ats_dynamic_width(name):
    """model.dynamic_width (opt-in): after every sampling block the executor keeps the BATCH MAXIMUM of unique ids like the reference
    (ats.py:77-78) instead of the static bound.  The rows it drops are the masked padding rows of the static run, so: the token counts are
    1 + the Kept_Tokens widths, never above the static counts; the sampled ids are the static run's (later stages: the attention sums run over
    fewer zero terms, so a near-tie may flip -- none does on these fixtures); the logits agree to the bf16 executor's summation-order noise;
    and the switch is eval-only and reversible."""
    from tokenreduction_amd import ops
    case = GOLDEN_CASES[name]
    model, _, _ = build_model(case)
    x = make_images(case["batch"], case.get("img_size", 224), case["xseed"]).cuda()
    out_s = model(x)
    logits_s, viz_s = (out_s[0], out_s[1]) if isinstance(out_s, tuple) else (out_s, None)
    tok_s = list(model._last_tokens)
    model.dynamic_width = True
    out_d = model(x)
    logits_d, viz_d = (out_d[0], out_d[1]) if isinstance(out_d, tuple) else (out_d, None)
    tok_d = list(model._last_tokens)
    assert all(d <= s_ for d, s_ in zip(tok_d, tok_s)) and tok_d != tok_s, (tok_d, tok_s)
    for blk, kept in viz_s["Kept_Tokens"].items():
        np.testing.assert_array_equal(viz_d["Kept_Tokens"][blk], kept)
        assert tok_d[blk] == 1 + kept.shape[1], (blk, tok_d[blk], kept.shape)
    rel = float((logits_d - logits_s).norm() / logits_s.norm())
    assert torch.isfinite(logits_d).all() and rel < 5e-3, rel
    model.dynamic_width = False
    again = model(x)
    assert torch.equal((again[0] if isinstance(again, tuple) else again), logits_s) and list(model._last_tokens) == tok_s
    # the two ops behind it
    m = torch.zeros(5, 9, device="cuda")
    m[:, 0] = 1
    m[2, 1:6] = 1
    m[4, 1:3] = 1
    assert ops.ats_width(m) == 6
    ids = torch.arange(45, dtype=torch.int32, device="cuda").view(5, 9)
    i2, m2 = ops.ats_narrow(ids, m, 6)
    assert torch.equal(i2, ids[:, :6].contiguous()) and torch.equal(m2, m[:, :6].contiguous())


def _ats_parity(name, case, g, model, params, cfg, x, logits, viz, info):
    """ATS leg: executor == stepwise ids; the sampling op pinned bit-exact on the device's own cdf; teacher-forced logits."""
    from tests._params import assert_valid_sampling
    counts = oracle.ats_sample_counts(cfg)
    forced = {}
    for blk, ids in info["kept"].items():
        ids = ids.cpu().long()
        K = counts[blk]
        assert ids.shape == (x.shape[0], K) and (ids[:, 0] == 0).all()
        width = int((ids[:, 1:] != 0).sum(1).max())
        np.testing.assert_array_equal(viz["Kept_Tokens"][blk], (ids[:, 1:1 + width] - 1).numpy())      # executor == stepwise
        # op-boundary pin: torch's own cdist/argmin/unique on the DEVICE's cdf gives the device's ids, bit exact
        want, _ = oracle.ats_ids_from_cdf(info["scores"][blk].cpu(), oracle.ats_sample_steps(K), pad_to=K)
        np.testing.assert_array_equal(ids.numpy(), want.numpy())
        forced[blk] = ids
        assert model._last_tokens[blk] == K
    kept_keys = sorted((k for k in g.files if k.startswith("kept_")), key=lambda k: int(k.split("_")[1]))
    assert sorted(viz["Kept_Tokens"].keys()) == [int(k.split("_")[1]) for k in kept_keys]
    blk0 = int(kept_keys[0].split("_")[1])
    # first stage: same token set as the reference -> the device's samples must be valid samples of the reference's cdf up to
    # bf16 noise in the cdf (1e-2) -- coarse; the fp32 path holds this at 5e-4
    assert_valid_sampling(viz["Kept_Tokens"][blk0], g[f"cdf_{blk0}"], oracle.ats_sample_steps(counts[blk0]).numpy(), tol=2e-2)
    lb = oracle.ats_forward(params, x, cfg, precision="bf16")
    lf = oracle.ats_forward(params, x, cfg, precision="bf16", forced=forced)
    ref = torch.from_numpy(g["logits"])
    rel_bf = ((logits - lb).norm() / lb.norm()).item()
    rel_ref = ((logits - ref).norm() / ref.norm()).item()
    rel_forced = ((logits - lf).norm() / lf.norm()).item()
    print(f"\n[{name}] relative L2 of logits: vs oracle_bf16 {rel_bf:.3e}, vs reference fp32 {rel_ref:.3e}, teacher-forced ids "
          f"{rel_forced:.3e}; tokens {model._last_tokens}")
    assert rel_forced < FORCED_TOL, rel_forced
    # free-running: informational only.  bf16 noise in the cdf (~1e-3) against a grid spacing of ~7e-3 moves ~15 % of the
    # samples to a neighbouring token, and on random-weight models neighbours are unrelated; the pins are the op-boundary
    # equality above, the teacher-forced logits, and the fp32 path (ids identical to the reference on the DeiT-S cases)
    # -> NO free-running bound is asserted here (a bound that cannot fail is not a test); the numbers above are printed for the record
    assert torch.isfinite(logits).all()


def _dpcknn_parity(name, case, g, model, params, cfg, x, logits, viz, info, noise):
    """DPC-KNN leg: executor == stepwise on centres/assignment; decisions teacher-forced into the oracle for the logits."""
    forced = {}
    for blk, centers in info["kept"].items():
        np.testing.assert_array_equal(viz["Kept_Tokens"][blk], centers.cpu().numpy())
        np.testing.assert_array_equal(viz["Assignment_Maps"][blk], info["compl"][blk].cpu().numpy())
        forced[blk] = centers.cpu().long()
        a = viz["Assignment_Maps"][blk]
        K = centers.shape[1]
        assert a.min() >= 0 and a.max() < K
        if case["family"] == "dpcknn":
            # every centre is assigned to itself, every cluster is non-empty (dpcknn.py:95-98)
            np.testing.assert_array_equal(np.take_along_axis(a, viz["Kept_Tokens"][blk], axis=1),
                                          np.broadcast_to(np.arange(K), (a.shape[0], K)))
    kept_keys = sorted((k for k in g.files if k.startswith("kept_")), key=lambda k: int(k.split("_")[1]))
    assert sorted(viz["Kept_Tokens"].keys()) == [int(k.split("_")[1]) for k in kept_keys]
    for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
        assert model._last_tokens[int(blk)] == int(n)
    extra = None
    if case["family"] == "kmedoids" and case.get("equal_weight"):
        extra = {"equal_first": dict(zip(sorted(oracle.dpcknn_cluster_counts(cfg)), g["first_medoid"].tolist()))}
        assert list(model._kmed_draws) == g["first_medoid"].tolist()          # same numpy stream as the reference run
    lb, vb = oracle.forward(params, x, cfg, precision="bf16", return_viz=True, noise=noise, extra=extra)
    if case["family"] == "dpcknn":
        # the token -> centre assignment is a discrete decision too (a token between two centres flips with bf16-level changes of
        # the residual stream): given the device's centres the oracle's own nearest-centre map must agree almost everywhere at the
        # first stage (same input up to rounding), and the logits are compared with centres AND assignment forced
        _, vc = oracle.forward(params, x, cfg, precision="bf16", return_viz=True, forced=forced, noise=noise)
        first = min(forced)
        agree = float((vc["Assignment_Maps"][first] == viz["Assignment_Maps"][first]).mean())
        print(f"   first-stage assignment agreement with the oracle given the device's centres: {agree:.4f}")
        assert agree > 0.97, agree
        forced = {blk: (c, torch.from_numpy(viz["Assignment_Maps"][blk]).long()) for blk, c in forced.items()}
    lf = oracle.forward(params, x, cfg, precision="bf16", forced=forced, noise=noise, extra=extra)
    ref = torch.from_numpy(g["logits"])
    rel_bf = ((logits - lb).norm() / lb.norm()).item()
    rel_ref = ((logits - ref).norm() / ref.norm()).item()
    rel_forced = ((logits - lf).norm() / lf.norm()).item()
    ov_bf = [_overlap(viz["Kept_Tokens"][b], vb["Kept_Tokens"][b]) for b in sorted(vb["Kept_Tokens"])]
    ov_ref = [_overlap(viz["Kept_Tokens"][int(k.split("_")[1])], g[k]) for k in kept_keys]
    print(f"\n[{name}] relative L2 of logits: vs oracle_bf16 {rel_bf:.3e}, vs reference fp32 {rel_ref:.3e}, teacher-forced decisions "
          f"{rel_forced:.3e}; centre-set overlap vs oracle_bf16 {ov_bf} vs reference {ov_ref}")
    assert rel_forced < FORCED_TOL, rel_forced
    # free-running: informational (see the DyViT note above); against the fp32 reference the 9-token end of a keep_rate 0.25
    # schedule shares almost no medoid once the sets fork, so only the same-rounding oracle is bounded
    # (measured 0.1-0.43 across kernel revisions: a medoid flip at stage 2 re-seeds stage 3) -> NO free-running logit bound is asserted (a
    # bound that cannot fail is not a test); what IS held free-running is the first stage's centre sets, below
    assert torch.isfinite(logits).all()
    assert ov_bf[0] >= 0.9 and ov_ref[0] >= 0.9, (ov_bf, ov_ref)


def _sit_parity(name, case, g, model, params, cfg, x, logits, viz, info):
    """SiT leg: no discrete decision anywhere, so the free-running logits are held to the teacher-forced tolerance."""
    for blk, soft in info["soft"].items():
        np.testing.assert_array_equal(viz["Soft_Assignment_Maps"][blk], soft.cpu().numpy())     # executor == stepwise
    lb, vb = oracle.forward(params, x, cfg, precision="bf16", return_viz=True)
    ref = torch.from_numpy(g["logits"])
    rel_bf = ((logits - lb).norm() / lb.norm()).item()
    rel_ref = ((logits - ref).norm() / ref.norm()).item()
    akeys = sorted((k for k in g.files if k.startswith("assign_")), key=lambda k: int(k.split("_")[1]))
    assert sorted(viz["Assignment_Maps"].keys()) == [int(k.split("_")[1]) for k in akeys]
    ag, dsoft = [], []
    for k in akeys:
        blk = int(k.split("_")[1])
        got = viz["Soft_Assignment_Maps"][blk]
        assert viz["Assignment_Maps"][blk].shape == g[k].shape and viz["Assignment_Maps"][blk].dtype == np.int64
        if case["family"] in ("sit", "patchmerger"):
            np.testing.assert_allclose(got.sum(axis=2), 1.0, atol=1e-5)         # softmax over the token axis
        else:                                                                   # Sinkhorn plan: column marginals ~ (K+P)/(K+P) = 1
            K_, P_ = got.shape[1], got.shape[2]
            np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-4)         # last half-iteration normalises the token marginal
            assert abs(got.sum() / got.shape[0] - P_) < 1e-2 * P_ and K_ > 0
        dsoft.append(float(np.abs(got - vb["Soft_Assignment_Maps"][blk]).max() / vb["Soft_Assignment_Maps"][blk].max()))
        ag.append(_agree(viz["Assignment_Maps"][blk], g[k]))
    print(f"\n[{name}] relative L2 of logits: vs oracle_bf16 {rel_bf:.3e}, vs reference fp32 {rel_ref:.3e}; soft assignment "
          f"max err / max vs oracle_bf16 {dsoft}; hard-assignment agreement vs reference {ag}")
    for blk, n in zip(g["token_count_blocks"], g["token_counts"]):
        assert model._last_tokens[int(blk)] == int(n)
    assert rel_bf < FORCED_TOL, rel_bf
    assert rel_ref < 0.05, rel_ref
    # hard assignment = argmax over K clusters of nearly equal soft weights at the late stages: noise-dominated in bf16
    # (printed); the fp32 path pins it (tests/test_hip_fp32.py)
    assert max(dsoft) < 0.05 and ag[0] > 0.9, (dsoft, ag)


def _agree(a, b):
    return float((np.asarray(a) == np.asarray(b)).mean())


def _tome_parity(name, case, g, model, params, cfg, x, logits, viz, info):
    """ToMe leg of test_model_parity: executor == stepwise on (unm, src, dst); viz contract; bf16 statistics."""
    B = x.shape[0]
    n_in = 197
    sched = oracle.tome_schedule(cfg)
    forced = {}
    for blk in range(cfg.depth):
        r = oracle.tome_block_r(sched.get(blk, 0), n_in)
        if r > 0:
            unm, src, dst = info["tome"][blk]
            forced[blk] = (unm.cpu().long(), src.cpu().long(), dst.cpu().long())
            # op-boundary pin: the oracle's matching on the device's own K (bf16) == the device's matching, bit exact
            k = info["trace"].tensors[f"qkv_{blk}"].float().cpu().reshape(B, n_in, 3, cfg.num_heads, 64)[:, :, 1].permute(0, 2, 1, 3)
            for got, want in zip(forced[blk], oracle.tome_match(k.mean(1), r)):
                np.testing.assert_array_equal(got.numpy(), want.numpy())
            a = oracle.tome_assignment(unm.cpu().long(), src.cpu().long(), dst.cpu().long(), n_in).numpy()
            np.testing.assert_array_equal(viz["Assignment_Maps"][blk], a)      # executor slab == stepwise, via the oracle's map
            # structure: unm ascending with CLS first, src/unm a partition of the even set, dst inside the odd set
            u, s_ = unm.cpu().numpy(), src.cpu().numpy()
            assert (u[:, 0] == 0).all() and (np.diff(u, axis=1) > 0).all()
            both = np.sort(np.concatenate([u, s_], axis=1), axis=1)
            np.testing.assert_array_equal(both, np.broadcast_to(np.arange((n_in + 1) // 2), both.shape))
            assert (dst.cpu().numpy() >= 0).all() and (dst.cpu().numpy() < n_in // 2).all()
        n_in -= r
        assert model._last_tokens[blk] == n_in
    akeys = sorted((k for k in g.files if k.startswith("assign_")), key=lambda k: int(k.split("_")[1]))
    assert sorted(viz["Assignment_Maps"].keys()) == [int(k.split("_")[1]) for k in akeys]
    for k in akeys:
        blk = int(k.split("_")[1])
        assert viz["Assignment_Maps"][blk].shape == g[k].shape and viz["Assignment_Maps"][blk].dtype == np.int64
    lb, vb = oracle.tome_forward(params, x, cfg, precision="bf16", return_viz=True)
    ref = torch.from_numpy(g["logits"])
    rel_bf = ((logits - lb).norm() / lb.norm()).item()
    rel_ref = ((logits - ref).norm() / ref.norm()).item()
    ag_bf = [_agree(viz["Assignment_Maps"][b], vb["Assignment_Maps"][b]) for b in sorted(vb["Assignment_Maps"])]
    ag_ref = [_agree(viz["Assignment_Maps"][int(k.split("_")[1])], g[k]) for k in akeys]
    print(f"\n[{name}] relative L2 of logits: vs oracle_bf16 {rel_bf:.3e}, vs reference fp32 {rel_ref:.3e}; "
          f"assignment agreement vs oracle_bf16 {ag_bf} vs reference {ag_ref}")
    # teacher-forced: the oracle (HIP rounding points) given the DEVICE's merge decisions, which the loop above pinned
    # bit-exact to the device's own K -- continuous arithmetic only
    lf = oracle.tome_forward(params, x, cfg, precision="bf16", forced=forced)
    rel_forced = ((logits - lf).norm() / lf.norm()).item()
    print(f"   teacher-forced (device matchings into the oracle_bf16): relative L2 {rel_forced:.3e}, "
          f"max abs {(logits - lf).abs().max().item():.2e}")
    assert rel_forced < FORCED_TOL, rel_forced
    # free-running: every flipped merge changes the token set of all later blocks, and these synthetic models (random
    # weights, qkv_gain 4-6) are ill-conditioned, so free-running logits are informative only at the coarse level; the
    # fp32 validation path (tests/test_hip_fp32.py) is the end-to-end pin: every assignment map bit-exact
    tol = 0.05 if case["embed_dim"] <= 128 else 0.35
    assert rel_bf < tol and rel_ref < tol, (rel_bf, rel_ref)
    # the first merge sees identical inputs up to bf16 rounding; later stages drift with the token set (printed above)
    assert ag_bf[0] >= 0.90 and ag_ref[0] >= 0.80, (ag_bf, ag_ref)


def test_batch_independence():
    """Images are independent in every op: an image's logits do not depend on its batch."""
    case = GOLDEN_CASES["topk_micro"]
    model, _, _ = build_model(case)
    model.viz_mode = False
    x = make_images(5, 224, 99).cuda()
    full = model(x)
    for b in (0, 3):
        single = model(x[b:b + 1].contiguous())
        assert torch.equal(single[0], full[b])


def test_repack_after_weight_update():
    case = GOLDEN_CASES["deit_micro"]
    model, params, _ = build_model(case)
    model.viz_mode = False
    x = make_images(2, 224, 5).cuda()
    a = model(x).clone()
    with torch.no_grad():
        model.head.bias.add_(1.0)
    b = model(x)
    torch.testing.assert_close(b, a + 1.0, atol=1e-6, rtol=0)


@pytest.mark.parametrize("name,kr,loc", [("deit_small_patch16_224_local", [1.0], []), ("topk_small_patch16_224", [0.7], [3, 6, 9])])
def test_bf16_path_stays_near_the_fp32_executor_on_a_well_conditioned_model(name, kr, loc):
    """An ABSOLUTE bound for the timed (bf16) path, so that a numerical regression in it fails a test: DeiT-S with the reference's own
    initialisation (trunc_normal(0.02), zero biases -- topk.py:163-176; no qkv gain: near-uniform attention, the well-conditioned case)
    against the same executor in fp32 arithmetic.  Measured (bench.py drift_vs_fp32_path.plain_init): max-abs 1.3e-2 dense / 2.7e-2
    Top-K, top-1 agreement 98-100 %; bf16x3 2.3e-5.  Asserted: < 5e-2 / relative L2 < 3e-2 / agreement >= 0.9; bf16x3 < 1e-3."""
    import tokenreduction_amd as tra
    torch.manual_seed(0)
    args = types.SimpleNamespace(keep_rate=list(kr), reduction_loc=list(loc))
    model = tra.create_model(name, pretrained=False, num_classes=1000, args=args).cuda().eval()
    x = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(3)).cuda()
    model.precision = "fp32"
    lf = model(x).float().clone()
    model.precision = "bf16"
    lb = model(x).float().clone()
    model.precision = "bf16x3"
    l3 = model(x).float().clone()
    model.precision = "bf16"
    max_abs, rel = (lb - lf).abs().max().item(), ((lb - lf).norm() / lf.norm()).item()
    agree = (lb.argmax(1) == lf.argmax(1)).float().mean().item()
    print(f"   {name}: bf16 vs fp32 executor max-abs {max_abs:.2e}, relative L2 {rel:.2e}, top-1 agreement {agree:.3f}; "
          f"bf16x3 max-abs {(l3 - lf).abs().max().item():.2e}")
    assert max_abs < 5e-2 and rel < 3e-2 and agree >= 0.9, (max_abs, rel, agree)
    assert (l3 - lf).abs().max().item() < 1e-3


def test_full_size_batch_properties():
    """BASELINE configs[1] at its full size (DeiT-S Top-K kr 0.7, batch 256): size-independent properties --
    finite logits, every kept index in range and unique per image, descending-score order, batch independence."""
    import tokenreduction_amd as tra
    from tests._stepwise import forward_stepwise
    args = types.SimpleNamespace(keep_rate=[0.7], reduction_loc=[3, 6, 9], viz_mode=True)
    model = tra.create_model("topk_small_patch16_224", args=args)
    case = GOLDEN_CASES["topk_small_kr07"]
    model.load_state_dict(make_params(case_config(case), case["wseed"], case["qkv_gain"]))
    model = model.cuda().eval()
    x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
    logits, viz = model(x)
    assert logits.shape == (256, 1000) and torch.isfinite(logits).all()
    n_in = 196
    for blk, K in ((3, 137), (6, 96), (9, 67)):
        idx = viz["Kept_Tokens"][blk]
        assert idx.shape == (256, K) and idx.min() >= 0 and idx.max() < n_in
        assert all(len(set(r.tolist())) == K for r in idx)
        n_in = K
    l2, info = forward_stepwise(model, x[:4].contiguous())
    assert torch.equal(l2, logits[:4])
    for blk, idx in info["kept"].items():
        sc = torch.gather(info["scores"][blk], 1, idx.long())
        assert (sc[:, :-1] >= sc[:, 1:]).all()          # sorted=True: descending score order


def test_forward_is_graph_capturable():
    """tr_vit_forward only enqueues kernels (no allocation, no sync, no host readback), so a forward can be captured in a
    hipGraph and replayed: same logits bit for bit, also after the input buffer's contents change."""
    case = GOLDEN_CASES["topk_micro"]
    model, _, _ = build_model(case)
    model.viz_mode = False
    x = make_images(4, 224, 7).cuda()
    ref = model(x).clone()                       # also warms the workspace / weight pack
    static_x = x.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(static_x)                          # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model(static_x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    x2 = make_images(4, 224, 8).cuda()
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, model(x2))


@pytest.mark.parametrize("name", ["deit_micro", "topk_micro", "evit_micro", "ats_micro", "tome_micro", "dyvit_micro", "sit_micro", "dpcknn_micro",
                                  "kmedoids_micro", "sinkhorn_micro", "patchmerger_micro", "heuristic_micro_l2", "topk_small_kr07",
                                  "dpcknn_base_kr05", "topk_micro_384"])
def test_lazy_norm2_is_bit_identical_to_the_eager_sequence(golden_dir, name):
    """Without viz_mode the eval executor skips the stream write of every norm2 that no reduction follows and lets the next norm1 (or the
    final norm) absorb both pending residuals (tr_layernorm2_bf16); with viz_mode (`Features` wants the stream after every block) it
    runs the eager sequence.  Same fp32 additions in the same order: the logits must agree bit for bit, for every family."""
    case = GOLDEN_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model, _, _ = build_model(case)
    noise = {int(k.split("_")[1]): torch.from_numpy(g[k]) for k in g.files if k.startswith("noise_")}
    if noise:
        model.density_noise = noise
    x = make_images(case["batch"], case.get("img_size", 224), case["xseed"]).cuda()
    model.viz_mode = True
    eager = model(x)[0].clone()
    model.viz_mode = False
    lazy = model(x).clone()
    assert torch.equal(lazy, eager)


@pytest.mark.parametrize("name,batch", [("topk_small_kr07", 256), ("topk_small_kr07", 130), ("evit_small_kr05", 200)])
def test_norm2_inside_the_fused_mlp_gives_the_same_logits(golden_dir, name, batch):
    """tr_set_mlp_ln: the executor runs a lazy norm2 inside the fused Mlp launch that follows it (one-round launches by default, everywhere
    with mode 2) or as its own LayerNorm launch (mode 0) -- bit-identical by construction, on DeiT-S shapes where the fused Mlp is in play
    (batch 256: stream-K stages and a single-round stage; 130 / 200: single rounds of other fill)."""
    from tokenreduction_amd import ops
    case = GOLDEN_CASES[name]
    x = make_images(batch, 224, case["xseed"]).cuda()
    outs = []
    prev = ops.set_mlp_ln(0)
    try:
        for mode in (0, 1, 2):
            ops.set_mlp_ln(mode)
            model, _, _ = build_model(case)          # a fresh model: a captured graph keeps the form it was captured with
            model.viz_mode = False
            outs.append(model(x).clone())
            model.check_status()
    finally:
        ops.set_mlp_ln(prev)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("name", ["topk_small_kr07", "topk_micro", "tome_small_r16", "dpcknn_micro", "sinkhorn_micro", "ats_micro", "deit_micro",
                                  "evit_micro", "dyvit_micro", "sit_micro", "kmedoids_micro", "patchmerger_micro", "heuristic_micro_l2",
                                  "evit_small_kr05", "topk_micro_384"])
def test_forward_async_equals_forward_with_two_forwards_in_flight(name):
    """model.forward_async enqueues eval forwards on two side streams (own workspace and captured graph each), so that two are in flight;
    `handle.result()` makes the caller's stream wait and returns what model(x) returns.  Same kernels: the logits of six batches launched
    back to back -- two alternating inputs, results taken out of order, the caller's stream busy meanwhile -- equal model(x) BIT FOR BIT;
    the status check covers every slot; a deep copy (ModelEma) starts without the streams; viz_mode runs synchronously."""
    import copy
    case = GOLDEN_CASES[name]
    model, _, _ = build_model(case)
    model.viz_mode = False
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    noise = {int(k.split("_")[1]): torch.from_numpy(g[k]) for k in g.files if k.startswith("noise_")}
    if noise:
        model.density_noise = noise
    B = 64 if case["embed_dim"] >= 384 else case["batch"]
    if noise:
        B = case["batch"]                       # (the recorded noise has the fixture's batch size)
    xa, xb = make_images(B, case.get("img_size", 224), 31).cuda(), make_images(B, case.get("img_size", 224), 32).cuda()
    want_a, want_b = model(xa).clone(), model(xb).clone()
    busy = torch.zeros(1 << 22, device="cuda")
    handles = []
    for k in range(6):
        handles.append(model.forward_async(xa if k % 2 == 0 else xb))
        busy.add_(1.0)                           # work on the caller's stream between the launches
    for k in (5, 0, 3, 2, 4, 1):                 # results in any order
        got = handles[k].result()
        assert torch.equal(got, want_a if k % 2 == 0 else want_b), f"forward {k} differs from model(x)"
    torch.cuda.synchronize()
    model.check_status()
    assert len(model._ws) >= 3                   # slot 0 (model(x)) and the two side slots
    ema = copy.deepcopy(model)
    assert not ema.__dict__.get("_pipe_streams") and ema._ws == {}
    assert torch.equal(ema.cuda().eval().forward_async(xa).result(), want_a)
    model.viz_mode = True                        # (Features wants every block's stream: the eager, synchronous path)
    out = model.forward_async(xa).result()
    assert isinstance(out, tuple) and torch.equal(out[0], want_a)


def test_graph_replay_gives_way_to_plain_launches_when_inputs_keep_moving():
    """The eval forward replays a hipGraph keyed on the input's address.  A caller whose batches land at a new address every time would
    re-capture on every call: after GRAPH_MISS_LIMIT misses in a row the workspace goes back to plain launches, with one warning; a caller
    that keeps one static input buffer stays on the replay.  Same logits either way."""
    case = GOLDEN_CASES["topk_micro"]
    model, _, _ = build_model(case)
    model.viz_mode = False
    x = make_images(3, 224, 11).cuda()
    want = model(x).clone()
    for _ in range(5):
        assert torch.equal(model(x), want)                   # one address: hits after the first capture
    ws = model._last_ws
    assert not ws.get("graph_off") and len(ws["graphs"]) == 1
    keep = []                                                # hold the copies so the allocator cannot hand an address out twice
    with pytest.warns(RuntimeWarning, match="hipGraph replay is off"):
        for _ in range(model.GRAPH_MISS_LIMIT + 1):
            keep.append(x.clone())
            assert torch.equal(model(keep[-1]), want)
    assert ws.get("graph_off") and not ws["graphs"]
    assert torch.equal(model(x), want)                       # plain launches from here on


def test_graph_replay_survives_a_ring_of_static_input_buffers():
    """A caller that rotates a few static input buffers (prefetch ring) misses once per buffer on its first pass and hits ever after: the
    first-pass misses must not switch the replay off (GRAPH_MISS_LIMIT used to be 4, below the cache's 8 entries)."""
    case = GOLDEN_CASES["topk_micro"]
    model, _, _ = build_model(case)
    model.viz_mode = False
    ring = [make_images(3, 224, 20 + i).cuda() for i in range(model.GRAPH_CACHE)]
    want = [model(x).clone() for x in ring]                  # GRAPH_CACHE consecutive misses
    ws = model._last_ws
    assert not ws.get("graph_off") and len(ws["graphs"]) == model.GRAPH_CACHE
    for _ in range(2):
        for x, w in zip(ring, want):
            assert torch.equal(model(x), w)
    assert not ws.get("graph_off") and ws["graph_misses"] == 0


def test_repack_after_a_parameter_moved_drops_the_captured_graphs():
    """Biases and LayerNorm parameters are read through the parameter's own storage: giving one a NEW storage (p.data = ..., an optimizer
    that flattens its parameters) must drop the workspaces and their captured graphs, which hold the old address."""
    case = GOLDEN_CASES["topk_micro"]
    model, _, _ = build_model(case)
    model.viz_mode = False
    x = make_images(3, 224, 12).cuda()
    before = model(x).clone()
    assert torch.equal(model(x), before)                     # replayed
    fresh, _, _ = build_model(case)
    fresh.viz_mode = False
    with torch.no_grad():
        for m in (model, fresh):
            new_b = m.blocks[1].norm1.bias.data.clone() + 0.25
            new_fc = m.blocks[0].mlp.fc2.bias.data.clone() - 0.5
            if m is model:
                m.blocks[1].norm1.bias.data = new_b          # a NEW storage, same shape: no slot moves, no version bump on the old tensor
                m.blocks[0].mlp.fc2.bias.data = new_fc
            else:
                m.blocks[1].norm1.bias.copy_(new_b)
                m.blocks[0].mlp.fc2.bias.copy_(new_fc)
    got, want = model(x), fresh(x)
    assert not torch.equal(got, before)
    assert torch.equal(got, want)


def test_pretrained_checkpoint_through_the_factory_runs_on_the_hip_path(tmp_path, monkeypatch):
    """f3, positive path (models_act.py:1130-1137): a DeiT-layout ./deit_weights/<name>.pth is ingested by
    create_model(..., pretrained=True) and the HIP forward on the ingested weights agrees with the oracle on the SAME weights --
    Top-K selections bit-exact on the device's scores, logits teacher-forced within FORCED_TOL."""
    import tokenreduction_amd as tra
    from tokenreduction_amd.registry import deit_url_paths
    from tests._stepwise import forward_stepwise
    monkeypatch.chdir(tmp_path)
    cfg = oracle.VitConfig(family="topk", embed_dim=192, depth=12, num_heads=3, num_classes=1000, keep_rate=[0.7], reduction_loc=[3, 6, 9])
    params = make_params(cfg, 4242, 2.0)
    os.makedirs("deit_weights")
    torch.save({"model": params}, os.path.join("deit_weights", os.path.basename(deit_url_paths["deit_tiny_patch16_224"])))
    args = types.SimpleNamespace(keep_rate=[0.7], reduction_loc=[3, 6, 9], viz_mode=True)
    model = tra.create_model("topk_tiny_patch16_224", pretrained=True, args=args).cuda().eval()
    assert all(torch.equal(v.cpu(), params[k]) for k, v in model.state_dict().items())
    x = make_images(2, 224, 99)
    logits, viz = model(x.cuda())
    l2, info = forward_stepwise(model, x.cuda())
    assert torch.equal(l2, logits)
    forced = {}
    for blk, idx in info["kept"].items():
        assert torch.equal(idx.cpu().long(), oracle.cls_topk_select(info["scores"][blk].cpu(), idx.shape[1]))
        forced[blk] = idx.cpu().long()
    want = oracle.vit_forward(params, x, cfg, precision="bf16", forced=forced)
    err = float((logits.cpu() - want).norm() / want.norm())
    assert err < FORCED_TOL, err


def test_dyvit_teacher_returns_logits_and_normed_tokens():
    """VisionTransformerTeacher.forward dyvit.py:325-334: (head(norm(x)[:, 0]), norm(x)[:, 1:])."""
    import tokenreduction_amd as tra
    case = GOLDEN_CASES["deit_micro"]
    cfg, params = case_params(case)
    m = tra.VisionTransformerTeacher(patch_size=16, embed_dim=case["embed_dim"], depth=case["depth"], num_heads=case["num_heads"],
                                     mlp_ratio=4, qkv_bias=True, num_classes=case["num_classes"])
    m.load_state_dict(params, strict=True)
    m = m.cuda().eval()
    x = make_images(2, 224, 3)
    logits, tokens = m(x.cuda())
    lb, vb = oracle.vit_forward(params, x, cfg, precision="bf16", return_viz=True)
    want = oracle.layer_norm(vb["Final_Tokens"], params["norm.weight"], params["norm.bias"], cfg.ln_eps)[:, 1:]
    assert tokens.shape == want.shape == (2, 196, case["embed_dim"])
    assert ((logits.cpu() - lb).norm() / lb.norm()).item() < FORCED_TOL
    assert ((tokens.cpu() - want).norm() / want.norm()).item() < FORCED_TOL


@pytest.mark.parametrize("family", ["dyvit", "sit"])
def test_tiny_width_predictor_modules(family):
    """DeiT-T (D = 192): the D/2 = 96-wide hidden layer of the DyViT predictor / SiT slimming MLP is packed zero-padded to 128 so
    the bf16 GEMMs keep K %% 64; results must not notice."""
    from tests._stepwise import forward_stepwise
    case = dict(family=family, embed_dim=192, depth=3, num_heads=3, num_classes=16, keep_rate=[0.6], reduction_loc=[1, 2],
                batch=2, wseed=901, xseed=902, qkv_gain=6.0)
    model, params, cfg = build_model(case)
    x = make_images(2, 224, case["xseed"])
    logits, viz = model(x.cuda())
    l2, info = forward_stepwise(model, x.cuda())
    assert torch.equal(l2.cpu(), logits.cpu())
    forced = {blk: idx.cpu().long() for blk, idx in info["kept"].items()} if family == "dyvit" else None
    want = oracle.forward(params, x, cfg, precision="bf16", forced=forced)
    rel = ((logits.cpu() - want).norm() / want.norm()).item()
    print(f"\n[{family} tiny-width] relative L2 vs oracle_bf16 (teacher-forced where there are decisions): {rel:.3e}")
    assert rel < FORCED_TOL, rel


@pytest.mark.parametrize("name", ["topk_micro", "evit_micro", "dpcknn_micro", "kmedoids_micro", "dyvit_micro", "tome_micro", "ats_micro"])
def test_non_finite_input_gives_nan_logits_and_no_fault(name):
    """A NaN pixel makes every score of that image NaN.  The reference returns NaN logits for it (torch.topk / argsort order NaN as
    the largest value and still return valid indices); the decision kernels here must likewise stay inside their index ranges -- the
    index slabs feed gathers -- and leave the other images of the batch untouched."""
    case = GOLDEN_CASES[name]
    model, _, _ = build_model(case)
    x = make_images(case["batch"], 224, case["xseed"])
    clean, _ = model(x.cuda())
    bad = x.clone()
    bad[1, 0, 5, 7] = float("nan")
    logits, viz = model(bad.cuda())
    torch.cuda.synchronize()
    assert torch.isnan(logits[1]).all()
    assert torch.equal(logits[0], clean[0]) and torch.equal(logits[2:], clean[2:])
    for blk, kept in viz.get("Kept_Tokens", {}).items():
        kept = np.asarray(kept)
        assert kept.min() >= -1 and kept.max() <= 196, (blk, kept.min(), kept.max())      # -1: EViT's fused-token / padding marker


@pytest.mark.parametrize("kr,loc", [([0.8], [1, 2]), ([0.5], [0, 1, 2, 3, 4])])
def test_ats_sample_counts_whose_grid_has_k_points(kr, loc):
    """ats.py:48 builds the inverse-CDF grid with a float arange; for sample counts such as 126 (keep_rate 0.8, second stage) or 7 rounding
    lets the end point in: K grid points, up to K + 1 kept tokens.  The static bound must follow the grid (it used to raise)."""
    case = dict(family="ats", embed_dim=128, depth=max(loc) + 1, num_heads=2, num_classes=16, keep_rate=kr, reduction_loc=loc, batch=3,
                wseed=41, xseed=42, qkv_gain=4.0)
    cfg = case_config(case)
    counts, bounds = oracle.ats_sample_counts(cfg), oracle.ats_token_bounds(cfg)
    assert any(bounds[b] == counts[b] + 1 for b in counts), (counts, bounds)         # the case this test is about
    model, params, _ = build_model(case)
    model.precision = "fp32"
    x = make_images(3, 224, 42)
    logits, viz = model(x.cuda())
    assert [t for i, t in enumerate(model._last_tokens) if i in bounds] == [bounds[b] for b in sorted(bounds)]
    want, oviz = oracle.forward(params, x, cfg, precision="fp32", return_viz=True)
    for b in sorted(bounds):
        kept = viz["Kept_Tokens"][b]
        assert kept.shape[1] <= bounds[b] - 1 and kept.max() <= 195
    blk0 = sorted(bounds)[0]
    same = viz["Kept_Tokens"][blk0].shape == oviz["Kept_Tokens"][blk0].shape and (viz["Kept_Tokens"][blk0] == oviz["Kept_Tokens"][blk0]).all()
    d = (logits.cpu() - want).abs().max().item()
    print(f"\\nATS kr {kr} loc {loc}: bounds {bounds}; first-stage ids identical to the oracle: {bool(same)}; max|logit - oracle_fp32| = {d:.2e}")
    assert d < (2e-4 if same and len(loc) == 1 else 0.5)
    model.precision = "bf16"
    model.train()
    model.viz_mode = False
    torch.nn.functional.cross_entropy(model(x.cuda()), torch.randint(0, 16, (3,)).cuda()).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())
