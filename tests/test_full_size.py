"""GPU: every BASELINE.json config at the size bench.py times it, compared at that size.

The fixture-size tests pin the HIP path to the oracle on batches of 2-4 images.  What they cannot see is anything that only
happens at the timed size: GEMM row counts that are not a multiple of the tile (128 x 197 = 25,216 token rows), the tail round of the
persistent tile schedule, the 256 x 6 attention workgroups, clustering scratch at B x P x P.  Every kernel treats an image's tokens
independently of the batch around it and in a batch-independent summation order, so the comparison is exact:

    logits (and every decision) of 8 images spread over the full batch -- first, middle, LAST -- must equal, BIT FOR BIT, the same 8
    images run as a batch of 8,

and the batch of 8 is what tests/test_hip_model.py / test_hip_train.py compare with the oracle (configs[1] repeats that comparison
here on two of the images, so that the chain full size -> small batch -> oracle is closed in one place).  For the training step of
configs[3] (engine.py:50-76) the loss only looks at those 8 images (the other rows' d logits are zero, so every gradient
contribution is theirs): the parameter gradients must agree with the batch-of-8 run up to the summation order of the weight-gradient
reductions (measured 1e-6 .. 2e-5 relative L2; asserted < 2e-4)."""
import types

import pytest
import torch

import oracle
from tests._params import GOLDEN_CASES, case_config, make_params
from tests.test_hip_model import FORCED_TOL, build_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _spread(B):
    """first, second, around the middle (both sides of it), LAST: eight indices"""
    return [0, 1, B // 3, B // 2 - 1, B // 2, (2 * B) // 3, B - 2, B - 1]


def _images(B, size, seed):
    return torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(seed))


def _eval_pair(model, x, sel, **per_batch):
    """(full-batch outputs, the same images as a batch of 8): logits + viz_data"""
    model.viz_mode = True
    for k, v in per_batch.items():
        setattr(model, k, v["full"])
    lf, vf = model(x.cuda())
    lf, vf = lf.clone(), {k: {b: t.clone() if torch.is_tensor(t) else t for b, t in d.items()} for k, d in vf.items() if k != "Features"}
    for k, v in per_batch.items():
        setattr(model, k, v["sel"])
    ls, vs = model(x[sel].contiguous().cuda())
    return lf, vf, ls.clone(), vs


def _assert_same(lf, vf, ls, vs, sel, what):
    assert torch.isfinite(lf).all(), what
    assert torch.equal(lf[sel], ls), f"{what}: logits of images {sel} differ between the full batch and a batch of 8: max {float((lf[sel] - ls).abs().max())}"
    for key, d in vs.items():
        if key == "Features":
            continue
        for blk, t in d.items():
            full = vf[key][blk]
            t, full = torch.as_tensor(t), torch.as_tensor(full)
            assert torch.equal(full[sel], t), f"{what}: viz_data[{key}][{blk}] differs"


@pytest.mark.parametrize("name,kr,factory", [("topk_small_kr07", 0.7, "topk_small_patch16_224"), ("topk_small_kr05", 0.5, "topk_small_patch16_224"),
                                            ("evit_small_kr05", 0.5, "evit_small_patch16_224")])
def test_configs1_small_batch256(name, kr, factory):
    """configs[1]: DeiT-S Top-K keep_rate 0.7 at blocks 3/6/9, batch 256 (M = 50,432 / 35,328 / 24,832 / 17,408 token rows) -- the headline
    workload -- and north_star's own target line, keep_rate 0.5 (Top-K: 99 / 50 / 25 tokens per image, M = 25,344 / 12,800 / 6,400 after
    the stages, where every kernel of a block sits near its launch floor; EViT: one fused token more; models/topk.py:55-56, :141-150)."""
    import tokenreduction_amd as tra
    from tests._stepwise import forward_stepwise
    case = GOLDEN_CASES[name]
    args = types.SimpleNamespace(keep_rate=[kr], reduction_loc=[3, 6, 9], viz_mode=True)
    model = tra.create_model(factory, args=args)
    cfg = case_config(case)
    params = make_params(cfg, case["wseed"], case["qkv_gain"])
    model.load_state_dict(params)
    model = model.cuda().eval()
    B = 256
    x, sel = _images(B, 224, 0), _spread(B)
    _assert_same(*_eval_pair(model, x, sel), sel, f"{name} B=256")
    # ... and the batch of 8 against the oracle: decisions bit-exact on the device's own scores, logits teacher-forced
    xs = x[sel].contiguous()
    l2, info = forward_stepwise(model, xs.cuda())
    forced = {}
    for blk, idx in info["kept"].items():
        ref = oracle.cls_topk_select(info["scores"][blk].cpu(), idx.shape[1])
        assert torch.equal(idx.cpu().long(), ref), f"block {blk}: Top-K indices differ from the oracle on identical scores"
        forced[blk] = idx.cpu().long()[[0, 7]]
    want = oracle.vit_forward(params, xs[[0, 7]], cfg, precision="bf16", forced=forced)
    got = l2.cpu()[[0, 7]]
    err = float((got - want).norm() / want.norm())
    assert err < FORCED_TOL, f"images 0 and 255 of the full batch against the oracle with the HIP rounding points: {err}"


def test_configs2_tome_small_r16_batch256():
    """configs[2]: DeiT-S ToMe r = 16 in every block (bipartite matching + weighted merge 12 times), batch 256."""
    model, _, _ = build_model(GOLDEN_CASES["tome_small_r16"])
    B = 256
    x, sel = _images(B, 224, 1), _spread(B)
    _assert_same(*_eval_pair(model, x, sel), sel, "tome_small r16 B=256")


@pytest.mark.parametrize("name", ["ats_base_kr05", "dpcknn_base_kr05"])
def test_configs3_base_train_step_batch128(name):
    """configs[3]: DeiT-B ATS / DPC-KNN keep_rate 0.5, ONE training step at the per-GPU batch of 128 (25,216 token rows: not a multiple
    of any tile height): train-mode logits bit-identical to the batch of 8, parameter gradients equal up to summation order."""
    case = GOLDEN_CASES[name]
    B = 128
    x, sel = _images(B, 224, 2), _spread(B)
    w = torch.randn(len(sel), case["num_classes"], generator=torch.Generator().manual_seed(5)).cuda() / case["num_classes"]
    noise = None
    if case["family"] == "dpcknn":
        g = torch.Generator().manual_seed(6)
        noise = {blk: torch.rand(B, P, generator=g) for blk, P in ((3, 196), (6, 98), (9, 49))}

    def step(xb, rows, nz):
        model, _, _ = build_model(case)
        model.viz_mode = False
        model.train()
        if nz is not None:
            model.density_noise = nz
        out = model(xb.cuda())
        (out[rows] * w).sum().backward()
        flat = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).clone()
        return out.detach().clone(), flat

    full_logits, full_grad = step(x, sel, noise)
    sel_noise = None if noise is None else {blk: t[sel] for blk, t in noise.items()}
    sel_logits, sel_grad = step(x[sel].contiguous(), list(range(len(sel))), sel_noise)
    assert torch.isfinite(full_logits).all() and torch.isfinite(full_grad).all()
    assert torch.equal(full_logits[sel], sel_logits), f"{name}: train-mode logits differ: {float((full_logits[sel] - sel_logits).abs().max())}"
    assert full_grad.shape == sel_grad.shape and float(sel_grad.norm()) > 0
    rel = float((full_grad.double() - sel_grad.double()).norm() / sel_grad.double().norm())
    print(f"   {name}: gradient of the 8 selected images, batch 128 vs batch 8: relative L2 {rel:.2e}")
    assert rel < 2e-4, rel


@pytest.mark.parametrize("name", ["sinkhorn_base_384_kr025", "kmedoids_base_384_kr025"])
def test_configs4_base_384_batch64(name):
    """configs[4]: DeiT-B Sinkhorn / K-Medoids keep_rate 0.25 at 384 x 384 (577 tokens: the online-softmax attention, clustering over
    576 patches), batch 64."""
    model, _, _ = build_model(GOLDEN_CASES[name])
    B = 64
    x, sel = _images(B, 384, 3), _spread(B)
    _assert_same(*_eval_pair(model, x, sel), sel, f"{name} B=64")
