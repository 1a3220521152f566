"""GPU: the training path (tr_vit_forward_train + tr_vit_backward through model.train() / loss.backward()).

Gradient parity is a two-link chain, like the forward's:
  (1) tests/test_oracle_grad.py (CPU): torch.autograd over the oracle == the reference's recorded `loss.backward()` gradients;
  (2) here: HIP gradients vs torch.autograd over the oracle with the HIP rounding points (precision="bf16") and the DEVICE's own
      discrete decisions (kept ids / ToMe matches read back from the tape) -- the decisions themselves are pinned bit-exact at
      the op boundary by tests/test_hip_model.py.  The oracle's backward is fp32 arithmetic on the bf16-rounded forward values (rounding has
      an identity gradient); the HIP backward additionally rounds every gradient GEMM operand (dY, dS, P) to bf16 -- 2^-9 relative
      each, like torch autocast does -- so the error grows with depth: measured on MI355X whole-model relative L2 1.2-1.4e-2 at
      depth 4 and 2.9-5.0e-2 at depth 12 (EViT-S highest: its fused token adds a bf16 gradient path into the attention; worst
      single parameter 5.6e-2).  Asserted: 2e-2 (depth 4) / 6e-2 (depth 12) for the whole
      gradient vector, 2x that for the worst parameter.
  (3) decision-free models (DeiT) are also compared directly with the reference's recorded gradients.
"""
import os

import numpy as np
import pytest
import torch

from tests._params import GOLDEN_CASES, GRAD_CASES, grad_labels, grad_sample_index, make_images, oracle_param_grads
from tests.test_hip_model import build_model

pytestmark = pytest.mark.gpu

def grad_tol(case):
    return 2.5e-2 if case["depth"] <= 4 else 6e-2


GRAD_TOL = 5e-2


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _noise(case, golden_dir):
    """DPC-KNN: the reference's recorded density draws {blk: [B,P_in]} (gradient fixture), else None."""
    if case["family"] == "dyvit":
        name = [n for n, c in GOLDEN_CASES.items() if c is case][0]
        g = np.load(os.path.join(golden_dir, f"grad_{name}.npz"))
        return {n: torch.from_numpy(g[f"gumbel_{n}"]) for n in range(len(case["reduction_loc"]))}
    if case.get("drop_path"):
        from tests._params import drop_path_draws
        name = [n for n, c in GOLDEN_CASES.items() if c is case][0]
        g = np.load(os.path.join(golden_dir, f"grad_{name}.npz"))
        return drop_path_draws(case, [g[k] for k in sorted((k for k in g.files if k.startswith("rand_")), key=lambda k: int(k.split("_")[1]))])
    if case["family"] != "dpcknn":
        return None
    import oracle
    from tests._params import case_config
    name = [n for n, c in GOLDEN_CASES.items() if c is case][0]
    g = np.load(os.path.join(golden_dir, f"grad_{name}.npz"))
    return {blk: torch.from_numpy(g[f"rand_{n}"]) for n, blk in enumerate(sorted(oracle.dpcknn_cluster_counts(case_config(case))))}


def _dropout(case, golden_dir):
    """The reference's recorded nn.Dropout keep masks of a drop_rate case (gradient fixture), else None."""
    if not case.get("drop_rate"):
        return None
    from tests._params import dropout_masks
    name = [n for n, c in GOLDEN_CASES.items() if c is case][0]
    return dropout_masks(np.load(os.path.join(golden_dir, f"grad_{name}.npz")))


def _train_step(case, noise=None, dropout=None):
    from tokenreduction_amd import training
    model, params, cfg = build_model(case)
    model.viz_mode = False
    model.train()
    if dropout is not None:
        model.dropout_draws = dropout                       # the reference's keep masks, in its module call order
    if noise is not None and case.get("drop_path"):
        model.drop_path_draws = noise                       # the reference's DropPath draws [2*depth, B]
    elif noise is not None and case["family"] == "dyvit":
        model.gumbel_noise = noise                          # the reference's Gumbel draws, stage by stage
    elif noise is not None:
        model.density_noise = noise
    x = make_images(case["batch"], case.get("img_size", 224), case["xseed"]).cuda()
    out = model(x)
    if case["family"] == "dyvit":
        from tests._params import dyvit_train_loss
        logits = out[0]
        loss = dyvit_train_loss(out, grad_labels(case).cuda(), case)
    else:
        logits = out
        loss = torch.nn.functional.cross_entropy(logits, grad_labels(case).cuda())
    loss.backward()
    return model, logits.detach().cpu(), loss.item(), training.train_decisions(model)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("name", GRAD_CASES)
def test_gradients_match_the_oracle_on_the_device_decisions(golden_dir, name):
    case = GOLDEN_CASES[name]
    noise = _noise(case, golden_dir)
    dropout = _dropout(case, golden_dir)
    model, logits, loss, decisions = _train_step(case, noise, dropout)
    forced = {blk: (tuple(t.cpu() for t in d) if isinstance(d, tuple) else d.cpu()) for blk, d in decisions.items()}
    if case["family"] == "dyvit":          # the oracle indexes DyViT's stages 0..S-1
        forced = {j: forced[blk] for j, blk in enumerate(sorted(forced))}
    o_loss, o_logits, o_grads = oracle_param_grads(case, forced=forced or None, precision="bf16", noise=noise, dropout=dropout)
    rl = _rel(logits, o_logits)
    print(f"\n[{name}] loss {loss:.5f} (oracle {o_loss:.5f}); logits rel L2 {rl:.3e}")
    assert rl < 3e-2
    gnorm = float(torch.cat([o_grads[n].reshape(-1) for n, _ in model.named_parameters()]).double().norm())
    errs = []
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        # relative to the parameter's own gradient, floored at 1e-3 of the whole gradient's norm: some gradients are zero up to
        # cancellation (the key bias under softmax; DPC-KNN's score bias: sum_i w_i (x_i - x_c) = 0 within a cluster)
        d = float((p.grad.cpu().double() - o_grads[n].double()).norm())
        errs.append((d / max(float(o_grads[n].double().norm()), 1e-3 * gnorm), n))
    errs.sort(reverse=True)
    worst = errs[0]
    total = _rel(torch.cat([p.grad.reshape(-1).cpu() for _, p in model.named_parameters()]),
                 torch.cat([o_grads[n].reshape(-1) for n, _ in model.named_parameters()]))
    print(f"   gradients: whole-model rel L2 {total:.3e}; worst parameters {[(n, round(e, 4)) for e, n in errs[:6]]}")
    assert total < grad_tol(case), total
    assert worst[0] < 2 * grad_tol(case), worst


def test_deit_gradients_match_the_reference_fixture(golden_dir):
    """No discrete decision anywhere: HIP gradients straight against the reference's recorded loss.backward()."""
    g = np.load(os.path.join(golden_dir, "grad_deit_micro.npz"))
    model, logits, loss, _ = _train_step(GOLDEN_CASES["deit_micro"])
    assert abs(loss - float(g["loss"])) < 2e-2
    for n, p in model.named_parameters():
        flat = p.grad.reshape(-1).cpu()
        ref_norm = float(g["norm:" + n])
        assert abs(float(flat.double().norm()) - ref_norm) <= 2e-2 * ref_norm, n
        smp = flat[torch.from_numpy(grad_sample_index(flat.numel()))]
        ref = torch.from_numpy(g["sample:" + n])
        assert _rel(smp, ref) < 4e-2, (n, _rel(smp, ref))


def test_grad_accumulation_and_zero_grad():
    """Two backward passes accumulate (engine.py:41-84 grad accumulation); zero_grad(set_to_none=True) starts over; the
    gradients live in one flat buffer (p.grad are views) and a repeated step is bitwise reproducible."""
    case = GOLDEN_CASES["topk_micro"]
    model, *_ = _train_step(case)
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    st = model._train_state()
    assert all(p.grad.data_ptr() == st.views[n].data_ptr() for n, p in model.named_parameters())
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    loss = torch.nn.functional.cross_entropy(model(x), grad_labels(case).cuda())
    loss.backward()
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=1e-5, atol=1e-7), n
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model(x), grad_labels(case).cuda())
    loss.backward()
    for n, p in model.named_parameters():
        assert torch.equal(p.grad, g1[n]), n


@pytest.mark.parametrize("name", ["deit_micro", "topk_micro", "evit_micro", "tome_micro", "dpcknn_micro", "ats_micro", "kmedoids_micro", "heuristic_micro_l2",
                                  "dyvit_micro_train", "sinkhorn_micro_384", "ats_base_kr05"])
def test_a_fresh_backward_overwrites_every_gradient_slot(name):
    """With no gradient alive the backward passes accumulate = 0 and does NOT clear the flat gradient buffer (training.py): correctness then
    rests on tr_vit_backward writing every slot of every parameter exactly once per pass.  Enforced here: the buffer is poisoned
    with NaN before the backward of a fresh step; afterwards every parameter's view must be finite and equal to the gradients of an
    unpoisoned step, and nothing but the alignment padding between the slices may still hold the poison."""
    case = GOLDEN_CASES[name]
    model, *_ = _train_step(case)
    want = {n: p.grad.clone() for n, p in model.named_parameters()}
    st = model._train_state()
    model.zero_grad(set_to_none=True)
    x = make_images(case["batch"], case.get("img_size", 224), case["xseed"]).cuda()
    out = model(x)
    if case["family"] == "dyvit":
        from tests._params import dyvit_train_loss
        loss = dyvit_train_loss(out, grad_labels(case).cuda(), case)
    else:
        loss = torch.nn.functional.cross_entropy(out, grad_labels(case).cuda())
    st.flat.fill_(float("nan"))
    loss.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    assert not bad, f"gradient slots the backward did not write: {bad[:6]}"
    # what may stay poisoned: the alignment padding between the parameter slices, which no kernel reads (the optimizer works through its slot
    # table, the data-parallel mean reduces it along but nothing consumes it) -- at most as many elements as lie outside every view (some families keep derived gradients there)
    covered = sum(v.numel() for v in st.views.values())
    assert int((~torch.isfinite(st.flat)).sum()) <= st.flat.numel() - covered, "an element inside a parameter view was not written"
    if case["family"] not in ("dyvit", "dpcknn"):          # (these two draw fresh noise per training forward: Gumbel / density tie-break)
        for n, p in model.named_parameters():
            assert torch.equal(p.grad, want[n]), n


def test_backward_of_an_overwritten_tape_raises():
    """The tape belongs to the model, not to the autograd node: after a second train-mode forward the first one's activations are
    gone, and its backward must say so instead of writing gradients of the wrong forward (round-2 advisor finding)."""
    case = GOLDEN_CASES["topk_micro"]
    model, *_ = _train_step(case)
    model.zero_grad(set_to_none=True)
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    y = grad_labels(case).cuda()
    first = torch.nn.functional.cross_entropy(model(x), y)
    second = torch.nn.functional.cross_entropy(model(x + 0.5), y)
    with pytest.raises(RuntimeError, match="overwritten"):
        first.backward()
    second.backward()                                    # the latest forward's backward is the valid one
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    # an eval / no-grad forward in between does not touch the tape
    third = torch.nn.functional.cross_entropy(model(x), y)
    with torch.no_grad():
        model.eval()(x)
    model.train()
    third.backward()


def test_ema_copy_after_a_captured_eval_forward():
    """harness.ModelEma deep-copies the model; a model that has run an eval forward owns captured hipGraphs and a GPU workspace, which
    must not be (and cannot be) copied: the copy starts with empty executor caches and builds its own (round-2 advisor finding)."""
    import copy
    from tokenreduction_amd import harness
    case = GOLDEN_CASES["topk_micro"]
    model, params, cfg = build_model(case)
    model.viz_mode = False
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    want = model.eval()(x).clone()
    model.train()(x).sum().backward()                    # ... and a tape + flat gradient buffer
    assert model._ws and model._packed is not None and model._tstate is not None
    ema = harness.ModelEma(model, 0.99)
    assert ema.module._ws == {} and ema.module._packed is None and ema.module._tstate is None
    assert model._ws and model._packed is not None       # the original keeps its caches
    assert torch.equal(ema.module(x), want)              # same weights, its own workspace and graph
    twin = copy.deepcopy(model)
    assert twin._packed is None and all(a.data_ptr() != b.data_ptr() for a, b in zip(twin.parameters(), model.parameters()))


def test_optimizer_step_reduces_the_loss():
    """A few AdamW steps on one batch through the whole HIP training path: the loss falls (weights are re-packed each step)."""
    case = GOLDEN_CASES["evit_micro"]
    model, *_ = _train_step(case)
    model.zero_grad(set_to_none=True)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    y = grad_labels(case).cuda()
    losses = []
    for _ in range(6):
        loss = torch.nn.functional.cross_entropy(model(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("\nlosses", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0] - 0.3


def test_eval_and_train_logits_agree():
    case = GOLDEN_CASES["topk_micro"]
    model, params, cfg = build_model(case)
    model.viz_mode = False
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    le = model.eval()(x).clone()
    lt = model.train()(x).detach()
    # same kernels except GELU as a separate pass on the bf16 pre-activation (one extra rounding of fc1's output, which can
    # also move a token across a Top-K boundary): measured 1.2e-2
    assert _rel(lt.cpu(), le.cpu()) < 3e-2


def test_dropout_draws_its_own_masks_and_every_family_takes_it():
    """Without recorded masks the keep mask is a fresh Bernoulli(1 - p) draw on the device: two steps differ, every gradient is finite,
    eval ignores drop_rate; DyViT's policy-attention blocks go through the same dropout sites; attn_drop_rate still refuses."""
    for name in ("topk_micro", "tome_micro", "ats_micro", "dyvit_micro_train", "sit_micro"):
        case = dict(GOLDEN_CASES[name], drop_rate=0.2)
        model, params, cfg = build_model(case)
        model.viz_mode = False
        x = make_images(case["batch"], 224, case["xseed"]).cuda()
        y = grad_labels(case).cuda()
        want_eval = model.eval()(x).clone()
        model.train()
        losses = []
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            out = model(x)
            loss = torch.nn.functional.cross_entropy(out[0] if isinstance(out, tuple) else out, y)
            loss.backward()
            losses.append(loss.item())
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.named_parameters() if "score_predictor" not in n), name
        assert losses[0] != losses[1], (name, losses)                 # different masks
        assert torch.equal(model.eval()(x), want_eval), name          # dropout is a training-time thing
    model.attn_drop_rate = 0.1
    with pytest.raises(NotImplementedError, match="attn_drop_rate"):
        model.train()(x)


def test_unsupported_configuration_raises_in_train_mode():
    """What the training path does not cover raises up front, never a fallback: more than 640 tokens (448 x 448 inputs: 785)."""
    case = dict(GOLDEN_CASES["topk_micro"], img_size=448)
    model, params, cfg = build_model(case)
    x = make_images(1, 448, case["xseed"]).cuda()
    with pytest.raises(NotImplementedError, match="no training path"):
        model.train()(x)


@pytest.mark.parametrize("name", ["evit_micro", "ats_micro", "tome_micro", "kmedoids_micro", "sit_micro", "dpcknn_micro", "deit_micro"])
def test_gradient_reducer_on_rccl_world1(name):
    """The data-parallel path on RCCL with one rank (the GPU box has one GPU): the backward runs in per-bucket block ranges, each
    bucket is reduce-scattered + all-gathered in place on the side stream; with world size 1 the mean is the identity, so the
    gradients must equal the plain single-call backward bit for bit -- for every way a block moves the stream-gradient buffers
    (gather / merge swaps, the spare rotation of norm2's backward, ATS's scatter, K-Medoids' scatter-add, none): a range call has to
    find the buffers where the previous range left them."""
    import torch.distributed as dist
    from tokenreduction_amd.dp import FlatGradReducer
    case = GOLDEN_CASES[name]
    noise = None
    if case["family"] == "dpcknn":          # the density noise is drawn per forward unless given: the same (zero) draws for every forward here
        probe, *_ = build_model(case)
        noise = {blk: torch.zeros(case["batch"], P) for blk, _, P in probe._stage_shapes()}
    model, *_ = _train_step(case, noise)
    want = {n: p.grad.clone() for n, p in model.named_parameters()}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        red = FlatGradReducer(bucket_bytes=512 * 1024).attach(model)
        assert red.algorithm == "rs_ag"
        red.broadcast_parameters(model)
        x = make_images(case["batch"], 224, case["xseed"]).cuda()
        for rep in range(2):
            model.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(model(x), grad_labels(case).cuda()).backward()
            torch.cuda.synchronize()
            assert len(red.launched) >= 3 and red.launched[-1][1] == model._train_state().flat.numel()
            for n, p in model.named_parameters():
                assert torch.equal(p.grad, want[n]), (rep, n)
        with red.no_sync():                                   # accumulation micro-step: no collective, gradients add up
            red.launched = []
            torch.nn.functional.cross_entropy(model(x), grad_labels(case).cuda()).backward()
            assert red.launched == []
        for n, p in model.named_parameters():
            assert torch.allclose(p.grad, 2 * want[n], rtol=1e-5, atol=1e-7), n
    finally:
        model._grad_reducer = None
        dist.destroy_process_group()


def test_dyvit_train_return_contract():
    """dyvit.py:257-261: training returns (logits, out_pred_prob) or, with the DyViT distillation scheme, (logits, features [B,P,D],
    prev_decision.detach() [B,P,1], out_pred_prob: one [B,P] 0/1 tensor per stage, non-increasing: a dropped token stays dropped)."""
    from tests._params import dyvit_train_loss
    for distill in (False, True):
        case = dict(GOLDEN_CASES["dyvit_micro_train"], dyvit_distill=distill)
        model, params, cfg = build_model(case)
        model.viz_mode = False
        x = make_images(case["batch"], 224, case["xseed"]).cuda()
        out = model.train()(x)
        B, P, D = case["batch"], 196, case["embed_dim"]
        assert isinstance(out, tuple) and len(out) == (4 if distill else 2)
        preds = out[-1]
        assert len(preds) == 3 and all(p.shape == (B, P) and set(p.unique().tolist()) <= {0.0, 1.0} for p in preds)
        assert all(bool((b <= a).all()) for a, b in zip(preds, preds[1:]))
        assert out[0].shape == (B, case["num_classes"]) and out[0].requires_grad
        if distill:
            assert out[1].shape == (B, P, D) and out[2].shape == (B, P, 1) and not out[2].requires_grad
            assert torch.equal(out[2].squeeze(-1), preds[-1].detach())
            dyvit_train_loss(out, grad_labels(case).cuda(), case).backward()
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        # eval mode still prunes for real (dyvit.py:230-238)
        le = model.eval()(x)
        assert le.shape == (B, case["num_classes"]) and model._last_tokens[-1] == int(196 * 0.7 ** 3) + 1


@pytest.mark.parametrize("name", ["topk_micro", "dpcknn_micro", "sit_micro"])
def test_train_one_epoch_drives_the_hip_model(name):
    """engine.py:14-114 through harness.train_one_epoch on the real thing: gradient accumulation over two micro-steps (the second one
    ADDS into the flat buffer), norm clipping, EMA, a frozen parameter group, and the loss goes down over two epochs of one repeated
    batch; every p.grad is still a view into the flat gradient buffer afterwards."""
    from tokenreduction_amd import finetune, harness
    case = GOLDEN_CASES[name]
    model, params, cfg = build_model(case)
    model.viz_mode = False
    x = make_images(4 * case["batch"], 224, case["xseed"])
    y = torch.cat([grad_labels(dict(case, xseed=case["xseed"] + i)) for i in range(4)])
    loader = [(x[i * case["batch"]:(i + 1) * case["batch"]], y[i * case["batch"]:(i + 1) * case["batch"]]) for i in range(4)]
    groups = finetune.get_parameter_groups(model, 2e-3, 0.05, 1.0, 0)
    opt = torch.optim.AdamW(groups, lr=2e-3)
    ema = harness.ModelEma(model, 0.9)
    crit = harness.plain_criterion(torch.nn.functional.cross_entropy)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses = []
    for epoch in range(3):
        stats, total = harness.train_one_epoch(model, crit, loader, opt, "cuda", epoch, max_norm=5.0, model_ema=ema, grad_accum_steps=2,
                                               num_steps_epoch=2)
        assert np.isfinite(stats["loss"])
        losses.append(stats["loss"])
    assert total == 2 * 2 + 2                 # epoch * num_steps_epoch + optimizer steps of the last epoch (engine.py:113)
    assert losses[-1] < losses[0], losses
    moved = [n for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert len(moved) >= len(before) - 2, set(before) - set(moved)          # (the key bias may stay: its gradient is zero)
    st = model._train_state()
    lo, hi = st.flat.data_ptr(), st.flat.data_ptr() + st.flat.numel() * 4
    torch.nn.functional.cross_entropy(model(loader[0][0].cuda()), loader[0][1].cuda()).backward()
    assert all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
    # the EMA copy follows the parameters without being them
    ema_p = dict(ema.module.named_parameters())
    assert any(not torch.equal(ema_p[n].detach(), p.detach()) for n, p in model.named_parameters())


def test_every_factory_name_takes_a_training_step():
    """All 42 factory names of models_act.py:8-51 through one fwd + loss + bwd (tools/all_models_train_smoke.py): finite loss, a finite
    gradient on every parameter; nothing raises (no whitelist: DyViT / SiT at DeiT-T width train with their hidden layers zero-padded);
    the DyViT teachers run their inference executor whatever the module's mode."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "all_models_train_smoke.py")], capture_output=True, text=True, timeout=900,
                         cwd=root)
    assert out.returncode == 0 and out.stdout.strip().endswith("ALL OK"), out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok ") >= 42 and "raises" not in out.stdout


@pytest.mark.parametrize("name", ["topk_small_patch16_224", "dyvit_small_patch16_224", "dpcknn_small_patch16_224", "tome_small_patch16_224",
                                  "ats_small_patch16_224", "sit_small_patch16_224"])
def test_training_gradients_are_the_same_bits_run_after_run(name):
    """Race screen of the training path (tools/lab/train_soak.py is the long form, 12 families x 60 steps at full width): the same forward +
    loss + backward -- same weights, same batch, same noise seed, no optimizer step -- eight times back to back; every gradient equals the
    first run's bit for bit.  The backward kernels sum in a fixed order (since round 6 also DyViT's predictor head, whose weight gradient was
    accumulated with LDS float atomics), so a difference is a race or an uninitialised read."""
    import types
    import tokenreduction_amd as tra
    tome = name.startswith("tome")
    args = types.SimpleNamespace(keep_rate=[196 - 16 * (i + 1) for i in range(12)] if tome else [0.7], reduction_loc=list(range(12)) if tome else [3, 6, 9],
                                 dyvit_distill=False, k_neighbors=5, equal_weight=False, cluster_iters=3, sinkhorn_eps=1.0, heuristic_pattern="l2",
                                 not_contiguous=False, min_radius=None)
    torch.manual_seed(0)
    model = tra.create_model(name, pretrained=False, num_classes=100, drop_rate=0.0, drop_path_rate=0.0, drop_block_rate=None, img_size=224,
                             args=args).cuda().train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(24, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 100, (24,), generator=g).cuda()
    ref, bad = None, {}
    for i in range(8):
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        out = model(x)
        loss = torch.nn.functional.cross_entropy(out[0] if isinstance(out, tuple) else out, y)
        for p_ in model.parameters():
            p_.grad = None
        loss.backward()
        grads = {k: p_.grad.clone() for k, p_ in model.named_parameters() if p_.grad is not None}
        if ref is None:
            ref = grads
            assert len(ref) > 100 and all(bool(torch.isfinite(v).all()) for v in ref.values())
            continue
        for k, v in grads.items():
            if not torch.equal(v, ref[k]):
                bad[k] = bad.get(k, 0) + 1
    assert not bad, f"gradients that differed from the first run: {sorted(bad.items())[:6]}"


@pytest.mark.parametrize("classes", [555, 81, 10, 1])
def test_any_number_of_classes_trains_and_evaluates(classes):
    """train.py:334 `model.reset_classifier(args.num_classes)`: NABirds has 555 classes, NUS-WIDE 81.  The kernels see the classifier padded to a
    multiple of 8 rows; logits, loss and gradients must be those of the unpadded head (against torch on the CLS features)."""
    case = GOLDEN_CASES["topk_micro"]
    model, params, cfg = build_model(case)
    model.viz_mode = False
    torch.manual_seed(0)
    model.reset_classifier(classes)
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    y = torch.randint(0, classes, (case["batch"],)).cuda()
    model.eval()
    le = model(x)
    assert le.shape == (case["batch"], classes)
    model.train()
    lt = model(x)
    assert lt.shape == (case["batch"], classes) and _rel(lt.detach().cpu(), le.cpu()) < 1e-1      # eval / train kernels differ by roundings only
    torch.nn.functional.cross_entropy(lt, y).backward()
    assert model.head.weight.grad.shape == (classes, case["embed_dim"]) and model.head.bias.grad.shape == (classes,)
    # d bias of cross-entropy = mean over the batch of (softmax - onehot): independent of everything upstream
    want_db = (torch.softmax(lt.detach().float(), -1) - torch.nn.functional.one_hot(y, classes).float()).mean(0)
    torch.testing.assert_close(model.head.bias.grad, want_db, atol=2e-3, rtol=2e-2)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())


@pytest.mark.parametrize("name", ["topk_micro", "dpcknn_micro", "topk_small_kr07"])
def test_one_launch_adamw_is_bit_identical_to_torch_fused_adamw(name):
    """tokenreduction_amd.optim.FusedAdamW (AdamW + bf16 / transposed operand refresh [+ gradient zeroing] in one launch, csrc/tr_optim.hip)
    against torch.optim.AdamW(fused=True) -- what the fine-tune leg used before: same start, same batches, two parameter groups with
    different learning rates and weight decay, a schedule that changes lr every step.  After every one of 10 steps all parameters,
    exp_avg and exp_avg_sq must be BIT-identical, and so must the training logits (the refreshed operand copies are the parameters,
    rounded, whoever rounds them).  One case runs the optimizer with zero_grads=True and keeps the gradient views alive
    (zero_grad(set_to_none=False)): the backward then ACCUMULATES into what the step zeroed instead of overwriting."""
    from tokenreduction_amd.optim import FusedAdamW
    keep_views = name == "dpcknn_micro"
    case = GOLDEN_CASES[name]
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    y = grad_labels(case).cuda()

    def groups(model):
        decay = [p for n, p in model.named_parameters() if p.dim() >= 2]
        rest = [p for n, p in model.named_parameters() if p.dim() < 2]
        return [dict(params=decay, weight_decay=0.05), dict(params=rest, weight_decay=0.0, lr=3e-3)]

    runs = {}
    for kind in ("torch", "hip"):
        model, _, _ = build_model(case)
        model.viz_mode = False
        model.train()
        if case["family"] == "dpcknn":
            model.density_noise = None
            torch.manual_seed(7)          # the device-side noise draws: same seed, same draws in both runs
        opt = (torch.optim.AdamW(groups(model), lr=2e-3, betas=(0.9, 0.98), eps=1e-8, fused=True) if kind == "torch"
               else FusedAdamW(groups(model), lr=2e-3, betas=(0.9, 0.98), eps=1e-8, model=model, zero_grads=keep_views))
        base = [g["lr"] for g in opt.param_groups]
        trace = []
        for it in range(10):
            for g, b in zip(opt.param_groups, base):
                g["lr"] = b * (1.0 - 0.07 * it)
            out = model(x)
            loss = torch.nn.functional.cross_entropy(out, y)
            opt.zero_grad(set_to_none=not (keep_views and kind == "hip"))
            loss.backward()
            opt.step()
            if keep_views and kind == "hip":
                assert all(not bool(p.grad.any()) for p in model.parameters())
            trace.append((out.detach().clone(), [p.detach().clone() for p in model.parameters()]))
        states = [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for p in model.parameters()]
        runs[kind] = (trace, states, model)
    for it, ((lt, pt), (lh, ph)) in enumerate(zip(runs["torch"][0], runs["hip"][0])):
        assert torch.equal(lt, lh), f"step {it}: training logits differ ({float((lt - lh).abs().max())})"
        for (n, _), a, b in zip(runs["torch"][2].named_parameters(), pt, ph):
            assert torch.equal(a, b), f"step {it}: parameter {n} differs by {float((a - b).abs().max())}"
    for (n, _), (m0, v0), (m1, v1) in zip(runs["torch"][2].named_parameters(), runs["torch"][1], runs["hip"][1]):
        assert torch.equal(m0, m1) and torch.equal(v0, v1), f"optimizer state of {n} differs"
    # eval after training: the operands the step refreshed are what a repack would have produced
    e_t, e_h = runs["torch"][2].eval()(x), runs["hip"][2].eval()(x)
    assert torch.equal(e_t, e_h)


def test_one_launch_adamw_resumes_from_a_state_dict():
    """train.py:373-383 resumes optimizers with load_state_dict: FusedAdamW carries torch's state layout (step, exp_avg, exp_avg_sq per
    parameter), so three steps, a state_dict round trip into a NEW optimizer (tensor-valued steps included) and three more steps equal six
    uninterrupted ones bit for bit -- and the state_dict loads into torch.optim.AdamW as well."""
    from tokenreduction_amd.optim import FusedAdamW
    case = GOLDEN_CASES["topk_micro"]
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    y = grad_labels(case).cuda()

    def steps(model, opt, n):
        for _ in range(n):
            loss = torch.nn.functional.cross_entropy(model(x), y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()

    finals = []
    for resume in (False, True):
        model, _, _ = build_model(case)
        model.viz_mode = False
        model.train()
        opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.05, model=model)
        steps(model, opt, 3)
        if resume:
            sd = opt.state_dict()
            for st in sd["state"].values():
                st["step"] = torch.tensor(float(st["step"]))                # what torch's fused AdamW checkpoints hold
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
            torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05).load_state_dict(sd)     # same layout as torch's
            opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.05, model=model)
            opt.load_state_dict(sd)
        steps(model, opt, 3)
        finals.append([p.detach().clone() for p in model.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*finals))


def test_fused_optimizer_steps_reach_the_executor():
    """torch's fused optimizers update parameters without bumping `_version`; the executor's operand copies must follow anyway (round 3:
    they did not -- AdamW(fused=True) trained on the initial bf16 matrices).  Fused and unfused AdamW from the same start must walk the
    same loss curve, the eval logits must move with the parameters, and workspaces / captured graphs survive the in-place repack."""
    case = GOLDEN_CASES["topk_micro"]
    x = make_images(case["batch"], 224, case["xseed"]).cuda()
    y = grad_labels(case).cuda()
    curves, finals = {}, {}
    for fused in (False, True):
        model, params, cfg = build_model(case)
        model.viz_mode = False
        e0 = model.eval()(x).clone()
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05, fused=fused)
        losses, ws_id = [], None
        for it in range(5):
            loss = torch.nn.functional.cross_entropy(model(x), y)
            if it == 0:           # (the first training forward adds the transposed copies: one reallocation)
                ws_id = id(next(iter(model._ws.values())))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        curves[fused] = losses
        e1 = model.eval()(x).clone()
        finals[fused] = e1
        assert not torch.equal(e0, e1), "eval logits did not move with the optimizer steps"
        assert id(next(iter(model._ws.values()))) == ws_id, "the workspace (and its captured graph) was thrown away by a repack"
        # the packed copies ARE the parameters, rounded
        pk = model._pack()
        w = model.blocks[1].mlp.fc1.weight.detach()
        slot = [t for (i, kind), t in model._pack_slots.items() if kind == "w" and t.shape == w.shape]
        assert any(torch.equal(t, w.to(torch.bfloat16)) for t in slot)
        tq = pk["tblocks"][1][2]
        assert torch.equal(tq, w.t().to(torch.bfloat16).contiguous())
    print("\nunfused", [round(v, 4) for v in curves[False]], "\nfused  ", [round(v, 4) for v in curves[True]])
    assert curves[True][-1] < curves[True][0] - 0.5
    assert all(abs(a - b) < 5e-2 * max(1.0, abs(a)) for a, b in zip(curves[False], curves[True])), curves
