"""Fine-tune bookkeeping: parameter groups against lists recorded from the reference's optim.py on the
reference's own models (tests/golden/param_groups.json), LR scaling, backbone freezing and the cosine-in-steps schedule."""
import json
import math
import os
import types

import pytest
import torch

from tokenreduction_amd import finetune

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "param_groups.json")

# key -> (factory name, kwargs of get_parameter_groups [+ freeze_patch_embed])
CASES = {
    "deit_tiny": ("deit_tiny_patch16_224_local", dict(learning_rate=1e-3, weight_decay=0.05)),
    "topk_tiny": ("topk_tiny_patch16_224", dict(learning_rate=2e-3, weight_decay=0.05, bone_lr_scale=0.1, fix_steps=3)),
    "dyvit_tiny_constant_tokens": ("dyvit_tiny_patch16_224", dict(learning_rate=1e-3, weight_decay=0.1, constant_cls=True, constant_pos=True)),
    "sit_tiny": ("sit_tiny_patch16_224", dict(learning_rate=1e-3, weight_decay=0.05, bone_lr_scale=0.01, fix_steps=5)),
    "dpcknn_tiny_skip": ("dpcknn_tiny_patch16_224", dict(learning_rate=5e-4, weight_decay=0.05, skip_list=("cls_token", "blocks.0.attn.qkv.weight"))),
    "sinkhorn_tiny_frozen_patch": ("sinkhorn_tiny_patch16_224", dict(learning_rate=1e-3, weight_decay=0.05, freeze_patch_embed=True)),
    "patchmerger_tiny": ("patchmerger_tiny_patch16_224", dict(learning_rate=1e-3, weight_decay=0.0)),
    "tome_tiny": ("tome_tiny_patch16_224", dict(learning_rate=1e-3, weight_decay=0.05)),
}

ORDER_FACTORIES = ["deit_tiny_patch16_224_local", "deit_tiny_patch16_224_local_viz", "topk_tiny_patch16_224", "evit_tiny_patch16_224",
                   "tome_tiny_patch16_224", "dyvit_tiny_patch16_224", "sit_tiny_patch16_224", "dpcknn_tiny_patch16_224",
                   "ats_tiny_patch16_224", "sinkhorn_tiny_patch16_224", "kmedoids_tiny_patch16_224", "patchmerger_tiny_patch16_224",
                   "heuristic_tiny_patch16_224"]


def case_args():
    return types.SimpleNamespace(keep_rate=[0.7], reduction_loc=[3, 6, 9], dyvit_distill=False, k_neighbors=5, equal_weight=False,
                                 cluster_iters=3, sinkhorn_eps=1.0, viz_mode=False, heuristic_pattern="l2", not_contiguous=False,
                                 min_radius=None)


@pytest.mark.parametrize("key", list(CASES))
def test_parameter_groups_match_the_reference(key):
    import tokenreduction_amd as tra
    factory, kw = CASES[key]
    kw = dict(kw)
    want = json.load(open(GOLDEN))[key]
    model = tra.create_model(factory, pretrained=False, num_classes=10, drop_rate=0.0, drop_path_rate=0.0, drop_block_rate=None,
                             img_size=224, args=case_args())
    if kw.pop("freeze_patch_embed", False):
        for p in model.patch_embed.parameters():
            p.requires_grad = False
    named = finetune.get_parameter_groups(model, with_names=True, **kw)
    assert [g["params"] for g in named] == [g["params"] for g in want.values()]          # membership AND order, group by group
    for g, w in zip(named, want.values()):
        assert g["lr"] == pytest.approx(w["lr"], rel=1e-12) and g["weight_decay"] == w["weight_decay"] and g["fix_step"] == w["fix_step"]
    groups = finetune.get_parameter_groups(model, **kw)
    lookup = dict(model.named_parameters())
    for g, n in zip(groups, named):
        assert all(p is lookup[name] for p, name in zip(g["params"], n["params"]))
    opt = torch.optim.AdamW(groups)                                                       # the dicts are what torch.optim takes
    assert [pg["fix_step"] for pg in opt.param_groups] == [w["fix_step"] for w in want.values()]


def test_lr_scaling_and_freezing():
    assert finetune.scale_lr(0.001, 1024, 1024) == 0.001
    assert finetune.scale_lr(0.001, 256, 1024) == pytest.approx(0.00025)
    groups = [dict(lr=0.1, fix_step=2), dict(lr=0.2, fix_step=0)]
    finetune.frozen_lr(groups, 1)
    assert [g["lr"] for g in groups] == [0, 0.2]
    groups[0]["lr"] = 0.1
    finetune.frozen_lr(groups, 2)
    assert [g["lr"] for g in groups] == [0.1, 0.2]


def test_cosine_schedule_in_steps_closed_form():
    args = types.SimpleNamespace(sched="cosine", sched_in_steps=True, epochs=10, warmup_epochs=2, num_steps_epoch=50, min_lr=1e-5,
                                 warmup_lr=1e-6, decay_rate=0.1)
    groups = [dict(lr=1e-3), dict(lr=1e-5)]
    sch = finetune.CosineSchedule.from_args(args, groups)
    assert [g["lr"] for g in groups] == [1e-6, 1e-6]                 # warm-up starts every group at warmup_lr
    sch.step(3)                                                      # epoch ticks are ignored when counting in steps
    assert [g["lr"] for g in groups] == [1e-6, 1e-6]
    sch.step_update(50)
    assert groups[0]["lr"] == pytest.approx(1e-6 + 50 * (1e-3 - 1e-6) / 100)
    sch.step_update(100)                                             # first cosine step: warm-up is NOT a prefix, t counts from 0
    assert groups[0]["lr"] == pytest.approx(1e-5 + 0.5 * (1e-3 - 1e-5) * (1 + math.cos(math.pi * 100 / 500)))
    sch.step_update(250)
    assert groups[0]["lr"] == pytest.approx(1e-5 + 0.5 * (1e-3 - 1e-5))
    assert groups[1]["lr"] == pytest.approx(1e-5)                    # base lr == lr_min: flat
    sch.step_update(499)
    assert 1e-5 < groups[0]["lr"] < 1.1e-5
    sch.step_update(500)                                             # past the single cycle (cycle_limit 1): lr_min
    assert groups[0]["lr"] == 1e-5
    lrs = [sch.lr_at(t)[0] for t in range(100, 500)]
    assert all(a > b for a, b in zip(lrs, lrs[1:]))                  # monotone decay after the warm-up


def test_cosine_schedule_in_epochs_and_restarts():
    groups = [dict(lr=1.0)]
    sch = finetune.CosineSchedule(groups, t_initial=4, lr_min=0.0, t_in_epochs=True, t_mul=2.0, decay_rate=0.5, cycle_limit=3)
    sch.step_update(7)
    assert groups[0]["lr"] == 1.0
    want = {0: 1.0, 2: 0.5, 4: 0.5, 8: 0.25, 12: 0.25, 28: 0.0}       # cycles of 4, 8, 16 epochs with peaks 1, 0.5, 0.25
    for t, v in want.items():
        sch.step(t)
        assert groups[0]["lr"] == pytest.approx(v, abs=1e-12), t


@pytest.mark.parametrize("factory", ORDER_FACTORIES)
def test_parameter_and_state_dict_order_match_the_reference(factory):
    """An optimizer state_dict addresses parameters by POSITION: a fine-tune checkpoint of the reference resumes on these models
    only if named_parameters() runs in the reference's order (its topk/evit/tome/dyvit/kmedoids classes rebuild `blocks` last)."""
    import tokenreduction_amd as tra
    want = json.load(open(GOLDEN))["__order__"][factory]
    model = tra.create_model(factory, pretrained=False, num_classes=10, drop_rate=0.0, drop_path_rate=0.0, drop_block_rate=None,
                             img_size=224, args=case_args())
    assert [n for n, _ in model.named_parameters()] == want["parameters"]
    assert list(model.state_dict().keys()) == want["state_dict"]
