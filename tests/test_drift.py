"""north_star's drift target as a test: "<= 0.1 % top-1 drift vs. reference" on TRAINED weights, and "logits within 1e-3 abs" for the
tolerance-conformant executor.  There are no ImageNet weights in this container and random-init logits are nearly flat, so the
BASELINE configs[1] model (DeiT-S Top-K) is fine-tuned here by the build's own HIP training path -- 600 AdamW steps on a separable
synthetic 1000-class task, about six seconds -- and 10,240 held-out images go through the three executors (tools/drift_trained.py, the
leg bench.py reports as `drift_trained`).  The fp32 executor stands for the reference: tests/test_hip_fp32.py pins it to the reference's
golden vectors (every index identical, logits <= 1e-5)."""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def drift_record():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    spec = importlib.util.spec_from_file_location("drift_trained", os.path.join(ROOT, "tools", "drift_trained.py"))
    dt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dt)
    rec = dt.run("cuda", steps=600, eval_images=10240)
    assert rec["fp32"]["top1_acc"] > 0.9, f"the synthetic task did not train: {rec['fp32']}"      # (0.97-0.99 in every recorded run)
    return rec


@pytest.mark.gpu
@pytest.mark.parametrize("schedule,min_agreement", [("keep_rate 0.7", 0.997), ("keep_rate 0.5", 0.995)])
def test_bf16_product_path_top1_drift_on_trained_weights(drift_record, schedule, min_agreement):
    """The timed bf16 path against the fp32 executor on trained weights: top-1 accuracy within 0.1 % (north_star's drift target) and the
    top-1 decision itself equal on >= 99.7 % of the images at the headline schedule (recorded: 99.92-99.98 %), >= 99.5 % at keep_rate
    0.5 (recorded: 99.74 %: fewer tokens, more boundary flips)."""
    rec = drift_record if schedule == "keep_rate 0.7" else drift_record["keep_rate_0.5"]
    bf, ref = rec["bf16"], rec["fp32"]
    assert abs(bf["top1_acc"] - ref["top1_acc"]) <= 1e-3, (schedule, bf["top1_acc"], ref["top1_acc"])
    assert bf["top1_agreement_with_fp32"] >= min_agreement, (schedule, bf)
    # and where the token decisions equal the reference's the logits are close: what remains is bf16 arithmetic, not a flipped token
    assert bf["max_abs_logit_diff_on_images_with_the_reference_token_sets"] < 0.1, (schedule, bf)


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["keep_rate 0.7", "keep_rate 0.5"])
def test_split_bf16_path_meets_1e_3_wherever_the_token_sets_are_the_references(drift_record, schedule):
    """precision="bf16x3" (north_star's tolerance on the matrix cores): no image whose kept-token sets equal the fp32 executor's misses
    1e-3 abs; every miss is an image where a boundary token flipped (a handful of 10,240); no top-1 decision differs."""
    rec = drift_record if schedule == "keep_rate 0.7" else drift_record["keep_rate_0.5"]
    x3 = rec["bf16x3"]
    assert x3["images_over_1e-3_with_the_reference_token_sets"] == 0, (schedule, x3)
    assert x3["max_abs_logit_diff_on_images_with_the_reference_token_sets"] < 1e-3, (schedule, x3)
    assert x3["images_over_1e-3"] == x3["images_over_1e-3_where_a_kept_set_differs"] <= 64, (schedule, x3)
    assert x3["top1_agreement_with_fp32"] >= 0.9995 and abs(x3["top1_acc"] - rec["fp32"]["top1_acc"]) <= 5e-4, (schedule, x3)
