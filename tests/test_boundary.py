"""CPU-side checks of the drop-in boundary: the registry surface, state-dict keys, the C-ABI symbols,
and that the product path fails loudly instead of falling back."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

import tokenreduction_amd as tra
from tokenreduction_amd import _lib
from tests._params import make_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# models_act.py:8-51 __all__
REFERENCE_NAMES = [f"{fam}_{size}_patch16_224{suf}"
                   for size in ("tiny", "small", "base")
                   for fam, suf in (("deit", "_local"), ("deit", "_local_viz"), ("dpcknn", ""), ("dyvit", ""),
                                    ("dyvit", "_teacher"), ("kmedoids", ""), ("patchmerger", ""), ("sinkhorn", ""),
                                    ("ats", ""), ("heuristic", ""), ("topk", ""), ("evit", ""), ("tome", ""), ("sit", ""))]


def _args(**kw):
    return types.SimpleNamespace(**kw)


def test_registry_has_every_reference_name():
    assert len(REFERENCE_NAMES) == 42
    for n in REFERENCE_NAMES:
        assert tra.is_model(n), n


def test_every_reference_factory_name_constructs():
    """All 42 names of models_act.py:8-51 build a module of this package (construction is host-side; compute needs the GPU)."""
    args = _args(keep_rate=[0.7], reduction_loc=[3, 6, 9], dyvit_distill=False, k_neighbors=5, equal_weight=False,
                 cluster_iters=3, sinkhorn_eps=1.0, heuristic_pattern="l2", not_contiguous=False, min_radius=None)
    for n in REFERENCE_NAMES:
        if "_tiny_" not in n:          # one size is enough for the smoke; tiny = 5.7 M params keeps this fast
            continue
        m = tra.create_model(n, pretrained=False, num_classes=10, img_size=224, args=args)
        assert isinstance(m, torch.nn.Module) and m.embed_dim == 192, n
    m = tra.create_model("kmedoids_tiny_patch16_224", args=_args(keep_rate=[0.7], reduction_loc=[3, 6, 9], equal_weight=True,
                                                                cluster_iters=3))       # the numpy-RNG branch (kmedoids.py:43-58)
    assert m.equal_weight


@pytest.mark.parametrize("name,dims", [("topk_tiny_patch16_224", (192, 3)), ("evit_small_patch16_224", (384, 6)),
                                       ("deit_small_patch16_224_local", (384, 6))])
def test_factory_dims_and_state_dict_keys(name, dims):
    m = tra.create_model(name, pretrained=False, num_classes=1000, drop_rate=0.0, drop_path_rate=0.1,
                         drop_block_rate=None, img_size=224, args=_args(keep_rate=[0.7], reduction_loc=[3, 6, 9]))
    assert (m.embed_dim, m.num_heads, m.depth) == (dims[0], dims[1], 12)
    cfg = types.SimpleNamespace(embed_dim=dims[0], depth=12, num_heads=dims[1], mlp_ratio=4, num_classes=1000,
                                img_size=224, patch_size=16, in_chans=3)
    ref_keys = set(make_params(cfg, 0).keys())          # the reference's key names (checked against it in gen_golden)
    assert set(m.state_dict().keys()) == ref_keys
    m.load_state_dict(make_params(cfg, 0), strict=True)
    assert m.patch_embed.num_patches == 196 and m.pos_embed.shape == (1, 197, dims[0])
    assert m.no_weight_decay() == {'pos_embed', 'cls_token', 'dist_token'}
    assert m.get_new_module_names() == []
    m.reset_classifier(10)
    assert m.head.out_features == 10


def test_keep_schedule_matches_reference_rule():
    m = tra.create_model("topk_small_patch16_224", args=_args(keep_rate=[0.7], reduction_loc=[3, 6, 9]))
    assert [m._keep[i] for i in (3, 6, 9)] == [137, 96, 67] and m.get_reduction_count() == [3, 6, 9]
    assert m.token_ratio == pytest.approx([0.7, 0.49, 0.343])
    m = tra.create_model("evit_small_patch16_224", args=_args(keep_rate=[0.9, 0.8, 0.5], reduction_loc=[1, 2, 3]))
    assert [m._keep[i] for i in (1, 2, 3)] == [176, 156, 98]
    with pytest.raises(AssertionError):
        tra.create_model("topk_small_patch16_224", args=_args(keep_rate=[0.7, 0.5], reduction_loc=[3, 6, 9]))


def test_create_model_drops_none_kwargs():
    m = tra.create_model("deit_tiny_patch16_224_local", pretrained=False, num_classes=5, drop_block_rate=None,
                         args=_args())
    assert m.num_classes == 5


def test_no_cpu_path_in_eval_or_train_mode():
    m = tra.create_model("topk_tiny_patch16_224", args=_args(keep_rate=[0.7], reduction_loc=[3, 6, 9]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.train()(torch.zeros(1, 3, 224, 224))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.eval()(torch.zeros(1, 3, 224, 224))
    # the reference's default --drop-path 0.1 (train.py:48) has no HIP path yet: loud, not silently ignored
    m = tra.create_model("topk_tiny_patch16_224", drop_path_rate=0.1, args=_args(keep_rate=[0.7], reduction_loc=[3, 6, 9]))
    with pytest.raises((NotImplementedError, RuntimeError)):
        m.train()(torch.zeros(1, 3, 224, 224))


def test_pretrained_offline_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    with pytest.raises(RuntimeError, match="deit_weights"):
        tra.create_model("topk_small_patch16_224", pretrained=True, args=_args(keep_rate=[0.7], reduction_loc=[3, 6, 9]))


def test_library_exports_every_declared_symbol():
    """Every function declared in include/tokenreduction_hip.h is exported by the built .so and bound in _lib."""
    hdr = open(os.path.join(ROOT, "include", "tokenreduction_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                                   # raises if the .so is not built
    for name in declared:
        assert hasattr(lib, name)
    assert lib.tr_version() >= 100


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.TrBlockWeights) == 13 * 8          # 12 parameter pointers + mlp_pk
    assert ctypes.sizeof(_lib.TrStageWeights) == 10 * 8 + 16
    assert ctypes.sizeof(_lib.TrVitWeights) == 8 * 8 + 32 * 13 * 8 + 32 * 96
    assert ctypes.sizeof(_lib.TrVitConfig) == (9 + 1 + 32 + 1 + 1 + 2 + 32 + 1 + 1) * 4      # ... + ats_dynamic + concurrent


def test_argument_validation_without_gpu():
    """Shape/null checks run on the host before any launch, so they are testable without a GPU."""
    lib = _lib.load()
    assert lib.tr_gemm_bf16(None, None, None, None, None, 0, 1, 1, 1, 0, None) == -3
    assert b"null" in lib.tr_last_error()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 255) // 256 * 256
    assert lib.tr_gemm_bf16(p, p, p, p, None, 0, 8, 8, 48, 0, None) == -1          # K % 64
    assert lib.tr_attention_bf16(p, p, None, p, p, 1, 700, 1, None) == -1            # column sums WITH a key bias: N > 608
    assert lib.tr_tome_match(p, 0, p, p, p, 1, 197, 6, 120, None) == -1              # r > (N-1)//2
    assert lib.tr_cls_topk(p, p, None, None, 1, 1, 10, 10, None) == -1               # K > P
    cfg = _lib.TrVitConfig()
    assert lib.tr_vit_workspace_bytes(ctypes.byref(cfg), 4) == 0
    cfg.family, cfg.img_size, cfg.patch, cfg.in_chans = 1, 224, 16, 3
    cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.mlp_hidden, cfg.num_classes = 384, 12, 6, 1536, 1000
    n = lib.tr_vit_workspace_bytes(ctypes.byref(cfg), 256)
    assert 0 < n < 2 ** 31
    # round-3 entry points: host-side checks only
    assert lib.tr_patch_embed_supported(3, 224, 16, 384) == 1 and lib.tr_patch_embed_supported(3, 384, 16, 768) == 1
    assert lib.tr_patch_embed_supported(3, 224, 16, 192) == 0 and lib.tr_patch_embed_supported(3, 224, 32, 384) == 0      # DeiT-T width, patch 32
    assert lib.tr_patch_embed_bf16(p, p, p, p, p, p, 1, 3, 224, 16, 128, None) == -1                                         # embed_dim % 384
    assert lib.tr_layernorm2_bf16(p, 384, None, 0, None, 384, None, 0, p, p, p, 4, 384, 1e-6, None) == -3                    # needs a residual
    assert lib.tr_attention_policy_bwd_long_bf16(p, p, p, p, p, p, 8, 1, 300, 1, None) == -1                                 # workspace too small
    # dropout keep mask: one byte per element the training forward drops -- pos_drop, then per block proj's rows, the hidden layer, fc2's rows
    # (Top-K reduces inside a block, after attention + proj: proj sees the tokens that entered the block, the Mlp those that are left)
    n_mlp = [197] * 3 + [138] * 3 + [97] * 3 + [68] * 3
    n_att = [197] * 4 + [138] * 3 + [97] * 3 + [68] * 2
    for i, k in ((3, 137), (6, 96), (9, 67)):
        cfg.keep[i] = k
    want = 4 * 197 * 384 + sum(4 * (a * 384 + t * (1536 + 384)) for a, t in zip(n_att, n_mlp))
    assert lib.tr_vit_dropout_mask_bytes(ctypes.byref(cfg), 4) == want
    # round-4 entry points
    assert lib.tr_dpcknn_fused_supported(197, 384, 5) == 1 and lib.tr_dpcknn_fused_supported(197, 768, 5) == 1       # 224^2 inputs: the matrix fits the LDS
    assert lib.tr_dpcknn_fused_supported(577, 768, 5) == 0 and lib.tr_dpcknn_fused_supported(197, 100, 5) == 0       # 384^2 inputs; a width that is no multiple of 32
    z8 = (ctypes.c_double * 8)()
    assert lib.tr_adamw_step(None, None, 1, 1, 0.9, 0.999, 1e-8, 0.1, 0.03, z8, z8, 0, None) == -3                    # null item table
    assert lib.tr_adamw_step(p, p, 0, 0, 0.9, 0.999, 1e-8, 0.1, 0.03, z8, z8, 0, None) == -1                          # nothing to update
    assert lib.tr_adamw_step(p, p, 1, 1, 0.9, 0.999, 1e-8, 0.0, 0.03, z8, z8, 0, None) == -1                          # bias correction 0: a step count of 0


def test_fused_adamw_refuses_what_it_does_not_build():
    """optim.FusedAdamW mirrors torch.optim.AdamW's constructor; the variants it has no kernel for raise at construction, and a CPU model
    fails loudly at the first step (no CPU fallback)."""
    import torch
    from tokenreduction_amd.optim import FusedAdamW
    w = torch.nn.Parameter(torch.zeros(4, 4))
    with pytest.raises(NotImplementedError, match="amsgrad"):
        FusedAdamW([w], amsgrad=True)
    with pytest.raises(NotImplementedError, match="maximize"):
        FusedAdamW([w], maximize=True)
    opt = FusedAdamW([dict(params=[w], weight_decay=0.0)], lr=1e-3, betas=(0.9, 0.98))
    assert opt.param_groups[0]["lr"] == 1e-3 and opt.param_groups[0]["betas"] == (0.9, 0.98) and opt.param_groups[0]["weight_decay"] == 0.0
    assert opt.step() is None                                      # no gradients: nothing to do, nothing launched
    w.grad = torch.ones_like(w)
    with pytest.raises(Exception):                                 # CPU tensors: the launch is refused (invalid device pointer / no HIP device)
        opt.step()
    assert torch.equal(w.detach(), torch.zeros(4, 4))              # ... and nothing was updated on the host behind the library's back


def test_finetune_ingest_matches_the_reference(golden_dir):
    """f3: a DeiT-layout 224 x 224 checkpoint into a 384 x 384 model with another head.  tests/golden/finetune_ingest.npz holds what the
    REFERENCE's own statements (train.py:343-370, executed by gen_golden.py on a reference model) leave behind for the seeded checkpoint
    of tests/_params.finetune_ingest_setup; finetune.load_finetune_checkpoint must leave the same: the resized position embedding bit
    for bit (same torch bicubic kernel), the same keys loaded, the trunk weights in place."""
    from tests._params import finetune_ingest_setup
    from tokenreduction_amd.finetune import load_finetune_checkpoint
    g = np.load(os.path.join(golden_dir, "finetune_ingest.npz"))
    _, cfg384, ck = finetune_ingest_setup()
    model = tra.TopKVisionTransformer(img_size=384, patch_size=16, embed_dim=cfg384.embed_dim, depth=cfg384.depth, num_heads=cfg384.num_heads,
                                      mlp_ratio=4, qkv_bias=True, num_classes=cfg384.num_classes, args=_args(keep_rate=[0.5], reduction_loc=[1, 2]))
    head_before = model.head.weight.detach().clone()
    missing, unexpected = load_finetune_checkpoint(model, {"model": {k: v.clone() for k, v in ck.items()}})
    assert sorted(missing) == ["head.bias", "head.weight"] and not unexpected
    assert g["sizes"].tolist() == [14, 24, 1] and model.pos_embed.shape == (1, 577, cfg384.embed_dim)
    assert np.array_equal(model.pos_embed.detach().numpy(), g["pos_embed"])
    assert np.array_equal(model.blocks[1].mlp.fc1.weight.detach().numpy(), g["fc1_w_block1"])
    assert sorted(k for k in ck if k not in ("head.weight", "head.bias")) == g["keys_loaded"].tolist()
    assert torch.equal(model.head.weight, head_before)             # the new classifier keeps its initialisation
    # the reference's statements read state_dict(), patch_embed.num_patches and pos_embed of the model object: same attribute surface
    assert model.patch_embed.num_patches == 576 and "pos_embed" in model.state_dict()


def test_pretrained_true_loads_a_deit_checkpoint_from_deit_weights(tmp_path, monkeypatch):
    """The positive path of models_act.py:1130-1137: with ./deit_weights/<file of the DeiT url> present, create_model(pretrained=True)
    loads checkpoint["model"] non-strictly -- also into a model with its own classifier-sized head and extra stage modules."""
    from tokenreduction_amd.registry import deit_url_paths
    monkeypatch.chdir(tmp_path)
    src = tra.create_model("deit_tiny_patch16_224_local", pretrained=False, num_classes=1000)
    sd = {k: v.detach().clone() for k, v in src.state_dict().items()}
    os.makedirs("deit_weights")
    torch.save({"model": sd}, os.path.join("deit_weights", os.path.basename(deit_url_paths["deit_tiny_patch16_224"])))
    for name, extra in (("topk_tiny_patch16_224", {}), ("dpcknn_tiny_patch16_224", {"k_neighbors": 5, "equal_weight": False})):
        m = tra.create_model(name, pretrained=True, args=_args(keep_rate=[0.7], reduction_loc=[3, 6, 9], **extra))
        own = m.state_dict()
        assert all(torch.equal(own[k], v) for k, v in sd.items()), name
        assert set(own) >= set(sd)


@pytest.mark.parametrize("family", ["topk", "evit", "dyvit", "tome", "ats", "sit", "dpcknn", "sinkhorn", "kmedoids", "patchmerger"])
def test_stage_schedules_match_the_oracle(family):
    """The per-block token schedule every family derives from (keep_rate, reduction_loc) -- one rate (geometric) or an explicit
    list (ratios for topk/evit/dyvit, ABSOLUTE counts for the others; tome.py:145-156, ats.py:204-205, dpcknn.py:214-215) --
    must equal the oracle's, which is pinned on the reference."""
    import oracle
    from oracle import VitConfig
    cases = [([0.7], [3, 6, 9]), ([0.5], [1, 5]), ([0.9], [0, 4, 8, 11])]
    if family in ("topk", "evit", "dyvit"):
        cases += [([0.8, 0.5, 0.3], [2, 5, 8])]
    elif family == "ats":
        cases += [([120, 60, 30], [2, 5, 8])]
    else:
        cases += [([150, 100, 40], [2, 5, 8])]
    for kr, loc in cases:
        if family == "kmedoids" and 0 in loc:
            continue                                           # no previous attention at block 0 (kmedoids.py:240)
        args = _args(keep_rate=list(kr), reduction_loc=list(loc), dyvit_distill=False, k_neighbors=5, equal_weight=False,
                     cluster_iters=3, sinkhorn_eps=1.0)
        m = tra.create_model(f"{family}_tiny_patch16_224", pretrained=False, num_classes=10, img_size=224, args=args)
        cfg = VitConfig(family=family, embed_dim=192, depth=12, num_heads=3, keep_rate=list(kr), reduction_loc=list(loc))
        if family in ("topk", "evit"):
            want = oracle.stage_keep_counts(cfg)
        elif family == "dyvit":
            want = oracle.dyvit_keep_counts(cfg)
        elif family == "tome":
            want = oracle.tome_schedule(cfg)
        elif family == "ats":
            want = oracle.ats_token_bounds(cfg)
        else:
            want = oracle.sit_cluster_counts(cfg)              # sit / dpcknn / sinkhorn / kmedoids / patchmerger share the rule
        got = {i: k for i, k in enumerate(m._keep) if k}
        assert got == {int(k): int(v) for k, v in want.items() if v}, (family, kr, loc, got, want)


@pytest.mark.parametrize("name", ["heuristic_micro_l2", "heuristic_small_linf"])
def test_heuristic_masks_match_the_reference(name):
    """Constructor-time geometry (heuristic.py:157-224), no GPU needed: the visible patch ids of every block in the reduction range
    equal the reference's recorded Kept_Tokens_Abs, for the contiguous (linear radius ramp) and the listed-blocks variant."""
    import numpy as np
    from tests._params import GOLDEN_CASES
    case = GOLDEN_CASES[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    args = _args(keep_rate=list(case["keep_rate"]), reduction_loc=list(case["reduction_loc"]), heuristic_pattern=case["heuristic_pattern"],
                 not_contiguous=case["not_contiguous"], min_radius=case.get("min_radius"))
    m = tra.HeuristicVisionTransformer(img_size=224, patch_size=16, embed_dim=case["embed_dim"], depth=case["depth"],
                                       num_heads=case["num_heads"], mlp_ratio=4, qkv_bias=True, num_classes=case["num_classes"], args=args)
    keys = [k for k in g.files if k.startswith("keptabs_")]
    assert sorted(int(k.split("_")[1]) for k in keys) == sorted(m.reduction_loc)
    for k in keys:
        blk = int(k.split("_")[1])
        np.testing.assert_array_equal(m._block_mask(blk).nonzero(as_tuple=True)[0].numpy(), g[k][0])


def test_header_is_plain_c_and_the_integration_example_compiles(tmp_path):
    """The C ABI header must be usable from a C host: compile the documented snippet with gcc -std=c11."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "host.c"
    src.write_text('''
#include <stdio.h>
#include "tokenreduction_hip.h"
int run(const float* d_images_f32, float* d_logits_f32, void* d_workspace, int B, tr_stream_t stream, const tr_vit_weights* w) {
  tr_vit_config cfg = { .family = TR_FAMILY_TOPK, .img_size = 224, .patch = 16, .in_chans = 3, .embed_dim = 384, .depth = 12,
                        .num_heads = 6, .mlp_hidden = 1536, .num_classes = 1000, .ln_eps = 1e-6f, .precision = TR_PREC_BF16 };
  cfg.keep[3] = 137; cfg.keep[6] = 96; cfg.keep[9] = 67;
  size_t ws_bytes = tr_vit_workspace_bytes(&cfg, B);
  int rc = tr_vit_forward(&cfg, w, d_images_f32, d_logits_f32, d_workspace, ws_bytes, NULL, NULL, NULL, NULL, NULL, NULL, B, stream);
  if (rc != TR_OK) fprintf(stderr, "%s\\n", tr_last_error());
  return rc;
}
''')
    out = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-c", str(src), "-I", os.path.join(root, "include"), "-o", str(tmp_path / "host.o")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_heuristic_masks_match_the_oracle_for_every_range():
    """HeuristicVisionTransformer's constructor-time geometry (CPU code of the product) against the oracle's restatement of
    heuristic.py:157-224 (itself pinned on the reference): every pattern, depth, reduction range -- including ranges that start at
    block 0, where the reference's radius ramp has no block before the range -- min_radius, and the non-contiguous stage subsets."""
    import itertools
    import types
    import torch
    import tokenreduction_amd as tra
    from oracle.heuristic import heuristic_masks
    from tests._params import case_config
    checked = 0
    for pattern, depth in itertools.product(("l1", "l2", "linf"), (2, 5, 12)):
        ranges = [(s, e) for s in range(depth) for e in range(s, depth)]
        if depth == 12:
            ranges = [(0, 0), (0, 11), (0, 5), (1, 1), (1, 9), (3, 9), (10, 11), (11, 11)]
        for (start, end), mr, nc in itertools.product(ranges, (None, 2.0), (False, True)):
            loc = sorted({start, end}) if not nc else sorted({start, (start + end) // 2, end})
            for kr in ([0.7], [0.3]):
                case = dict(family="heuristic", embed_dim=64, depth=depth, num_heads=1, num_classes=8, keep_rate=kr, reduction_loc=loc, batch=1,
                            wseed=1, xseed=2)
                args = types.SimpleNamespace(keep_rate=kr, reduction_loc=loc, heuristic_pattern=pattern, not_contiguous=nc, min_radius=mr)
                m = tra.HeuristicVisionTransformer(img_size=224, patch_size=16, embed_dim=64, depth=depth, num_heads=1, mlp_ratio=1, qkv_bias=True,
                                                   num_classes=8, args=args)
                want = heuristic_masks(case_config(case), pattern, nc, mr)
                assert sorted(want) == sorted(int(b) for b in m.reduction_loc), (pattern, depth, loc, nc)
                for b, mask in want.items():
                    assert torch.equal(mask, m._block_mask(b)), (pattern, depth, loc, nc, mr, kr, b)
                    checked += 1
    assert checked > 1000


@pytest.mark.parametrize("name,kr,loc", [("evit_small_patch16_224", [0.3], [0, 1, 2, 3, 4]), ("topk_small_patch16_224", [0.004], [3]),
                                         ("dyvit_small_patch16_224", [0.3], [1, 2, 3, 4, 5]), ("dpcknn_small_patch16_224", [0.3], [1, 2, 3, 4, 5]),
                                         ("sit_small_patch16_224", [0.004], [3])])
def test_schedules_that_keep_no_token_raise(name, kr, loc):
    """int(0.3**5 * 196) = 0: the reference would carry on with the CLS token alone (a k = 0 topk is legal in torch); here 0 marks "no
    reduction at this block", so such a schedule must raise at construction instead of silently skipping the stage."""
    import types
    import tokenreduction_amd as tra
    args = types.SimpleNamespace(keep_rate=list(kr), reduction_loc=list(loc), dyvit_distill=False, k_neighbors=5, equal_weight=False, cluster_iters=3,
                                 sinkhorn_eps=1.0)
    with pytest.raises(ValueError, match="keeps 0 tokens"):
        tra.create_model(name, pretrained=False, num_classes=10, args=args)


def test_deepcopy_leaves_the_executor_caches_behind():
    """copy.deepcopy(model) (timm's ModelEmaV2, torch.optim.swa_utils) copies parameters, buffers and configuration, never the executor's
    state: workspaces hold captured graphs that cannot be copied, and a copied workspace would not match the copy's own packed weights."""
    import copy
    import types
    import tokenreduction_amd as tra

    class NoCopy:
        def __deepcopy__(self, memo):
            raise TypeError("cannot be deep-copied (stands in for torch.cuda.CUDAGraph)")

    args = types.SimpleNamespace(keep_rate=[0.7], reduction_loc=[1, 2], viz_mode=False)
    m = tra.TopKVisionTransformer(patch_size=16, embed_dim=128, depth=4, num_heads=2, mlp_ratio=4, qkv_bias=True, num_classes=8, args=args)
    m._ws = {2: {"graphs": {0: NoCopy()}}}
    m._packed = {"key": 1, "keep_alive": [NoCopy()]}
    m._last_ws = m._ws[2]
    m._tstate = NoCopy()
    m._noise_buf = NoCopy()
    twin = copy.deepcopy(m)
    assert twin._ws == {} and twin._packed is None and twin._tstate is None and twin._last_ws is None and twin._noise_buf is None
    assert twin._keep == m._keep and twin.precision == m.precision and twin.pruning_loc == m.pruning_loc
    for (n, a), (_, b) in zip(m.state_dict().items(), twin.state_dict().items()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), n
    assert isinstance(m._ws[2]["graphs"][0], NoCopy)      # the original is untouched
