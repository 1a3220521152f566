"""timm-style model registry -- the reference's plugin boundary (models_act.py:63-1470,
callers train.py:322-331, validate.py:88-94):

    create_model(name, pretrained=..., num_classes=..., drop_rate=..., drop_path_rate=...,
                 drop_block_rate=None, img_size=..., args=Namespace(keep_rate, reduction_loc, ...))

Same factory names and fixed dims as models_act.py (tiny 192/3, small 384/6, base 768/12; depth 12,
mlp_ratio 4, qkv_bias, LayerNorm eps 1e-6): deit_*_local[_viz], topk_*, evit_*, tome_*, dyvit_* (eval and train) and
dyvit_*_teacher, sit_*, dpcknn_*, ats_*, sinkhorn_*, kmedoids_* (both equal_weight branches), patchmerger_*, heuristic_*; every name
runs in eval and in training mode.  What is not built raises instead of silently falling back: the distillation token (the reference's
own forward fails for it), attn_drop_rate > 0, CPU tensors.
"""
from __future__ import annotations

import os
from functools import partial

import torch
import torch.nn as nn

from .models import (ATSVisionTransformer, HeuristicVisionTransformer, VisionTransformerTeacher, KMedoidsVisionTransformer, PatchMergerVisionTransformer, SinkhornVisionTransformer, DPCKNNVisionTransformer, DynamicVisionTransformer, EfficientVisionTransformer, SelfSlimmedVisionTransformer, ToMeVisionTransformer,
                     TopKVisionTransformer, VisionTransformer)

_model_entrypoints = {}

_DIMS = {"tiny": (192, 3), "small": (384, 6), "base": (768, 12)}

# models_act.py:54-60
deit_url_paths = {
    "deit_tiny_patch16_224": "https://dl.fbaipublicfiles.com/deit/deit_tiny_patch16_224-a1311bcf.pth",
    "deit_small_patch16_224": "https://dl.fbaipublicfiles.com/deit/deit_small_patch16_224-cd65a155.pth",
    "deit_base_patch16_224": "https://dl.fbaipublicfiles.com/deit/deit_base_patch16_224-b5f2ef4d.pth",
}


def register_model(fn):
    _model_entrypoints[fn.__name__] = fn
    return fn


def list_models():
    return sorted(_model_entrypoints)


def is_model(name):
    return name in _model_entrypoints


def create_model(model_name, pretrained=False, **kwargs):
    """timm.models.create_model semantics: kwargs whose value is None are dropped before the factory is called."""
    if model_name not in _model_entrypoints:
        raise RuntimeError("Unknown model (%s)" % model_name)
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _model_entrypoints[model_name](pretrained=pretrained, **kwargs)


def _load_deit_weights(model, key):
    """models_act.py:1130-1137 downloads DeiT weights into ./deit_weights; there is no network here, so only a
    file already present at that path is used."""
    fname = os.path.join("./deit_weights", os.path.basename(deit_url_paths[key]))
    if not os.path.exists(fname):
        raise RuntimeError(f"pretrained=True needs {fname} (download of {deit_url_paths[key]} is impossible offline)")
    checkpoint = torch.load(fname, map_location="cpu")
    model.load_state_dict(checkpoint["model"], strict=False)


def _make(cls, size, key_prefix, drop_args):
    D, H = _DIMS[size]

    def factory(pretrained=False, **kwargs):
        args = kwargs.get("args", None)
        if args is not None and getattr(args, "distillation_type", "none") != "none":
            raise NotImplementedError("DeiT distillation token models are not on the hot path (SURVEY App. A.10)")
        if cls is DynamicVisionTransformer:
            kwargs["dyvit_distillation"] = bool(args.dyvit_distill)          # models_act.py:351 (required attribute)
        if drop_args:
            kwargs.pop("args", None)      # models_act.py:74 deit_*_local drops args
        model = cls(patch_size=16, embed_dim=D, depth=12, num_heads=H, mlp_ratio=4, qkv_bias=True,
                    norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
        key = f"deit_{size}_patch16_224"
        model.default_cfg = {"url": deit_url_paths[key], "num_classes": 1000, "input_size": (3, 224, 224),
                             "first_conv": "patch_embed.proj", "classifier": "head"}
        if pretrained:
            _load_deit_weights(model, key)
        return model
    return factory


for _size in _DIMS:
    for _name, _cls, _drop in ((f"deit_{_size}_patch16_224_local", VisionTransformer, True),
                               (f"deit_{_size}_patch16_224_local_viz", VisionTransformer, False),
                               (f"topk_{_size}_patch16_224", TopKVisionTransformer, False),
                               (f"evit_{_size}_patch16_224", EfficientVisionTransformer, False),
                               (f"tome_{_size}_patch16_224", ToMeVisionTransformer, False),
                               (f"sit_{_size}_patch16_224", SelfSlimmedVisionTransformer, False),
                               (f"dyvit_{_size}_patch16_224", DynamicVisionTransformer, False),
                               (f"dpcknn_{_size}_patch16_224", DPCKNNVisionTransformer, False),
                               (f"ats_{_size}_patch16_224", ATSVisionTransformer, False),
                               (f"sinkhorn_{_size}_patch16_224", SinkhornVisionTransformer, False),
                               (f"kmedoids_{_size}_patch16_224", KMedoidsVisionTransformer, False),
                               (f"patchmerger_{_size}_patch16_224", PatchMergerVisionTransformer, False),
                               (f"heuristic_{_size}_patch16_224", HeuristicVisionTransformer, False)):
        _f = _make(_cls, _size, _name, _drop)
        _f.__name__ = _name
        register_model(_f)
    _f = _make(VisionTransformerTeacher, _size, f"dyvit_{_size}_patch16_224_teacher", False)    # models_act.py:383-405
    _f.__name__ = f"dyvit_{_size}_patch16_224_teacher"
    register_model(_f)
