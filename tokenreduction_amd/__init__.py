"""tokenreduction_amd -- MI355X-native (gfx950) ViT-with-token-reduction hot path.

Drop-in for the reference's timm-style factory (`create_model(name, ..., args=args)`); all arithmetic
runs in hand-written HIP kernels behind the C ABI in include/tokenreduction_hip.h.
"""
from .registry import create_model, register_model, list_models, is_model  # noqa: F401
from .models import (VisionTransformer, TopKVisionTransformer, EfficientVisionTransformer, ToMeVisionTransformer,  # noqa: F401
                     DynamicVisionTransformer, SelfSlimmedVisionTransformer, DPCKNNVisionTransformer,
                     ATSVisionTransformer, SinkhornVisionTransformer, KMedoidsVisionTransformer,
                     PatchMergerVisionTransformer, HeuristicVisionTransformer,
                     VisionTransformerTeacher)

from . import harness  # noqa: F401,E402  (evaluate_multiclass / validate / write_viz)

__version__ = "0.1.0"
