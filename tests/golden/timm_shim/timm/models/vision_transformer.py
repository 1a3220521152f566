"""Base class comes straight from the reference's own in-tree copy of timm's ViT
(/root/reference/models/deit_viz.py:75-212), imported at run time -- nothing copied."""
from timm.data import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD


def _cfg(url='', **kwargs):
    return {'url': url, 'num_classes': 1000, 'input_size': (3, 224, 224),
            'mean': IMAGENET_DEFAULT_MEAN, 'std': IMAGENET_DEFAULT_STD, **kwargs}


default_cfgs = {k: _cfg() for k in (
    'deit_tiny_patch16_224', 'deit_small_patch16_224', 'deit_base_patch16_224',
    'deit_tiny_distilled_patch16_224', 'deit_small_distilled_patch16_224',
    'deit_base_distilled_patch16_224')}


def __getattr__(name):
    if name == 'VisionTransformer':
        from models.deit_viz import VisionTransformer  # reference file, read-only mount
        return VisionTransformer
    raise AttributeError(name)
