from .registry import register_model, _model_entrypoints


def create_model(model_name, pretrained=False, **kwargs):
    # timm 0.4.12 drops kwargs whose value is None before calling the entrypoint.
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _model_entrypoints[model_name](pretrained=pretrained, **kwargs)
