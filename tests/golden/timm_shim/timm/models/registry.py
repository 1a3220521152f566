_model_entrypoints = {}


def register_model(fn):
    _model_entrypoints[fn.__name__] = fn
    return fn
