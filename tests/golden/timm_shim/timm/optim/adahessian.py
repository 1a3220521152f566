class Adahessian:  # placeholder, never instantiated by the golden generator
    pass
