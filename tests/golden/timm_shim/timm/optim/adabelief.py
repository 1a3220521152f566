class AdaBelief:  # placeholder, never instantiated by the golden generator
    pass
