class AdamP:  # placeholder, never instantiated by the golden generator
    pass
