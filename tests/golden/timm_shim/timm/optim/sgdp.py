class SGDP:  # placeholder, never instantiated by the golden generator
    pass
