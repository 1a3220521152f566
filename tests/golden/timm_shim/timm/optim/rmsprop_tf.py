class RMSpropTF:  # placeholder, never instantiated by the golden generator
    pass
