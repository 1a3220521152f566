class NvNovoGrad:  # placeholder, never instantiated by the golden generator
    pass
