class NovoGrad:  # placeholder, never instantiated by the golden generator
    pass
