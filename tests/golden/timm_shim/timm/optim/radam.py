class RAdam:  # placeholder, never instantiated by the golden generator
    pass
