class Nadam:  # placeholder, never instantiated by the golden generator
    pass
