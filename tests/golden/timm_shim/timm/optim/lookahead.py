class Lookahead:  # placeholder, never instantiated by the golden generator
    pass
