class Adafactor:  # placeholder, never instantiated by the golden generator
    pass
