from . import _rng


def randrange(*args):
    return _rng.randrange(*args)


def randint(a, b):
    return _rng.randint(a, b)
