"""Stand-in for numba: ``njit`` returns the undecorated Python function."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


jit = njit
