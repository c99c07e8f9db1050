"""Container-only stand-ins for gymnasium.spaces."""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = shape
        self.dtype = dtype
        self._rng = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]

    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        return True


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        super().__init__(shape, dtype, seed)
        self.low, self.high = low, high


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64, seed)
        self.n, self.start = int(n), int(start)

    def sample(self):
        return int(self.start + self._rng.integers(self.n))


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self):
        return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        super().__init__(None, None, seed)
        self.spaces = dict(spaces or {}, **kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def __setitem__(self, k, v):
        self.spaces[k] = v

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()


class Tuple(Space):
    def __init__(self, spaces, seed=None):
        super().__init__(None, None, seed)
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)
