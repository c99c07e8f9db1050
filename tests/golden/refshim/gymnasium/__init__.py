"""Stand-in for the parts of gymnasium the reference env classes touch.

``Env.reset(seed)`` reseeds ``np_random`` the way gymnasium's
``seeding.np_random`` does: ``numpy.random.default_rng(seed)`` (PCG64 seeded
through SeedSequence).
"""
import numpy as np

from . import spaces  # noqa: F401
from . import core  # noqa: F401
from . import envs  # noqa: F401
from .core import Env, Wrapper, ObservationWrapper  # noqa: F401


def make(*args, **kwargs):
    raise NotImplementedError
