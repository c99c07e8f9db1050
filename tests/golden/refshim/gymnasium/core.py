from typing import Any, TypeVar

import numpy as np

ActType = TypeVar("ActType")
ObsType = TypeVar("ObsType")


class Env:
    metadata: dict = {}
    render_mode = None
    _np_random = None

    @property
    def np_random(self) -> np.random.Generator:
        if self._np_random is None:
            self._np_random = np.random.default_rng()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.default_rng(seed)

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        return self.observation(obs), info

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info

    def observation(self, obs):
        raise NotImplementedError
