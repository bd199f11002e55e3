"""Minimal stand-in for the `gymnasium` package (not installed in this image, no network).

Test infrastructure only: it exists so that the *reference* env can be imported, unmodified,
from /root/reference when golden fixtures are (re)generated (tests/golden/make_golden.py), and so
that the product's optional gymnasium integration can be exercised. It implements just the
surface the reference touches: `Env.reset(seed)` seeding exactly like
gymnasium 0.29 (`Generator(PCG64(SeedSequence(seed)))`), the Wrapper family with attribute
forwarding, a handful of spaces with `contains`, and `register`/`make`.
"""
from __future__ import annotations

import numpy as np

from . import spaces  # noqa: F401
from .envs.registration import make, register  # noqa: F401


class Env:
    metadata: dict = {}
    render_mode = None
    _np_random = None

    @property
    def np_random(self) -> np.random.Generator:
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
        return self._np_random

    @np_random.setter
    def np_random(self, value) -> None:
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

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

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        return self.observation(obs), info

    def step(self, action):
        obs, rew, term, trunc, info = self.env.step(action)
        return self.observation(obs), rew, term, trunc, info

    def observation(self, observation):
        raise NotImplementedError


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))

    def action(self, action):
        raise NotImplementedError
