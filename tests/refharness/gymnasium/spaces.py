"""the few gymnasium spaces the reference env constructs (spark_sched_sim.py:85-125)."""
from __future__ import annotations

from typing import NamedTuple

import numpy as np


class GraphInstance(NamedTuple):
    nodes: np.ndarray
    edges: np.ndarray | None
    edge_links: np.ndarray | None


class Space:
    def contains(self, x) -> bool:
        raise NotImplementedError

    def __contains__(self, x) -> bool:
        return self.contains(x)


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        self.n = int(n)
        self.start = int(start)

    def contains(self, x) -> bool:
        if isinstance(x, int):
            as_int = x
        elif isinstance(x, (np.generic, np.ndarray)) and (
            np.issubdtype(x.dtype, np.integer) and x.shape == ()
        ):
            as_int = int(x)
        else:
            return False
        return bool(self.start <= as_int < self.start + self.n)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == tuple(self.shape) and bool(np.all(x >= self.low) and np.all(x <= self.high))


class MultiBinary(Space):
    def __init__(self, n, seed=None):
        self.n = n

    def contains(self, x) -> bool:
        return True


class Sequence(Space):
    def __init__(self, space, seed=None, stack=False):
        self.feature_space = space
        self.stack = stack

    def contains(self, x) -> bool:
        return all(self.feature_space.contains(v) for v in x)


class Graph(Space):
    def __init__(self, node_space, edge_space, seed=None):
        self.node_space = node_space
        self.edge_space = edge_space

    def contains(self, x) -> bool:
        return isinstance(x, GraphInstance)


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        self.spaces = dict(spaces or {}, **kw)

    def __getitem__(self, key):
        return self.spaces[key]

    def __setitem__(self, key, value):
        self.spaces[key] = value

    def keys(self):
        return self.spaces.keys()

    def contains(self, x) -> bool:
        # same rule as gymnasium.spaces.Dict.contains: exact key set, every value in its space
        if isinstance(x, dict) and x.keys() == self.spaces.keys():
            return all(x[k] in self.spaces[k] for k in self.spaces.keys())
        return False
