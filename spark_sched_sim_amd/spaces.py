"""the gymnasium types the reference env exposes (`gymnasium.spaces.GraphInstance` in the observation,
spark_sched_sim.py:394; `Dict` / `Discrete(n, start)` action space, :85-94, whose `stage_idx.n` follows
the observation, :403-404); re-used from gymnasium when it is installed, minimal stand-ins otherwise."""
from __future__ import annotations

from typing import NamedTuple

import numpy as np

try:  # pragma: no cover - gymnasium is optional
    from gymnasium.spaces import GraphInstance  # type: ignore
except Exception:

    class GraphInstance(NamedTuple):  # type: ignore
        nodes: np.ndarray
        edges: np.ndarray | None
        edge_links: np.ndarray | None


try:  # pragma: no cover - gymnasium is optional
    from gymnasium.spaces import Dict, Discrete  # type: ignore
except Exception:

    class Discrete:  # type: ignore
        """{start, ..., start + n - 1}"""

        def __init__(self, n: int, start: int = 0):
            self.n, self.start = int(n), int(start)

        def contains(self, x) -> bool:
            return isinstance(x, (int, np.integer)) and self.start <= int(x) < self.start + self.n

        def sample(self) -> int:
            return int(np.random.randint(self.start, self.start + self.n))

        def __repr__(self):
            return f"Discrete({self.n}, start={self.start})"

    class Dict(dict):  # type: ignore
        """dict of spaces; `contains` requires exactly the same keys (gymnasium's rule)"""

        def contains(self, x) -> bool:
            return isinstance(x, dict) and x.keys() == self.keys() and all(self[k].contains(x[k]) for k in self)

        def sample(self) -> dict:
            return {k: v.sample() for k, v in self.items()}


try:  # pragma: no cover - gymnasium is optional
    from gymnasium.spaces import Box, Graph, Sequence  # type: ignore
except Exception:

    class Box:  # type: ignore
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    class Graph:  # type: ignore
        def __init__(self, node_space, edge_space):
            self.node_space, self.edge_space = node_space, edge_space

        def contains(self, x) -> bool:
            return hasattr(x, "nodes") and hasattr(x, "edge_links") and np.asarray(x.nodes).ndim == 2 and \
                np.asarray(x.nodes).shape[1:] == self.node_space.shape

    class Sequence:  # type: ignore
        def __init__(self, space, stack: bool = False):
            self.feature_space, self.stack = space, stack

        def contains(self, x) -> bool:
            return all(self.feature_space.contains(int(v)) for v in x)


def make_observation_space(num_executors: int, num_node_features: int = 3) -> "Dict":
    """the reference's observation space at construction (spark_sched_sim.py:96-125). Two bounds follow
    the episode, as in the reference: `dag_ptr`'s feature space n = number of active stages + 1 after
    every observation (:403) and `source_job_idx`.n = number of jobs + 1 after reset (:157)."""
    return Dict({
        "dag_batch": Graph(node_space=Box(0, np.inf, (num_node_features,)), edge_space=Discrete(1)),
        "dag_ptr": Sequence(Discrete(1), stack=True),
        "num_committable_execs": Discrete(num_executors + 1),
        "source_job_idx": Discrete(1),
        "exec_supplies": Sequence(Discrete(2 * num_executors), stack=True),
    })


def make_action_space(num_executors: int) -> "Dict":
    """the reference's action space at construction (spark_sched_sim.py:85-94)"""
    return Dict({"stage_idx": Discrete(1, start=-1), "num_exec": Discrete(num_executors, start=1)})
