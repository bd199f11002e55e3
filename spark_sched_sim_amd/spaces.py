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


def make_action_space(num_executors: int) -> "Dict":
    """the reference's action space at construction (spark_sched_sim.py:85-94)"""
    return Dict({"stage_idx": Discrete(1, start=-1), "num_exec": Discrete(num_executors, start=1)})
