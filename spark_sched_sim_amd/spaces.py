"""the one gymnasium type the reference observation carries (`gymnasium.spaces.GraphInstance`,
used at reference spark_sched_sim.py:394); re-used from gymnasium when it is installed."""
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
