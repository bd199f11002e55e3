from __future__ import annotations

import importlib

_REGISTRY: dict[str, str] = {}


def register(id: str, entry_point: str, **kwargs) -> None:
    _REGISTRY[id] = entry_point


def make(id: str, **kwargs):
    """`make("module:EnvId", **kw)` imports `module` first (which registers), like gymnasium."""
    if ":" in id:
        mod, id = id.split(":", 1)
        importlib.import_module(mod)
    entry = _REGISTRY[id]
    if callable(entry):
        return entry(**kwargs)
    mod_name, attr = entry.split(":")
    cls = getattr(importlib.import_module(mod_name), attr)
    return cls(**kwargs)
