import torch


class Batch:
    def __init__(self, x=None, edge_index=None, ptr=None, batch=None, _num_graphs=None):
        self._d = dict(x=x, edge_index=edge_index, ptr=ptr, batch=batch)
        self._num_graphs = _num_graphs

    def __getattr__(self, k):
        d = object.__getattribute__(self, "_d")
        if k in d:
            return d[k]
        raise AttributeError(k)

    def __getitem__(self, k):
        return self._d[k]

    def __setitem__(self, k, v):
        self._d[k] = v

    def __contains__(self, k):
        return k in self._d

    def to(self, device, non_blocking=False):
        for k, v in self._d.items():
            if torch.is_tensor(v):
                self._d[k] = v.to(device)
        return self
