import torch


def index_to_mask(index, size=None):
    m = torch.zeros(size, dtype=torch.bool, device=index.device)
    m[index] = True
    return m


def mask_to_index(mask):
    return mask.nonzero(as_tuple=False).view(-1)


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    assert ptr is not None
    out = torch.empty_like(src)
    for a, b in zip(ptr[:-1].tolist(), ptr[1:].tolist()):
        out[a:b] = torch.softmax(src[a:b], dim)
    return out
