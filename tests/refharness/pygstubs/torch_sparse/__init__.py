"""functional stand-in for torch_sparse: COO adjacency + matmul via index_add (test infrastructure)"""
import torch


class SparseTensor:
    def __init__(self, row, col, sparse_sizes, is_sorted=False, trust_data=False):
        self.row, self.col, self.sizes = row, col, sparse_sizes

    def t(self):
        return SparseTensor(self.col, self.row, (self.sizes[1], self.sizes[0]))


def matmul(adj, x):
    out = torch.zeros((adj.sizes[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    out.index_add_(0, adj.row, x[adj.col])
    return out
