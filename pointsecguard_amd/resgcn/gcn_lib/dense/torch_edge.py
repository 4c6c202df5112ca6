"""Dense dilated kNN graph with the reference's names (ResGCN/gcn_lib/dense/torch_edge.py:6-79), computed by
libpsg: fp32 MFMA distance matrix in the reference's evaluation order + in-LDS bitonic selection."""
import torch
from torch import nn

from pointsecguard_amd import runtime

__all__ = ["DenseDilated", "pairwise_distance", "dense_knn_matrix", "DenseDilatedKnnGraph"]

_ws_cache = {}


def _ws(batch, n_point):
    key = (batch, n_point)
    if key not in _ws_cache:
        _ws_cache[key] = runtime.GCNWorkspace(batch, n_point, 1)
    return _ws_cache[key]


class DenseDilated(nn.Module):
    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super(DenseDilated, self).__init__()
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k

    def forward(self, edge_index):
        if self.stochastic:
            if torch.rand(1) < self.epsilon and self.training:      # same RNG consumption as the reference (:21)
                randnum = torch.randperm(self.k * self.dilation)[:self.k]
                return edge_index[:, :, :, randnum]
        return edge_index[:, :, :, ::self.dilation]


def pairwise_distance(x):
    """x [B,N,C] -> [B,N,N] = (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2 with the reference's fp32 order."""
    from pointsecguard_amd import ops
    return ops.pairwise_distance(x)


def dense_knn_matrix(x, k=16):
    """x [B,C,N,1] -> edge_index [2,B,N,k] (nn_idx, center_idx), torch_edge.py:45-59.  k must be a multiple of 16
    (k = 16 * dilation, which is how DenseDilatedKnnGraph calls it)."""
    if k % 16:
        raise NotImplementedError("the gfx950 kNN kernel emits 16 neighbours at a dilation; k must be 16*d")
    B, C, N, _ = x.shape
    rows = x.detach()[:, :, :, 0].transpose(1, 2).contiguous().float()
    d = k // 16
    ws = _ws(B, N)
    # the kernel returns ranks 0, d, 2d, ...; the full top-k list is not materialised
    nn_idx = ws.knn(rows, d).long()
    center = torch.arange(0, N, device=x.device).repeat(B, nn_idx.shape[2], 1).transpose(2, 1)
    return torch.stack((nn_idx, center), dim=0), d


class DenseDilatedKnnGraph(nn.Module):
    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super(DenseDilatedKnnGraph, self).__init__()
        if k != 16:
            raise NotImplementedError("the gfx950 kNN kernel is specialised for the default k = 16")
        self.dilation, self.stochastic, self.epsilon, self.k = dilation, stochastic, epsilon, k
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)

    def forward(self, x):
        if self.stochastic:
            torch.rand(1)                                            # RNG parity with DenseDilated.forward (:21)
        edge_index, _ = dense_knn_matrix(x, self.k * self.dilation)  # already dilated by the kernel
        return edge_index
