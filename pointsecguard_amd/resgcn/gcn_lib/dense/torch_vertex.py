"""Graph-convolution modules with the reference's names and parameter layout
(ResGCN/gcn_lib/dense/torch_vertex.py:8-115).  Inside DenseDeepGCN they are executed by the whole-network kernels of
libpsg; called on their own (`module(x)` / `module(x, edge_index)`, x [B,C,N,1] like the reference) they run on the
per-operator entry points (pointsecguard_amd/ops.py: psg_edgeconv_fwd/bwd, psg_mrconv_gather_fwd/bwd, psg_gcn_knn): eval mode only, gradients for the
input only, EdgeConv with the reference's configuration (relu, batch norm, 64 output channels, k = 16)."""
import torch
from torch import nn

from pointsecguard_amd import ops

from .torch_edge import DenseDilatedKnnGraph
from .torch_nn import BasicConv

__all__ = ["EdgeConv2d", "GraphConv2d", "DynConv2d", "ResDynBlock2d", "MRConv2d", "PlainDynBlock2d", "DenseDynBlock2d"]

def _edge_layer(seq):
    """BasicConv([2C, 64], 'relu', 'batch', bias) = Sequential(Conv2d, ReLU, BatchNorm2d) -> (W, b, scale, shift)."""
    conv = seq[0]
    if len(seq) != 3 or not isinstance(seq[1], nn.ReLU) or not isinstance(seq[2], nn.BatchNorm2d):
        raise NotImplementedError("the stand-alone graph convolutions implement the reference's configuration act='relu', "
                                  "norm='batch' (Conv -> ReLU -> BatchNorm, torch_nn.py:55-75)")
    if seq.training:
        raise NotImplementedError("pointsecguard_amd implements the eval-mode attack path only; call .eval()")
    scale, shift = ops.bn_affine(seq[2])
    bias = conv.bias if conv.bias is not None else torch.zeros(conv.weight.shape[0], device=conv.weight.device)
    return conv.weight.reshape(conv.weight.shape[0], -1), bias, scale, shift


def _neighbours(edge_index, n_point):
    """edge_index [2,B,N,k] -> int32 neighbour table [B,N,k].  The kernels take the centre of edge (b, i, k) to be vertex i
    itself (what DenseDilatedKnnGraph produces, torch_edge.py:57-58); the reference gathers x_i through edge_index[1]
    (torch_vertex.py:17,32), so any other centre row - and any index outside the room - is refused instead of silently
    computing something else."""
    if edge_index.dim() != 4 or edge_index.shape[0] != 2 or edge_index.shape[2] != n_point:
        raise ValueError("edge_index must be [2, B, N=%d, k], got %s" % (n_point, tuple(edge_index.shape)))
    nbr, ctr = edge_index[0], edge_index[1]
    own = torch.arange(n_point, device=edge_index.device, dtype=ctr.dtype).view(1, n_point, 1)
    if not bool((ctr == own).all()):
        raise NotImplementedError("edge_index[1] must be the vertex's own index (arange over N broadcast over B and k): the "
                                  "stand-alone graph convolutions do not gather the centre through an index")
    if bool((nbr < 0).any()) or bool((nbr >= n_point).any()):
        raise ValueError("edge_index[0] holds neighbour indices outside [0, %d)" % n_point)
    return nbr.to(torch.int32).contiguous()


def _rows(x):
    """[B,C,N,1] -> contiguous point-major [B,N,C]."""
    return x[:, :, :, 0].permute(0, 2, 1).contiguous().float()


class EdgeConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super(EdgeConv2d, self).__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index):
        """x [B,C,N,1], edge_index [2,B,N,16] (neighbour, centre; the centre row is arange) -> [B,64,N,1]:
        max_k BasicConv(cat[x_i, x_j - x_i]) (torch_vertex.py:31-35)."""
        w, b, scale, shift = _edge_layer(self.nn)
        nbr = _neighbours(edge_index, x.shape[2])
        y = ops.EdgeConv.apply(_rows(x), nbr, w, b, scale, shift)
        return y.permute(0, 2, 1).unsqueeze(-1)


class MRConv2d(nn.Module):
    """Max-relative graph convolution: BasicConv(cat[x, max_j (x_j - x_i)]) (torch_vertex.py:8-20)."""

    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super(MRConv2d, self).__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index):
        """x [B,C,N,1], edge_index [2,B,N,16] -> [B,M,N,1] (torch_vertex.py:16-20)."""
        w, b, scale, shift = _edge_layer(self.nn)
        nbr = _neighbours(edge_index, x.shape[2])
        y = ops.MRConv.apply(_rows(x), nbr, w, b, scale, shift)
        return y.permute(0, 2, 1).unsqueeze(-1)


class GraphConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, conv="edge", act="relu", norm=None, bias=True):
        super(GraphConv2d, self).__init__()
        if conv == "edge":
            self.gconv = EdgeConv2d(in_channels, out_channels, act, norm, bias)
        elif conv == "mr":
            self.gconv = MRConv2d(in_channels, out_channels, act, norm, bias)
        else:
            raise NotImplementedError("conv:{} is not supported".format(conv))

    def forward(self, x, edge_index):
        return self.gconv(x, edge_index)


class DynConv2d(GraphConv2d):
    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None,
                 bias=True, stochastic=False, epsilon=0.0, knn="matrix"):
        super(DynConv2d, self).__init__(in_channels, out_channels, conv, act, norm, bias)
        if knn != "matrix":
            raise NotImplementedError("only the default dense 'matrix' kNN is implemented")
        self.k, self.d = kernel_size, dilation
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)

    def forward(self, x):
        """x [B,C,N,1]: dilated kNN graph of x itself (torch_edge.py:45-79), then the graph convolution (torch_vertex.py:69-71)."""
        edge_index = self.dilated_knn_graph(x)
        return super(DynConv2d, self).forward(x, edge_index)


class ResDynBlock2d(nn.Module):
    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None, bias=True,
                 stochastic=False, epsilon=0.0, knn="matrix", res_scale=1):
        super(ResDynBlock2d, self).__init__()
        if res_scale != 1:
            raise NotImplementedError("res_scale != 1 is not used by the reference configuration")
        self.body = DynConv2d(in_channels, in_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                              epsilon, knn)
        self.res_scale = res_scale

    def forward(self, x):
        return self.body(x) + x * self.res_scale


class PlainDynBlock2d(nn.Module):
    """Plain dynamic graph-convolution block: body(x) (torch_vertex.py:74-85)."""

    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None, bias=True,
                 stochastic=False, epsilon=0.0, knn="matrix"):
        super(PlainDynBlock2d, self).__init__()
        self.body = DynConv2d(in_channels, in_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                              epsilon, knn)

    def forward(self, x):
        return self.body(x)


class DenseDynBlock2d(nn.Module):
    """Dense dynamic graph-convolution block: cat(x, body(x)) (torch_vertex.py:103-115)."""

    def __init__(self, in_channels, out_channels=64, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None,
                 bias=True, stochastic=False, epsilon=0.0, knn="matrix"):
        super(DenseDynBlock2d, self).__init__()
        self.body = DynConv2d(in_channels, out_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                              epsilon, knn)

    def forward(self, x):
        return torch.cat((x, self.body(x)), 1)
