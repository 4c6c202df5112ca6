"""Graph-convolution modules with the reference's names and parameter layout
(ResGCN/gcn_lib/dense/torch_vertex.py:8-115).  They carry the parameters; DenseDeepGCN executes them through
libpsg as whole-network kernels."""
from torch import nn

from .torch_edge import DenseDilatedKnnGraph
from .torch_nn import BasicConv

__all__ = ["EdgeConv2d", "GraphConv2d", "DynConv2d", "ResDynBlock2d", "MRConv2d", "PlainDynBlock2d", "DenseDynBlock2d"]

_MSG = "executed by the fused whole-network kernels of DenseDeepGCN; call the parent model"


class EdgeConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super(EdgeConv2d, self).__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index):
        raise NotImplementedError("EdgeConv2d.forward is " + _MSG)


class MRConv2d(nn.Module):
    """Max-relative graph convolution: BasicConv(cat[x, max_j (x_j - x_i)]) (torch_vertex.py:8-20)."""

    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super(MRConv2d, self).__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index):
        raise NotImplementedError("MRConv2d.forward is " + _MSG)


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
        raise NotImplementedError("DynConv2d.forward is " + _MSG)


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
        raise NotImplementedError("ResDynBlock2d.forward is " + _MSG)


class PlainDynBlock2d(nn.Module):
    """Plain dynamic graph-convolution block: body(x) (torch_vertex.py:74-85)."""

    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None, bias=True,
                 stochastic=False, epsilon=0.0, knn="matrix"):
        super(PlainDynBlock2d, self).__init__()
        self.body = DynConv2d(in_channels, in_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                              epsilon, knn)

    def forward(self, x):
        raise NotImplementedError("PlainDynBlock2d.forward is " + _MSG)


class DenseDynBlock2d(nn.Module):
    """Dense dynamic graph-convolution block: cat(x, body(x)) (torch_vertex.py:103-115)."""

    def __init__(self, in_channels, out_channels=64, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None,
                 bias=True, stochastic=False, epsilon=0.0, knn="matrix"):
        super(DenseDynBlock2d, self).__init__()
        self.body = DynConv2d(in_channels, out_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                              epsilon, knn)

    def forward(self, x):
        raise NotImplementedError("DenseDynBlock2d.forward is " + _MSG)
