"""Layer containers with the reference's names and parameter layout (ResGCN/gcn_lib/dense/torch_nn.py:9-98):
BasicConv = Sequential(Conv2d 1x1 [, ReLU] [, BatchNorm2d]) so that state_dicts are interchangeable.  Inside
DenseDeepGCN their arithmetic is executed by libpsg (csrc/psg_resgcn.hip)."""
import torch
from torch import nn
from torch.nn import Conv2d, Sequential as Seq

from pointsecguard_amd import runtime

__all__ = ["act_layer", "norm_layer", "BasicConv", "batched_index_select"]


def act_layer(act, inplace=False, neg_slope=0.2, n_prelu=1):
    if act.lower() != "relu":
        raise NotImplementedError("activation layer [%s]: the gfx950 kernels implement the default 'relu'" % act)
    return nn.ReLU(inplace)


def norm_layer(norm, nc):
    if norm.lower() != "batch":
        raise NotImplementedError("normalization layer [%s]: the gfx950 kernels implement the default 'batch'" % norm)
    return nn.BatchNorm2d(nc, affine=True)


class BasicConv(Seq):
    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0.0):
        m = []
        for i in range(1, len(channels)):
            m.append(Conv2d(channels[i - 1], channels[i], 1, bias=bias))
            if act:
                m.append(act_layer(act))
            if norm:
                m.append(norm_layer(norm, channels[-1]))
            if drop > 0:
                m.append(nn.Dropout2d(drop))
        super(BasicConv, self).__init__(*m)
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


def batched_index_select(inputs, index):
    """inputs [B,C,N,1], index [B,N,k] -> [B,C,N,k] (torch_nn.py:82-98), gathered by the HIP row-gather kernel."""
    B, C, N, _ = inputs.shape
    k = index.shape[2]
    rows = inputs[:, :, :, 0].transpose(1, 2).contiguous().float()                    # [B,N,C] point-major
    out = runtime.gather_points(rows, index.reshape(B, N * k).to(torch.int32).contiguous())   # [B,N*k,C]
    return out.view(B, N, k, C).permute(0, 3, 1, 2)
