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


def _conv_stack(widths, act, norm, bias, drop):
    """Sub-modules of one BasicConv in state_dict order: per layer a 1x1 Conv2d, then the optional activation,
    BatchNorm2d (sized by the LAST width, like the reference) and Dropout2d."""
    stack = []
    for c_in, c_out in zip(widths[:-1], widths[1:]):
        stack.append(Conv2d(c_in, c_out, 1, bias=bias))
        stack.extend(layer for layer, wanted in ((act_layer(act) if act else None, act),
                                                 (norm_layer(norm, widths[-1]) if norm else None, norm),
                                                 (nn.Dropout2d(drop) if drop > 0 else None, drop > 0)) if wanted)
    return stack


def _default_init(module):
    """Kaiming-normal conv weights, zero biases, unit BatchNorm scale (what the reference applies on construction)."""
    for sub in module.modules():
        if isinstance(sub, nn.Conv2d):
            nn.init.kaiming_normal_(sub.weight)
            if sub.bias is not None:
                nn.init.zeros_(sub.bias)
        elif isinstance(sub, nn.BatchNorm2d):
            nn.init.ones_(sub.weight)
            nn.init.zeros_(sub.bias)


class BasicConv(Seq):
    """Sequential(Conv2d 1x1 [, ReLU] [, BatchNorm2d] [, Dropout2d]) per consecutive pair of `channels`."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0.0):
        super().__init__(*_conv_stack(list(channels), act, norm, bias, drop))
        self.reset_parameters()

    def reset_parameters(self):
        _default_init(self)


def batched_index_select(inputs, index):
    """inputs [B,C,N,1], index [B,N,k] -> [B,C,N,k] (torch_nn.py:82-98), gathered by the HIP row-gather kernel."""
    B, C, N, _ = inputs.shape
    k = index.shape[2]
    rows = inputs[:, :, :, 0].transpose(1, 2).contiguous().float()                    # [B,N,C] point-major
    out = runtime.gather_points(rows, index.reshape(B, N * k).to(torch.int32).contiguous())   # [B,N*k,C]
    return out.view(B, N, k, C).permute(0, 3, 1, 2)
