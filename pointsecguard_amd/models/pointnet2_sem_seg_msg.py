"""PointNet++ MSG (multi-scale grouping) semantic-segmentation network behind the reference's module API.

Drop-in for the reference's PointNet/models/pointnet2_sem_seg_msg.py (get_model :7-42, get_loss :45-51): same class
names, constructor arguments, sub-module / parameter names (sa{l}.conv_blocks.{scale}.{layer} ...) and forward
contract  x[B,9,N] -> (log_softmax [B,N,13], l4_points [B,1024,16]).  The computation is one whole-network call
into libpsg.so (architecture PSG_PN2_ARCH_MSG of include/psg.h); autograd sees a single Function whose backward is
the HIP input-gradient pass, so every attack of pointsecguard_amd.attacks runs on it unchanged.

Eval mode only, like pointnet2_sem_seg.
"""
import torch.nn as nn
import torch.nn.functional as F

try:  # the reference imports `models.pointnet_util` with PointNet/ on sys.path
    from .pointnet_util import PointNetFeaturePropagation, PointNetSetAbstractionMsg
    from .pointnet2_sem_seg import _PN2Function, draw_fps_starts  # noqa: F401  (same RNG contract: 4 draws per forward)
except ImportError:  # pragma: no cover - flat import like the reference harness
    from pointnet_util import PointNetFeaturePropagation, PointNetSetAbstractionMsg
    from pointnet2_sem_seg import _PN2Function, draw_fps_starts  # noqa: F401

from pointsecguard_amd import runtime


class get_model(nn.Module):
    L4_CHANNELS = 1024
    ARCH = runtime.ARCH_MSG

    def __init__(self, num_classes):
        super(get_model, self).__init__()
        if num_classes != runtime.NUM_CLASSES:
            raise ValueError("the gfx950 kernels are specialised for the %d S3DIS classes" % runtime.NUM_CLASSES)
        self.sa1 = PointNetSetAbstractionMsg(1024, [0.05, 0.1], [16, 32], 9, [[16, 16, 32], [32, 32, 64]])
        self.sa2 = PointNetSetAbstractionMsg(256, [0.1, 0.2], [16, 32], 32 + 64, [[64, 64, 128], [64, 96, 128]])
        self.sa3 = PointNetSetAbstractionMsg(64, [0.2, 0.4], [16, 32], 128 + 128, [[128, 196, 256], [128, 196, 256]])
        self.sa4 = PointNetSetAbstractionMsg(16, [0.4, 0.8], [16, 32], 256 + 256, [[256, 256, 512], [256, 384, 512]])
        self.fp4 = PointNetFeaturePropagation(512 + 512 + 256 + 256, [256, 256])
        self.fp3 = PointNetFeaturePropagation(128 + 128 + 256, [256, 256])
        self.fp2 = PointNetFeaturePropagation(32 + 64 + 256, [256, 128])
        self.fp1 = PointNetFeaturePropagation(128, [128, 128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)
        self._psg_model = None
        self._psg_key = None
        self._psg_ws = {}
        self._generation = 0

    def _packed(self):
        tensors = list(self.parameters()) + list(self.buffers())
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._psg_model is None or key != self._psg_key:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._psg_model = runtime.PN2Model(runtime.fold_state_dict(sd, msg=True), arch=self.ARCH)
            self._psg_key = key
        return self._psg_model

    def _workspace(self, batch, n_point, n_forward):
        key = (batch, n_point)
        ws = self._psg_ws.get(key)
        if ws is None or ws.max_forwards < n_forward:
            ws = runtime.PN2Workspace(batch, n_point, n_forward, arch=self.ARCH)
            self._psg_ws[key] = ws
        return ws

    def forward(self, xyz):
        if self.training:
            raise NotImplementedError("pointsecguard_amd implements the eval-mode attack path only; call .eval() "
                                      "(training-mode BatchNorm / Dropout are out of scope)")
        runtime.require_cuda(xyz, "xyz")
        if xyz.dim() != 3 or xyz.shape[1] != 9:
            raise ValueError("expected input [B, 9, N], got %s" % (tuple(xyz.shape),))
        x, l4_points = _PN2Function.apply(xyz, self)
        return x, l4_points


class get_loss(nn.Module):
    def __init__(self):
        super(get_loss, self).__init__()

    def forward(self, pred, target, trans_feat, weight):
        return F.nll_loss(pred, target, weight=weight)
