"""PointNet++ SSG semantic-segmentation network behind the reference's module API.

Drop-in for the reference's PointNet/models/pointnet2_sem_seg.py (get_model :6-40, get_loss :43-49):
same class names, constructor arguments, sub-module / parameter names (so reference checkpoints
load with load_state_dict) and forward contract  x[B,9,N] -> (log_softmax [B,N,13], l4_points
[B,512,16]).  The computation itself is ONE whole-network call into libpsg.so (hand-written gfx950
kernels); autograd sees a single Function whose backward is the HIP input-gradient pass.

Eval mode only: the attacks always run the model in eval() (attack.py:188 of the reference);
training-mode BatchNorm is out of scope and raises.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

try:  # the reference imports `models.pointnet_util` with PointNet/ on sys.path
    from .pointnet_util import PointNetFeaturePropagation, PointNetSetAbstraction
except ImportError:  # pragma: no cover - flat import (`sys.path.append('models')`) like the reference harness
    from pointnet_util import PointNetFeaturePropagation, PointNetSetAbstraction

from pointsecguard_amd import _lib, runtime

LEVEL_N = (None, 1024, 256, 64)  # population each farthest_point_sample draws its start from


def draw_fps_starts(batch, n_point, n_forward=1, pinned=False):
    """The torch.randint draws of pointnet_util.py:75, in reference call order (4 per forward) from
    the global CPU generator, so a seeded run consumes the RNG stream exactly like the reference.
    Returns an int32 tensor [n_forward, 4, batch]; pinned=True puts it in pinned host memory, so that
    `upload(..., pin=True)` below is a copy the host does not wait for."""
    raw = torch.empty(n_forward, 4, batch, dtype=torch.long)
    for f in range(n_forward):
        for lvl in range(4):
            n = n_point if lvl == 0 else LEVEL_N[lvl]
            torch.randint(0, n, (batch,), dtype=torch.long, out=raw[f, lvl])      # (the same draws as one call per level)
    out = torch.empty(n_forward, 4, batch, dtype=torch.int32, pin_memory=pinned and torch.cuda.is_available())
    out.copy_(raw)
    return out


def upload(t, device, pin=False):
    """Host tensor -> device.  pin=True: through pinned memory, so the copy is enqueued on the current stream and the call
    returns without waiting for what is queued before it (torch's host allocator keeps the pinned block until the copy has
    run) - the one-thread API paths (NB_attack, the model's forward, the whole-scene harness) use it so that the host
    prepares call i + 1 while the GPU runs call i.  NOT for code that runs beside a hipGraph capture on another thread
    (the NU windows): the pinned allocator polls its events, and polling one that belongs to a capturing stream is an
    error (hipErrorCapturedEvent); those paths copy from pageable memory."""
    if pin and t.device.type == "cpu" and torch.device(device).type == "cuda":
        return (t if t.is_pinned() else t.pin_memory()).to(device, non_blocking=True)
    return t.to(device)


class _PN2Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module):
        B, C, N = x.shape
        model = module._packed()
        ws = module._workspace(B, N, 1)
        starts = upload(draw_fps_starts(B, N, pinned=True), x.device, pin=True)
        xin = x.detach().contiguous().float()
        x0 = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        _lib.call("psg_to_point_major", runtime.ptr(xin), B, C, N, runtime.ptr(x0), runtime.stream())
        ws.plan_build(x0, starts, 1)
        l4 = torch.empty(B, 16, module.L4_CHANNELS, device=x.device, dtype=torch.float32)
        logp = ws.forward(model, 0, x0, l4=l4)
        module._generation += 1
        ctx.module, ctx.model, ctx.ws, ctx.generation = module, model, ws, module._generation
        l4_points = l4.transpose(1, 2)  # [B,512,16] (MSG: [B,1024,16]) like the reference
        ctx.mark_non_differentiable(l4_points)
        return logp, l4_points

    @staticmethod
    def backward(ctx, dlogp, _dl4):
        if ctx.generation != ctx.module._generation:
            raise RuntimeError("the activations of this forward were overwritten by a later forward of the same "
                               "module; only the most recent forward can be back-propagated")
        ws = ctx.ws
        dx0 = ws.backward(ctx.model, 0, dlogp.contiguous().float())
        B, N = ws.batch, ws.n_point
        dx = torch.empty(B, 9, N, device=dlogp.device, dtype=torch.float32)
        _lib.call("psg_to_channel_major", runtime.ptr(dx0), B, 9, N, runtime.ptr(dx), runtime.stream())
        return dx, None


class get_model(nn.Module):
    L4_CHANNELS = 512

    def __init__(self, num_classes):
        super(get_model, self).__init__()
        if num_classes != runtime.NUM_CLASSES:
            raise ValueError("the gfx950 kernels are specialised for the %d S3DIS classes" % runtime.NUM_CLASSES)
        self.sa1 = PointNetSetAbstraction(1024, 0.1, 32, 9 + 3, [32, 32, 64], False)
        self.sa2 = PointNetSetAbstraction(256, 0.2, 32, 64 + 3, [64, 64, 128], False)
        self.sa3 = PointNetSetAbstraction(64, 0.4, 32, 128 + 3, [128, 128, 256], False)
        self.sa4 = PointNetSetAbstraction(16, 0.8, 32, 256 + 3, [256, 256, 512], False)
        self.fp4 = PointNetFeaturePropagation(768, [256, 256])
        self.fp3 = PointNetFeaturePropagation(384, [256, 256])
        self.fp2 = PointNetFeaturePropagation(320, [256, 128])
        self.fp1 = PointNetFeaturePropagation(128, [128, 128, 128])
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)
        self._psg_model = None
        self._psg_key = None
        self._psg_ws = {}
        self._generation = 0

    # ---- libpsg plumbing
    def _packed(self):
        """MFMA-packed, BN-folded weights on the device; rebuilt when a parameter/buffer changed."""
        tensors = list(self.parameters()) + list(self.buffers())
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._psg_model is None or key != self._psg_key:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._psg_model = runtime.PN2Model(runtime.fold_state_dict(sd))
            self._psg_key = key
        return self._psg_model

    def _workspace(self, batch, n_point, n_forward):
        key = (batch, n_point)
        ws = self._psg_ws.get(key)
        if ws is None or ws.max_forwards < n_forward:
            ws = runtime.PN2Workspace(batch, n_point, n_forward)
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
