"""Thin object layer over the C ABI: PyTorch supplies device memory and the HIP stream, libpsg.so does
all the computing.  Nothing here falls back to PyTorch ops for the hot path.
"""
import ctypes

import numpy as np
import torch

from . import _lib

SA_NPOINT = (1024, 256, 64, 16)   # PointNet/models/pointnet2_sem_seg.py:9-12 (reference)
SA_RADIUS = (0.1, 0.2, 0.4, 0.8)
NSAMPLE = 32
NUM_CLASSES = 13
ARCH_SSG, ARCH_MSG = 0, 1          # PSG_PN2_ARCH_* of include/psg.h
ARCH_LAYERS = {ARCH_SSG: 23, ARCH_MSG: 35}
MSG_NSAMPLE = (16, 32)             # PointNet/models/pointnet2_sem_seg_msg.py:10-13 (reference)
ACT_POINTS = (1024, 256, 64, 16, 64, 256, 1024)
ACT_SHAPES = ((1024, 64), (256, 128), (64, 256), (16, 512), (64, 256), (256, 256), (1024, 128))

_ctx = {}
_hip = None


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PsgError("%s must be a tensor on the MI355X device (got %r); there is no CPU path" % (
            name, getattr(t, "device", type(t))))
    if dtype is not None and t.dtype != dtype:
        raise _lib.PsgError("%s must have dtype %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.PsgError("%s must be contiguous" % name)
    return t


def context(device=None):
    """One psg_ctx per device index, created on first use."""
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ctx:
        lib = _lib.load()
        torch.cuda.init()
        h = ctypes.c_void_p()
        _lib.check(lib.psg_ctx_create(idx, ctypes.byref(h)), "psg_ctx_create")
        _ctx[idx] = h
    return _ctx[idx]


def _hip_memcpy_d2d(dst_ptr, src_ptr, nbytes):
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        _hip.hipMemcpyAsync.restype = ctypes.c_int
    rc = _hip.hipMemcpyAsync(dst_ptr, src_ptr, nbytes, 3, stream())
    if rc != 0:
        raise _lib.PsgError("hipMemcpyAsync failed with %d" % rc)


def fold_state_dict(sd, eps=1e-5, msg=False):
    """Eval-mode BatchNorm folded into the preceding 1x1 conv, in the layer order of
    psg_pn2_model_create.  sd: mapping name -> tensor/ndarray with the reference's state_dict keys
    (sa{1-4}.mlp_convs.N.weight ..., fp{1-4}..., conv1, bn1, conv2).  fp64 math, fp32 result."""
    def arr(k):
        v = sd[k]
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        return np.asarray(v, np.float64)

    def fold(conv, bn):
        w = arr(conv + ".weight")
        w = w.reshape(w.shape[0], -1)
        b = arr(conv + ".bias")
        if bn is not None:
            s = arr(bn + ".weight") / np.sqrt(arr(bn + ".running_var") + eps)
            w = w * s[:, None]
            b = (b - arr(bn + ".running_mean")) * s + arr(bn + ".bias")
        return np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)

    out = []
    if msg:   # sa{l}.conv_blocks.{scale}.{j} / bn_blocks (pointnet_util.py:216-227 of the reference)
        for l in range(1, 5):
            for i in range(2):
                for j in range(3):
                    out.append(fold("sa%d.conv_blocks.%d.%d" % (l, i, j), "sa%d.bn_blocks.%d.%d" % (l, i, j)))
    else:
        for name in ("sa1", "sa2", "sa3", "sa4"):
            for i in range(3):
                out.append(fold("%s.mlp_convs.%d" % (name, i), "%s.mlp_bns.%d" % (name, i)))
    for name, nl in (("fp4", 2), ("fp3", 2), ("fp2", 2), ("fp1", 3)):
        for i in range(nl):
            out.append(fold("%s.mlp_convs.%d" % (name, i), "%s.mlp_bns.%d" % (name, i)))
    out.append(fold("conv1", "bn1"))
    out.append(fold("conv2", None))
    return out


class PN2Model:
    """Device-resident MFMA-packed weights of get_model (psg_pn2_model)."""

    def __init__(self, folded, device=None, arch=ARCH_SSG):
        n = ARCH_LAYERS[arch]
        if len(folded) != n:
            raise _lib.PsgError("expected %d folded layers, got %d" % (n, len(folded)))
        self.ctx = context(device)
        self.arch = arch
        lib = _lib.load()
        ws = (ctypes.c_void_p * n)(*[w.ctypes.data_as(ctypes.c_void_p) for w, _ in folded])
        bs = (ctypes.c_void_p * n)(*[b.ctypes.data_as(ctypes.c_void_p) for _, b in folded])
        self._keep = folded
        self.handle = ctypes.c_void_p()
        _lib.check(lib.psg_pn2_model_create_arch(self.ctx, arch, ws, bs, n, ctypes.byref(self.handle)),
                   "psg_pn2_model_create_arch")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().psg_pn2_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class PN2Workspace:
    """Geometry plan + activations + gradient buffers for a batch (psg_pn2_ws)."""

    def __init__(self, batch, n_point, max_forwards, device=None, arch=ARCH_SSG):
        self.ctx = context(device)
        self.batch, self.n_point, self.max_forwards, self.arch = batch, n_point, max_forwards, arch
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().psg_pn2_ws_create_arch(self.ctx, arch, batch, n_point, max_forwards,
                                                      ctypes.byref(self.handle)), "psg_pn2_ws_create_arch")
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().psg_pn2_ws_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @property
    def nbytes(self):
        return _lib.load().psg_pn2_ws_bytes(self.handle)

    def plan_build(self, x0, starts, n_forward):
        require_cuda(x0, "x0", torch.float32)
        require_cuda(starts, "starts", torch.int32)
        assert x0.shape == (self.batch, self.n_point, 9) and starts.numel() == n_forward * 4 * self.batch
        _lib.call("psg_pn2_plan_build", self.handle, ptr(x0), ptr(starts), n_forward, stream())

    def forward(self, model, slot, x0, logp=None, l4=None, lean=False):
        """lean=True: the forward the fused attack loops run (no module outputs kept for activation(); l4 not available)."""
        require_cuda(x0, "x0", torch.float32)
        if logp is None:
            logp = torch.empty(self.batch, self.n_point, NUM_CLASSES, device=x0.device, dtype=torch.float32)
        if lean:
            assert l4 is None
            _lib.call("psg_pn2_forward_lean", model.handle, self.handle, slot, ptr(x0), ptr(logp), stream())
        else:
            _lib.call("psg_pn2_forward", model.handle, self.handle, slot, ptr(x0), ptr(logp), ptr(l4), stream())
        return logp

    def backward(self, model, slot, dlogp, dx0=None, colour_only=False):
        """colour_only=True: the backward of the fused attack loops - channels 3..5 of dx0 only (the rest is zero here)."""
        require_cuda(dlogp, "dlogp", torch.float32)
        if dx0 is None:
            dx0 = (torch.zeros if colour_only else torch.empty)(self.batch, self.n_point, 9, device=dlogp.device, dtype=torch.float32)
        _lib.call("psg_pn2_backward_colour" if colour_only else "psg_pn2_backward", model.handle, self.handle, slot, ptr(dlogp),
                  ptr(dx0), stream())
        return dx0

    def backward_pgd(self, model, slot, dlogp, x0, ori, alpha, eps, mask=None, descent=False, last=False):
        """The colour-only backward with the NB / tar_NB update applied to x0 in place by the gradient's last gather (what
        psg_pn2_nb_attack runs per iteration; nontarget.py:37-39, target.py:41-43)."""
        require_cuda(dlogp, "dlogp", torch.float32)
        require_cuda(x0, "x0", torch.float32)
        require_cuda(ori, "ori", torch.float32)
        if mask is not None:
            require_cuda(mask, "mask", torch.uint8)
        assert x0.shape == (self.batch, self.n_point, 9) and ori.shape == (self.batch, self.n_point, 3) and x0.is_contiguous() and ori.is_contiguous()
        _lib.call("psg_pn2_backward_colour_pgd", model.handle, self.handle, slot, ptr(dlogp), ptr(x0), ptr(ori), ptr(mask),
                  float(alpha), float(eps), -1.0 if descent else 1.0, 1 if last else 0, stream())
        return x0

    def nb_attack(self, model, images, labels, starts, eps, alpha, iters, mask=None, target=None, out=None):
        require_cuda(images, "images", torch.float32)
        require_cuda(starts, "starts", torch.int32)
        if labels is not None:
            require_cuda(labels, "labels", torch.int32)
        if mask is not None:
            require_cuda(mask, "mask", torch.uint8)
        if out is None:
            out = torch.empty_like(images)
        _lib.call("psg_pn2_nb_attack", model.handle, self.handle, ptr(images), ptr(labels), ptr(starts), ptr(mask),
                  float(eps), float(alpha), int(iters), 0 if target is None else 1, 0 if target is None else int(target),
                  ptr(out), stream())
        return out

    PROF_TAGS = ("sa1_fwd", "sa2_fwd", "sa3_fwd", "sa4_fwd", "fp1_head_fwd", "fp2_fwd", "fp3_fwd", "fp4_fwd",
                 "fp1_head_bwd", "fp2_bwd", "fp3_bwd", "fp4_bwd", "sa1_bwd", "sa2_bwd", "sa3_bwd", "sa4_bwd",
                 "fps", "ball_query", "three_nn", "gather", "ce_grad", "pgd_step", "dx0_gather", "pw_fwd", "pw_bwd")

    def prof_enable(self, on=True):
        _lib.call("psg_pn2_prof_enable", self.handle, 1 if on else 0)

    def prof_read(self):
        """{kernel tag: (total ms, launches)} measured with HIP events on the launch stream."""
        n = len(self.PROF_TAGS)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int * n)()
        _lib.call("psg_pn2_prof_read", self.handle, n, ms, cnt)
        return {t: (ms[i], cnt[i]) for i, t in enumerate(self.PROF_TAGS) if cnt[i]}

    # ---- read-back helpers (parity tests)
    def plan_tensor(self, what, level, forward, room):
        n_l = (self.n_point,) + SA_NPOINT
        k0 = MSG_NSAMPLE[0] if self.arch == ARCH_MSG else NSAMPLE
        shape, dt = {0: ((SA_NPOINT[level],), torch.int32), 1: ((SA_NPOINT[level], k0), torch.int32),
                     2: ((n_l[level], 3), torch.int32), 3: ((n_l[level], 3), torch.float32),
                     4: ((SA_NPOINT[level], 3), torch.float32),
                     5: ((SA_NPOINT[level], MSG_NSAMPLE[1]), torch.int32)}[what]
        src = _lib.load().psg_pn2_plan_ptr(self.handle, what, level, forward, room)
        if not src:
            raise _lib.PsgError("psg_pn2_plan_ptr: bad slice")
        out = torch.empty(shape, dtype=dt, device=self.device)
        _hip_memcpy_d2d(out.data_ptr(), src, out.numel() * 4)
        return out

    def activation(self, which):
        n, c = ACT_POINTS[which], _lib.load().psg_pn2_activation_channels(self.handle, which)
        src = _lib.load().psg_pn2_activation_ptr(self.handle, which)
        out = torch.empty(self.batch, n, c, dtype=torch.float32, device=self.device)
        _hip_memcpy_d2d(out.data_ptr(), src, out.numel() * 4)
        return out


# ---- unit ops -----------------------------------------------------------------------------------
def square_distance(src, dst):
    require_cuda(src, "src", torch.float32)
    require_cuda(dst, "dst", torch.float32)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = torch.empty(B, N, M, dtype=torch.float32, device=src.device)
    _lib.call("psg_square_distance", context(src.device), ptr(src), ptr(dst), B, N, M, ptr(out), stream())
    return out


def fps(xyz, npoint, start):
    """xyz [P,N,3] float32 cuda, start [P] int32 -> idx [P,npoint] int32 (pointnet_util.py:63-84)."""
    require_cuda(xyz, "xyz", torch.float32)
    require_cuda(start, "start", torch.int32)
    P, N, _ = xyz.shape
    out = torch.empty(P, npoint, dtype=torch.int32, device=xyz.device)
    _lib.call("psg_fps", context(xyz.device), ptr(xyz), P, P, N, npoint, ptr(start), ptr(out), stream())
    return out


def gather_points(points, idx):
    require_cuda(points, "points", torch.float32)
    require_cuda(idx, "idx", torch.int32)
    P, N, C = points.shape
    S = idx.shape[1]
    out = torch.empty(P, S, C, dtype=torch.float32, device=points.device)
    _lib.call("psg_gather_points", context(points.device), ptr(points), P, P, N, C, ptr(idx), S, ptr(out), stream())
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    require_cuda(xyz, "xyz", torch.float32)
    require_cuda(new_xyz, "new_xyz", torch.float32)
    P, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty(P, S, nsample, dtype=torch.int32, device=xyz.device)
    r2 = float(np.float32(radius ** 2))
    _lib.call("psg_ball_query", context(xyz.device), ptr(xyz), P, ptr(new_xyz), P, N, S, r2, nsample, ptr(out), stream())
    return out


def three_nn(xyz1, xyz2):
    require_cuda(xyz1, "xyz1", torch.float32)
    require_cuda(xyz2, "xyz2", torch.float32)
    P, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = torch.empty(P, N, 3, dtype=torch.int32, device=xyz1.device)
    w = torch.empty(P, N, 3, dtype=torch.float32, device=xyz1.device)
    _lib.call("psg_three_nn", context(xyz1.device), ptr(xyz1), P, ptr(xyz2), P, N, S, ptr(idx), ptr(w), stream())
    return idx, w


def seg_stats(logp, labels, n_cls=NUM_CLASSES, counters=None):
    """Accumulate (seen, inter, union) int64 [3, n_cls]; returns (counters, pred)."""
    require_cuda(logp, "logp", torch.float32)
    require_cuda(labels, "labels", torch.int32)
    rows = labels.numel()
    if counters is None:
        counters = torch.zeros(3, n_cls, dtype=torch.int64, device=logp.device)
    pred = torch.empty(rows, dtype=torch.int32, device=logp.device)
    _lib.call("psg_seg_stats", ptr(logp), ptr(labels), rows, n_cls, ptr(counters), ptr(pred), stream())
    return counters, pred.view(labels.shape)


# ---- ResGCN (dense DeepGCN) ------------------------------------------------------------------------
def gcn_tensor_list(sd, n_blocks):
    """state_dict (reference key names, ResGCN/sem_seg_dense/architecture.py) -> the flat fp32 tensor list of
    psg_gcn_model_create."""
    def a(k):
        v = sd[k]
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.ascontiguousarray(v, np.float32)
        return v.reshape(v.shape[0], -1) if v.ndim > 1 else v

    out = []
    for e in range(n_blocks):
        base = "head.gconv.nn" if e == 0 else "backbone.%d.body.gconv.nn" % (e - 1)
        out += [a(base + ".0.weight"), a(base + ".0.bias")]
        out += [a(base + ".2." + k) for k in ("weight", "bias", "running_mean", "running_var")]
    for conv, bn in (("fusion_block.0", "fusion_block.2"), ("prediction.0.0", "prediction.0.2"),
                     ("prediction.1.0", "prediction.1.2")):
        out += [a(conv + ".weight"), a(conv + ".bias")] + [a(bn + "." + k) for k in ("weight", "bias", "running_mean", "running_var")]
    out += [a("prediction.3.0.weight"), a("prediction.3.0.bias")]
    return out


GCN_BLOCK_RES, GCN_BLOCK_PLAIN, GCN_BLOCK_DENSE = 0, 1, 2   # PSG_GCN_BLOCK_* / PSG_GCN_CONV_* of include/psg.h
GCN_CONV_EDGE, GCN_CONV_MR = 0, 1


class GCNModel:
    def __init__(self, sd, n_blocks, device=None, block=GCN_BLOCK_RES, conv=GCN_CONV_EDGE):
        self.ctx = context(device)
        self.n_blocks, self.block, self.conv = n_blocks, block, conv
        self._keep = gcn_tensor_list(sd, n_blocks)
        arr = (ctypes.c_void_p * len(self._keep))(*[t.ctypes.data_as(ctypes.c_void_p) for t in self._keep])
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().psg_gcn_model_create_cfg(self.ctx, arr, len(self._keep), n_blocks, block, conv,
                                                        ctypes.byref(self.handle)), "psg_gcn_model_create_cfg")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().psg_gcn_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class GCNWorkspace:
    def __init__(self, batch, n_point, n_blocks, device=None, block=GCN_BLOCK_RES, conv=GCN_CONV_EDGE):
        self.ctx = context(device)
        self.batch, self.n_point, self.n_blocks = batch, n_point, n_blocks
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().psg_gcn_ws_create_cfg(self.ctx, batch, n_point, n_blocks, block, conv,
                                                     ctypes.byref(self.handle)), "psg_gcn_ws_create_cfg")
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().psg_gcn_ws_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def knn(self, x, dilation):
        require_cuda(x, "x", torch.float32)
        B, N, C = x.shape
        out = torch.empty(B, N, 16, dtype=torch.int32, device=x.device)
        _lib.call("psg_gcn_knn", self.handle, ptr(x), C, int(dilation), ptr(out), stream())
        return out

    def knn_stats(self, reset=True):
        """Counters of the bf16-prefilter kNN kernel (workspaces created under PSG_GCN_KNN_STATS=1): dict of tiles,
        exact_tiles, rows, finalists, cuts, entries, why (reasons of the exact-path tiles)."""
        buf = (ctypes.c_ulonglong * 8)()
        _lib.call("psg_gcn_knn_stats", self.handle, buf, 1 if reset else 0)
        out = dict(zip(("tiles", "exact_tiles", "rows", "finalists", "cuts", "entries"), [int(v) for v in buf[:6]]))
        # why tiles went to the exact path (a tile can count under several): no threshold from the sample, a row above its
        # buffer, a row short of KK entries, a row short of KK keys below the cut's key, more than 256 finalists, bound broken
        a, b = int(buf[6]), int(buf[7])
        out["why"] = {"cut": a & 0xFFFF, "overflow": (a >> 16) & 0xFFFF, "underflow": (a >> 32) & 0xFFFF,
                      "short": b & 0xFFFF, "finalists": (b >> 16) & 0xFFFF, "bound": (b >> 32) & 0xFFFF}
        return out

    PROF_TAGS = ("knn_fused", "knn_other", "vertex_gemm", "edge_max", "fusion_prediction", "backward")

    def prof_enable(self, on=True):
        _lib.call("psg_gcn_prof_enable", self.handle, 1 if on else 0)

    def prof_read(self):
        """{tag: (total ms, launches, algorithmic FLOPs)} measured with HIP events on the launch stream."""
        n = len(self.PROF_TAGS)
        ms, cnt, fl = (ctypes.c_double * n)(), (ctypes.c_int * n)(), (ctypes.c_double * n)()
        _lib.call("psg_gcn_prof_read", self.handle, n, ms, cnt, fl)
        return {t: (ms[i], cnt[i], fl[i]) for i, t in enumerate(self.PROF_TAGS) if cnt[i]}

    def set_graphs(self, nbr):
        if nbr is not None:
            require_cuda(nbr, "nbr", torch.int32)
        _lib.call("psg_gcn_set_graphs", self.handle, ptr(nbr), stream())

    def forward(self, model, x0, logits=None):
        require_cuda(x0, "x0", torch.float32)
        if logits is None:
            logits = torch.empty(self.batch, self.n_point, NUM_CLASSES, device=x0.device, dtype=torch.float32)
        _lib.call("psg_gcn_forward", model.handle, self.handle, ptr(x0), ptr(logits), stream())
        return logits

    def backward(self, model, dlogits, dx0=None):
        require_cuda(dlogits, "dlogits", torch.float32)
        if dx0 is None:
            dx0 = torch.empty(self.batch, self.n_point, 9, device=dlogits.device, dtype=torch.float32)
        _lib.call("psg_gcn_backward", model.handle, self.handle, ptr(dlogits), ptr(dx0), stream())
        return dx0

    def nb_attack(self, model, images, labels, eps, alpha, iters, out=None):
        require_cuda(images, "images", torch.float32)
        require_cuda(labels, "labels", torch.int32)
        if out is None:
            out = torch.empty_like(images)
        _lib.call("psg_gcn_nb_attack", model.handle, self.handle, ptr(images), ptr(labels), float(eps), float(alpha),
                  int(iters), ptr(out), stream())
        return out

    def edges(self, block):
        src = _lib.load().psg_gcn_edge_ptr(self.handle, block)
        out = torch.empty(self.batch, self.n_point, 16, dtype=torch.int32, device=self.device)
        _hip_memcpy_d2d(out.data_ptr(), src, out.numel() * 4)
        return out

    def feats(self):
        src = _lib.load().psg_gcn_feats_ptr(self.handle)
        out = torch.empty(self.batch, self.n_point, 64 * self.n_blocks, dtype=torch.float32, device=self.device)
        _hip_memcpy_d2d(out.data_ptr(), src, out.numel() * 4)
        return out
