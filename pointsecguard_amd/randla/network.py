"""RandLA-Net on the MI355X: thin object layer over the psg_rla_* C ABI (csrc/psg_randla_net.hip).

Restates Network.inference of RandLA-Net/RandLANet.py:150-190 (eval mode) and the BIM colour attack of
ares/ares/attack/bim.py of the reference; one cloud per workspace like the reference's validation batches.  PARITY
UNPINNED: the reference's TensorFlow-1 graph cannot run here (see oracle/randla_net.py).  `params` is a mapping with
the keys of pointsecguard_amd.synthetic.randla_layer_specs: name.weight [cout, cin], name.bias, name.bn.{gamma, beta,
mean, var} (a TF checkpoint maps onto it by transposing each [1, 1, cin, cout] kernel).
"""
import ctypes

import numpy as np
import torch

from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.synthetic import randla_layer_specs

NUM_CLASSES = 13


def tensor_list(params):
    out = []
    for name, cin, cout, bn in randla_layer_specs():
        w = np.ascontiguousarray(params[name + ".weight"], np.float32)
        if w.shape != (cout, cin):
            raise ValueError("%s.weight has shape %s, expected %s" % (name, w.shape, (cout, cin)))
        row = [w, params.get(name + ".bias")]
        row += [params.get(name + ".bn." + k) for k in ("gamma", "beta", "mean", "var")] if bn else [None] * 4
        out += [None if t is None else np.ascontiguousarray(t, np.float32) for t in row]
    return out


class RandLAModel:
    def __init__(self, params, device=None):
        self.ctx = runtime.context(device)
        self._keep = tensor_list(params)
        arr = (ctypes.c_void_p * len(self._keep))(*[None if t is None else t.ctypes.data_as(ctypes.c_void_p) for t in self._keep])
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().psg_rla_model_create(self.ctx, arr, len(self._keep), ctypes.byref(self.handle)),
                   "psg_rla_model_create")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().psg_rla_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class RandLAWorkspace:
    """Activations, index pyramid and attack state of `batch` clouds of n_points each (batch > 1: the clouds are
    attacked together, one kernel launch serving all of them; tensors are then [batch * n_points, ..], cloud-major)."""

    def __init__(self, n_points, device=None, batch=1):
        self.ctx = runtime.context(device)
        self.points_per_cloud, self.batch = n_points, batch
        self.n_points = n_points * batch            # rows of every per-point tensor
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().psg_rla_ws_create_batch(self.ctx, n_points, batch, ctypes.byref(self.handle)),
                   "psg_rla_ws_create_batch")
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().psg_rla_ws_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def prof_enable(self, on=True):
        _lib.call("psg_rla_prof_enable", self.handle, 1 if on else 0)

    def prof_read(self):
        """(total ms, launches, algorithmic FLOPs) of the GEMM launches since prof_enable (HIP events on the launch stream)."""
        ms, cnt, fl = (ctypes.c_double * 1)(), (ctypes.c_int * 1)(), (ctypes.c_double * 1)()
        _lib.call("psg_rla_prof_read", self.handle, 1, ms, cnt, fl)
        return ms[0], cnt[0], fl[0]

    PROF_KERNELS = ("gemm_rows_kernel<2,2,.,.,1,1> (64x64 tiles)", "gemm_rows_kernel<2,2> (128x128 tiles)",
                    "direct_gemm_kernel<.,2> (33-64 channels, levels 0-1)", "direct_gemm_kernel<.,1> / skinny (<= 32 channels)", "att_pool_split_fwd_kernel",
                    "att_pool_split_bwd_kernel", "lfa16_fwd_kernel", "lfa16_bwd_kernel")

    def prof_read_kernels(self):
        """{kernel: (total ms, launches, algorithmic FLOPs, algorithmic bytes)} of the GEMM and attentive-pooling launches
        since prof_enable."""
        n = len(self.PROF_KERNELS)
        ms, cnt = (ctypes.c_double * n)(), (ctypes.c_int * n)()
        fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)()
        _lib.call("psg_rla_prof_read_kernels", self.handle, n, ms, cnt, fl, by)
        return {k: (ms[i], cnt[i], fl[i], by[i]) for i, k in enumerate(self.PROF_KERNELS) if cnt[i]}

    @property
    def nbytes(self):
        return _lib.load().psg_rla_ws_bytes(self.handle)

    def set_cloud(self, xyz):
        runtime.require_cuda(xyz, "xyz", torch.float32)
        assert tuple(xyz.shape) == (self.n_points, 3)
        _lib.call("psg_rla_set_cloud", self.handle, runtime.ptr(xyz), runtime.stream())

    def index(self, what, level):
        n = self.n_points                          # (row numbers of the cloud-major buffers of that level)
        for r in (4, 4, 4, 4, 2)[:level]:
            n //= r
        shape = (n, 16) if what == 0 else (n,)
        src = _lib.load().psg_rla_index_ptr(self.handle, what, level)
        out = torch.empty(shape, dtype=torch.int32, device=self.device)
        runtime._hip_memcpy_d2d(out.data_ptr(), src, out.numel() * 4)
        return out

    def forward(self, model, features):
        runtime.require_cuda(features, "features", torch.float32)
        assert tuple(features.shape) == (self.n_points, 6)
        logits = torch.empty(self.n_points, NUM_CLASSES, dtype=torch.float32, device=features.device)
        _lib.call("psg_rla_forward", model.handle, self.handle, runtime.ptr(features), runtime.ptr(logits), runtime.stream())
        return logits

    def backward(self, model, dlogits):
        runtime.require_cuda(dlogits, "dlogits", torch.float32)
        dfeat = torch.empty(self.n_points, 6, dtype=torch.float32, device=dlogits.device)
        _lib.call("psg_rla_backward", model.handle, self.handle, runtime.ptr(dlogits), runtime.ptr(dfeat), runtime.stream())
        return dfeat

    def bim_attack(self, model, features, labels, eps, alpha, iters, metric="l_inf"):
        runtime.require_cuda(features, "features", torch.float32)
        runtime.require_cuda(labels, "labels", torch.int32)
        out = torch.empty_like(features)
        _lib.call("psg_rla_bim_attack", model.handle, self.handle, runtime.ptr(features), runtime.ptr(labels), float(eps),
                  float(alpha), int(iters), 1 if metric == "l_2" else 0, runtime.ptr(out), runtime.stream())
        return out


def colper_grad(logits, labels):
    """The BIM attack's loss and its gradient w.r.t. the logits (device)."""
    runtime.require_cuda(logits, "logits", torch.float32)
    runtime.require_cuda(labels, "labels", torch.int32)
    d = torch.empty_like(logits)
    loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
    _lib.call("psg_rla_colper_grad", runtime.ptr(logits), runtime.ptr(labels), logits.shape[0], runtime.ptr(d), runtime.ptr(loss),
              runtime.stream())
    return loss, d
