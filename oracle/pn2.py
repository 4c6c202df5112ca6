"""CPU oracle for the PointNet++ SSG sem-seg attack path (numpy driver over oracle/psg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker / reported CPU baseline.  The product never imports this module.

Restates (paths relative to /root/reference):
  PointNet/models/pointnet_util.py:63-143  sample_and_group (FPS, ball query, gather, concat)
  PointNet/models/pointnet_util.py:181-207 PointNetSetAbstraction.forward
  PointNet/models/pointnet_util.py:281-320 PointNetFeaturePropagation.forward
  PointNet/models/pointnet2_sem_seg.py:22-40 get_model.forward
and the input-gradient (d/d colour) backward that autograd derives from them.
Parity status: pinned by tests/golden/ (generated from the reference, tests/golden/make_golden.py).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# (npoint, radius, nsample) of sa1..sa4, PointNet/models/pointnet2_sem_seg.py:9-12
SA_CFG = ((1024, 0.1, 32), (256, 0.2, 32), (64, 0.4, 32), (16, 0.8, 32))
NUM_CLASSES = 13
BN_EPS = 1e-5

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int32)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpsg_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libpsg_oracle.so missing: run `make -C oracle` or __graft_entry__.build()")
        _LIB = ctypes.CDLL(path)
    return _LIB


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


# ------------------------------------------------------------------ primitives
def square_distance(src, dst):
    src, dst = _c(src), _c(dst)
    out = np.empty((src.shape[0], dst.shape[0]), np.float32)
    lib().orc_square_distance(_fp(src), _fp(dst), src.shape[0], dst.shape[0], _fp(out))
    return out


def fps(xyz, npoint, start):
    xyz = _c(xyz)
    out = np.empty(npoint, np.int32)
    lib().orc_fps(_fp(xyz), xyz.shape[0], npoint, int(start), _ip(out))
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _c(xyz), _c(new_xyz)
    out = np.empty((new_xyz.shape[0], nsample), np.int32)
    r2 = ctypes.c_float(float(np.float32(radius ** 2)))
    lib().orc_ball_query(_fp(xyz), _fp(new_xyz), xyz.shape[0], new_xyz.shape[0], r2, nsample, _ip(out))
    return out


def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _c(xyz1), _c(xyz2)
    idx = np.empty((xyz1.shape[0], 3), np.int32)
    w = np.empty((xyz1.shape[0], 3), np.float32)
    lib().orc_three_nn(_fp(xyz1), _fp(xyz2), xyz1.shape[0], xyz2.shape[0], _ip(idx), _fp(w))
    return idx, w


def linear_fwd(x, w, b, relu):
    x = _c(x)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    lib().orc_linear_fwd(_fp(x), _fp(w), _fp(b), x.shape[0], w.shape[1], w.shape[0], int(relu), _fp(y))
    return y


def linear_bwd(dy, y, w, relu):
    dy = _c(dy)
    dx = np.empty((dy.shape[0], w.shape[1]), np.float32)
    lib().orc_linear_bwd(_fp(dy), _fp(y), _fp(w), dy.shape[0], w.shape[1], w.shape[0], int(relu), _fp(dx))
    return dx


def smooth_knn(a, b, nb):
    a, b = _c(a), _c(b)
    d = np.empty((a.shape[0], nb), np.float32)
    idx = np.empty((a.shape[0], nb), np.int32)
    lib().orc_smooth_knn(_fp(a), _fp(b), a.shape[0], nb, _fp(d), _ip(idx))
    return d, idx


# ------------------------------------------------------------------ weights
def fold_bn(sd):
    """Fold eval-mode BatchNorm into the preceding 1x1 conv (fp64 math, fp32 result).

    sd: {name: ndarray} with the reference's state_dict keys (checkpoint row):
    sa{1-4}.mlp_convs.{i}.weight/bias, sa*.mlp_bns.{i}.weight/bias/running_mean/running_var,
    fp{1-4}.*, conv1, bn1, conv2.  Returns {'sa1': [(W,b),..], ..., 'head': [(W,b),(W,b)]}.
    """
    def fold(conv, bn):
        w = np.asarray(sd[conv + ".weight"], np.float64)
        w = w.reshape(w.shape[0], -1)
        b = np.asarray(sd[conv + ".bias"], np.float64)
        if bn is not None:
            s = np.asarray(sd[bn + ".weight"], np.float64) / np.sqrt(
                np.asarray(sd[bn + ".running_var"], np.float64) + BN_EPS)
            w = w * s[:, None]
            b = (b - np.asarray(sd[bn + ".running_mean"], np.float64)) * s + np.asarray(sd[bn + ".bias"], np.float64)
        return _c(w), _c(b)

    out = {}
    for name, nl in (("sa1", 3), ("sa2", 3), ("sa3", 3), ("sa4", 3), ("fp4", 2), ("fp3", 2), ("fp2", 2), ("fp1", 3)):
        out[name] = [fold("%s.mlp_convs.%d" % (name, i), "%s.mlp_bns.%d" % (name, i)) for i in range(nl)]
    out["head"] = [fold("conv1", "bn1"), fold("conv2", None)]
    return out


# ------------------------------------------------------------------ network
class PN2Oracle:
    """get_model.forward + input-gradient backward for ONE room at a time (rooms are independent)."""

    def __init__(self, state_dict):
        self.layers = fold_bn(state_dict)

    # -- geometry: functions of xyz and the 4 FPS start draws only
    def geometry(self, xyz0, starts):
        xyz = [_c(xyz0)]
        g = {"fps": [], "group": [], "nn_idx": [None] * 4, "nn_w": [None] * 4}
        for lvl, (npoint, radius, nsample) in enumerate(SA_CFG):
            fi = fps(xyz[lvl], npoint, starts[lvl])
            new_xyz = _c(xyz[lvl][fi])
            g["fps"].append(fi)
            g["group"].append(ball_query(radius, nsample, xyz[lvl], new_xyz))
            xyz.append(new_xyz)
        for lvl in range(4):  # fp_{lvl+1}: fine = level lvl, coarse = level lvl+1
            g["nn_idx"][lvl], g["nn_w"][lvl] = three_nn(xyz[lvl], xyz[lvl + 1])
        g["xyz"] = xyz
        return g

    def _mlp_fwd(self, name, x, last_relu=True):
        acts = [x]
        ls = self.layers[name]
        for i, (w, b) in enumerate(ls):
            acts.append(linear_fwd(acts[-1], w, b, last_relu or i + 1 < len(ls)))
        return acts

    def _mlp_bwd(self, name, acts, dy, last_relu=True):
        ls = self.layers[name]
        for i in range(len(ls) - 1, -1, -1):
            dy = linear_bwd(dy, acts[i + 1], ls[i][0], last_relu or i + 1 < len(ls))
        return dy

    def forward(self, x, geom):
        """x: [N,9] point-major room.  Returns (logp [N,13], cache)."""
        L = lib()
        x = _c(x)
        cache = {"geom": geom, "sa": [], "fp": [None] * 4}
        feats = [x]  # l0_points = all 9 channels (pointnet2_sem_seg.py:23)
        for lvl, (npoint, radius, nsample) in enumerate(SA_CFG):
            d = feats[lvl].shape[1]
            rows = np.empty((npoint * nsample, 3 + d), np.float32)
            L.orc_group(_fp(geom["xyz"][lvl]), _fp(feats[lvl]), _fp(geom["xyz"][lvl + 1]),
                        _ip(geom["group"][lvl]), npoint, nsample, d, _fp(rows))
            acts = self._mlp_fwd("sa%d" % (lvl + 1), rows)
            c = acts[-1].shape[1]
            out = np.empty((npoint, c), np.float32)
            arg = np.empty((npoint, c), np.int32)
            L.orc_maxpool_fwd(_fp(acts[-1]), npoint, nsample, c, _fp(out), _ip(arg))
            cache["sa"].append((acts, arg))
            feats.append(out)
        cache["sa_out"] = feats
        up = feats[4]
        fp_out = [None] * 4
        for lvl in (3, 2, 1, 0):  # fp4, fp3, fp2, fp1
            n = geom["xyz"][lvl].shape[0]
            interp = np.empty((n, up.shape[1]), np.float32)
            L.orc_interp_fwd(_fp(up), _ip(geom["nn_idx"][lvl]), _fp(geom["nn_w"][lvl]), n, up.shape[1], _fp(interp))
            cat = interp if lvl == 0 else np.concatenate([feats[lvl], interp], axis=1)
            acts = self._mlp_fwd("fp%d" % (lvl + 1), cat)
            cache["fp"][lvl] = acts
            up = acts[-1]
            fp_out[lvl] = up
        cache["fp_out"] = fp_out
        hacts = self._mlp_fwd("head", up, last_relu=False)
        z = hacts[-1]
        m = z.max(axis=1, keepdims=True)
        logp = (z - m) - np.log(np.exp(z - m).sum(axis=1, keepdims=True, dtype=np.float32))
        cache["head"] = hacts
        cache["logp"] = logp.astype(np.float32)
        return cache["logp"], cache

    def backward_color(self, cache, dlogp):
        """d loss / d colour [N,3] given d loss / d logp [N,13] (log_softmax output)."""
        L = lib()
        geom = cache["geom"]
        dlogp = _c(dlogp)
        p = np.exp(cache["logp"])
        dz = dlogp - p * dlogp.sum(axis=1, keepdims=True, dtype=np.float32)
        dup = self._mlp_bwd("head", cache["head"], dz, last_relu=False)
        feats = cache["sa_out"]
        dfeat = [None] + [np.zeros_like(feats[l]) for l in (1, 2, 3, 4)]
        for lvl in (0, 1, 2, 3):  # fp1, fp2, fp3, fp4
            dcat = self._mlp_bwd("fp%d" % (lvl + 1), cache["fp"][lvl], dup)
            if lvl == 0:
                dint = dcat
            else:
                c1 = feats[lvl].shape[1]
                dfeat[lvl] += dcat[:, :c1]
                dint = _c(dcat[:, c1:])
            s = geom["xyz"][lvl + 1].shape[0]
            if lvl == 3:
                tgt = dfeat[4]
            else:
                tgt = np.zeros((s, dint.shape[1]), np.float32)
            L.orc_interp_bwd(_fp(_c(dint)), _ip(geom["nn_idx"][lvl]), _fp(geom["nn_w"][lvl]),
                             dint.shape[0], s, dint.shape[1], _fp(tgt))
            dup = tgt
        dx0 = np.zeros((geom["xyz"][0].shape[0], 9), np.float32)
        for lvl in (3, 2, 1, 0):  # sa4..sa1
            npoint, _, nsample = SA_CFG[lvl]
            acts, arg = cache["sa"][lvl]
            c = acts[-1].shape[1]
            dh = np.empty((npoint * nsample, c), np.float32)
            L.orc_maxpool_bwd(_fp(dfeat[lvl + 1]), _ip(arg), npoint, nsample, c, _fp(dh))
            drows = self._mlp_bwd("sa%d" % (lvl + 1), acts, dh)
            tgt = dx0 if lvl == 0 else dfeat[lvl]
            L.orc_group_bwd(_fp(drows), _ip(geom["group"][lvl]), npoint, nsample, tgt.shape[1],
                            tgt.shape[0], _fp(tgt))
        cache["dsa"] = dfeat
        return _c(dx0[:, 3:6])


def nll_logp_grad(logp, labels, scale):
    """Gradient w.r.t. logp of scale * sum_i CE(logp_i, y_i), CE applied on top of log-probs
    (nontarget.py:26,34: nn.CrossEntropyLoss on the log_softmax output = a second log_softmax)."""
    m = logp.max(axis=1, keepdims=True)
    lp2 = (logp - m) - np.log(np.exp(logp - m).sum(axis=1, keepdims=True, dtype=np.float32))
    g = np.exp(lp2).astype(np.float32)
    g[np.arange(logp.shape[0]), labels] -= np.float32(1.0)
    cost = -lp2[np.arange(logp.shape[0]), labels].sum(dtype=np.float64) * scale
    return (g * np.float32(scale)).astype(np.float32), float(cost)
