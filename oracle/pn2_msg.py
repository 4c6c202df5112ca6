"""CPU oracle for the PointNet++ MSG (multi-scale grouping) sem-seg network.

TEST INFRASTRUCTURE ONLY: imported by tests/ as the checker.  The product never imports this module.

Restates (paths relative to /root/reference):
  PointNet/models/pointnet_util.py:210-267  PointNetSetAbstractionMsg (one FPS per level, one ball query +
                                            3-layer shared MLP + max-pool per radius, channel concat;
                                            grouped rows are [features, rel_xyz], :249-252)
  PointNet/models/pointnet2_sem_seg_msg.py:7-42  get_model (sa1-4 MSG, fp4-1, conv1/bn1, conv2, log_softmax)
on top of the primitives of oracle/pn2.py (FPS, ball query, 3-NN, 1x1 conv, max-pool, interpolation) and the
input-gradient (d/d colour) backward that autograd derives from them.
Parity status: pinned by tests/golden/pn2msg_room.npz (generated from the reference by
tests/golden/make_golden_msg.py).
"""
import numpy as np

from . import pn2
from .pn2 import _c, _fp, _ip

# (npoint, radius_list, nsample_list) of sa1..sa4, PointNet/models/pointnet2_sem_seg_msg.py:10-13
SA_CFG = ((1024, (0.05, 0.1), (16, 32)), (256, (0.1, 0.2), (16, 32)), (64, (0.2, 0.4), (16, 32)),
          (16, (0.4, 0.8), (16, 32)))
NUM_CLASSES = 13


def fold_bn(sd):
    """Eval-mode BatchNorm folded into the preceding 1x1 conv; keys sa{l}.conv_blocks.{i}.{j} /
    sa{l}.bn_blocks.{i}.{j} (pointnet_util.py:216-227), fp{l}.mlp_convs.{i}, conv1/bn1/conv2."""
    def fold(conv, bn):
        w = np.asarray(sd[conv + ".weight"], np.float64)
        w = w.reshape(w.shape[0], -1)
        b = np.asarray(sd[conv + ".bias"], np.float64)
        if bn is not None:
            s = np.asarray(sd[bn + ".weight"], np.float64) / np.sqrt(
                np.asarray(sd[bn + ".running_var"], np.float64) + pn2.BN_EPS)
            w = w * s[:, None]
            b = (b - np.asarray(sd[bn + ".running_mean"], np.float64)) * s + np.asarray(sd[bn + ".bias"], np.float64)
        return _c(w), _c(b)

    out = {}
    for l in range(1, 5):
        for i in range(2):
            out["sa%d.%d" % (l, i)] = [fold("sa%d.conv_blocks.%d.%d" % (l, i, j), "sa%d.bn_blocks.%d.%d" % (l, i, j))
                                       for j in range(3)]
    for name, nl in (("fp4", 2), ("fp3", 2), ("fp2", 2), ("fp1", 3)):
        out[name] = [fold("%s.mlp_convs.%d" % (name, i), "%s.mlp_bns.%d" % (name, i)) for i in range(nl)]
    out["head"] = [fold("conv1", "bn1"), fold("conv2", None)]
    return out


class PN2MsgOracle(pn2.PN2Oracle):
    """get_model.forward + input-gradient backward of the MSG network for ONE room at a time."""

    def __init__(self, state_dict):
        self.layers = fold_bn(state_dict)

    def geometry(self, xyz0, starts):
        xyz = [_c(xyz0)]
        g = {"fps": [], "group": [], "nn_idx": [None] * 4, "nn_w": [None] * 4}
        for lvl, (npoint, radii, nsamples) in enumerate(SA_CFG):
            fi = pn2.fps(xyz[lvl], npoint, starts[lvl])
            new_xyz = _c(xyz[lvl][fi])
            g["fps"].append(fi)
            g["group"].append([pn2.ball_query(r, k, xyz[lvl], new_xyz) for r, k in zip(radii, nsamples)])
            xyz.append(new_xyz)
        for lvl in range(4):
            g["nn_idx"][lvl], g["nn_w"][lvl] = pn2.three_nn(xyz[lvl], xyz[lvl + 1])
        g["xyz"] = xyz
        return g

    def forward(self, x, geom):
        L = pn2.lib()
        x = _c(x)
        cache = {"geom": geom, "sa": [], "fp": [None] * 4}
        feats = [x]   # l0_points = all 9 channels (pointnet2_sem_seg_msg.py:24)
        for lvl, (npoint, radii, nsamples) in enumerate(SA_CFG):
            d = feats[lvl].shape[1]
            outs, per_scale = [], []
            for i, nsample in enumerate(nsamples):
                rows = np.empty((npoint * nsample, 3 + d), np.float32)
                L.orc_group(_fp(geom["xyz"][lvl]), _fp(feats[lvl]), _fp(geom["xyz"][lvl + 1]),
                            _ip(geom["group"][lvl][i]), npoint, nsample, d, _fp(rows))
                rows = _c(np.concatenate([rows[:, 3:], rows[:, :3]], axis=1))   # [features, rel_xyz] (:249-252)
                acts = self._mlp_fwd("sa%d.%d" % (lvl + 1, i), rows)
                c = acts[-1].shape[1]
                out = np.empty((npoint, c), np.float32)
                arg = np.empty((npoint, c), np.int32)
                L.orc_maxpool_fwd(_fp(acts[-1]), npoint, nsample, c, _fp(out), _ip(arg))
                per_scale.append((acts, arg))
                outs.append(out)
            cache["sa"].append(per_scale)
            feats.append(_c(np.concatenate(outs, axis=1)))
        cache["sa_out"] = feats
        up = feats[4]
        fp_out = [None] * 4
        for lvl in (3, 2, 1, 0):
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
        L = pn2.lib()
        geom = cache["geom"]
        dlogp = _c(dlogp)
        p = np.exp(cache["logp"])
        dz = dlogp - p * dlogp.sum(axis=1, keepdims=True, dtype=np.float32)
        dup = self._mlp_bwd("head", cache["head"], dz, last_relu=False)
        feats = cache["sa_out"]
        dfeat = [None] + [np.zeros_like(feats[l]) for l in (1, 2, 3, 4)]
        for lvl in (0, 1, 2, 3):
            dcat = self._mlp_bwd("fp%d" % (lvl + 1), cache["fp"][lvl], dup)
            if lvl == 0:
                dint = dcat
            else:
                c1 = feats[lvl].shape[1]
                dfeat[lvl] += dcat[:, :c1]
                dint = _c(dcat[:, c1:])
            s = geom["xyz"][lvl + 1].shape[0]
            tgt = dfeat[4] if lvl == 3 else np.zeros((s, dint.shape[1]), np.float32)
            L.orc_interp_bwd(_fp(_c(dint)), _ip(geom["nn_idx"][lvl]), _fp(geom["nn_w"][lvl]),
                             dint.shape[0], s, dint.shape[1], _fp(tgt))
            dup = tgt
        dx0 = np.zeros((geom["xyz"][0].shape[0], 9), np.float32)
        for lvl in (3, 2, 1, 0):
            npoint, _, nsamples = SA_CFG[lvl]
            tgt = dx0 if lvl == 0 else dfeat[lvl]
            c_off = 0
            for i, nsample in enumerate(nsamples):
                acts, arg = cache["sa"][lvl][i]
                c = acts[-1].shape[1]
                dh = np.empty((npoint * nsample, c), np.float32)
                L.orc_maxpool_bwd(_fp(_c(dfeat[lvl + 1][:, c_off:c_off + c])), _ip(arg), npoint, nsample, c, _fp(dh))
                drows = self._mlp_bwd("sa%d.%d" % (lvl + 1, i), acts, dh)
                d = tgt.shape[1]
                drows = _c(np.concatenate([drows[:, d:], drows[:, :d]], axis=1))   # back to [rel_xyz, features]
                L.orc_group_bwd(_fp(drows), _ip(geom["group"][lvl][i]), npoint, nsample, d, tgt.shape[0], _fp(tgt))
                c_off += c
        cache["dsa"] = dfeat
        return _c(dx0[:, 3:6])
