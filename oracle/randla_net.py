"""CPU oracle for RandLA-Net's inference graph and its colour-attack gradient.

TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED**: the reference is a TensorFlow-1 graph (RandLA-Net/RandLANet.py) that
can neither be imported nor run in this environment (no TensorFlow), ships no checkpoint and no golden vectors, so this
restatement is anchored on a reading of its source only:
  RandLANet.py:150-190   inference (fc0, 5 dilated residual encoder blocks + random-sample max-pool, decoder_0,
                         5 nearest-interpolation decoder layers, fc1 / fc2 / fc)
  RandLANet.py:323-410   dilated_res_block, building_block, relative_pos_encoding, random_sample,
                         nearest_interpolation, gather_neighbour, att_pooling
  helper_tf_util.py:115-170, 173-250   conv2d / conv2d_transpose: 1x1 conv + bias -> batch_normalization(eps 1e-6)
                         -> leaky_relu(0.2) (when an activation is requested)
  ares/ares/attack/bim.py:66-98, 110-116  the attacked input is the colour half of the features; loss = sum over
                         points of max(0, max_{k != y} logit_k* - logit_y) with the reference's masked max (the masked
                         logit of the true class is 0, so a 0 takes part in the max); l_inf / l_2 update rules
A float32 torch restatement (autograd provides the gradient): what the HIP path is compared with.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-6
SLOPE = 0.2


class RandLAOracle:
    def __init__(self, params, d_out=(16, 64, 128, 256, 512), dtype=torch.float32):
        self.p = {k: torch.from_numpy(np.asarray(v, np.float32)).to(dtype) for k, v in params.items()}
        self.d_out = d_out
        self.dtype = dtype

    def conv(self, x, name, act=True):
        """x [..., cin] -> [..., cout]: 1x1 conv (+ bias) -> BatchNorm (eval) -> leaky_relu."""
        y = x @ self.p[name + ".weight"].T
        if name + ".bias" in self.p:
            y = y + self.p[name + ".bias"]
        if name + ".bn.gamma" in self.p:
            y = (y - self.p[name + ".bn.mean"]) / torch.sqrt(self.p[name + ".bn.var"] + BN_EPS) * self.p[name + ".bn.gamma"] \
                + self.p[name + ".bn.beta"]
        return F.leaky_relu(y, SLOPE) if act else y

    @staticmethod
    def relative_pos_encoding(xyz, neigh):
        nb = xyz[neigh]                                   # [N,K,3]
        tile = xyz[:, None, :].expand_as(nb)
        rel = tile - nb
        dis = torch.sqrt((rel * rel).sum(-1, keepdim=True))
        return torch.cat([dis, rel, tile, nb], dim=-1)    # [N,K,10]

    def att_pooling(self, f, name):
        s = f @ self.p[name + "fc.weight"].T               # [N,K,d], no bias
        a = torch.softmax(s, dim=1)
        return self.conv((f * a).sum(1), name + "mlp")

    def building_block(self, xyz, f, neigh, name):
        f_xyz = self.conv(self.relative_pos_encoding(xyz, neigh), name + "mlp1")
        f = self.att_pooling(torch.cat([f[neigh], f_xyz], -1), name + "att_pooling_1")
        f_xyz = self.conv(f_xyz, name + "mlp2")
        return self.att_pooling(torch.cat([f[neigh], f_xyz], -1), name + "att_pooling_2")

    def dilated_res_block(self, f, xyz, neigh, name):
        y = self.conv(f, name + "mlp1")
        y = self.building_block(xyz, y, neigh, name + "LFA")
        y = self.conv(y, name + "mlp2", act=False)
        return F.leaky_relu(y + self.conv(f, name + "shortcut", act=False), SLOPE)

    def forward(self, features, xyz, neigh, pool, up):
        """features [N,6] (xyz, rgb); xyz / neigh / pool / up: lists over the 5 levels -> logits [N,13]."""
        f = self.conv(features, "fc0")
        enc = []
        for i in range(len(self.d_out)):
            e = self.dilated_res_block(f, xyz[i], neigh[i], "Encoder_layer_%d" % i)
            f = e[pool[i]].max(dim=1)[0]                   # random_sample: max over the K gathered rows
            if i == 0:
                enc.append(e)
            enc.append(f)
        f = self.conv(enc[-1], "decoder_0")
        for j in range(len(self.d_out)):
            f = self.conv(torch.cat([enc[-j - 2], f[up[-j - 1][:, 0]]], -1), "Decoder_layer_%d" % j)
        f = self.conv(self.conv(f, "fc1"), "fc2")
        return self.conv(f, "fc", act=False)


def colper_loss(logits, labels):
    """bim.py:110-116: sum over points of max(0, max((1 - onehot) * logits) - logit_y)."""
    onehot = F.one_hot(labels, logits.shape[1]).to(logits.dtype)
    real = (onehot * logits).sum(-1)
    other = ((1 - onehot) * logits).max(-1)[0]
    return torch.clamp(other - real, min=0).sum()


def loss_and_grad(orc, xyz0, rgb, labels, pyr):
    """-> (loss, logits [N,13], d loss / d rgb [N,3]) for one cloud; pyr = (xyz, neigh, pool, up) lists (numpy)."""
    t = [[torch.from_numpy(np.ascontiguousarray(a)).long() if a.dtype.kind == "i" else torch.from_numpy(np.ascontiguousarray(a))
          for a in lst] for lst in pyr]
    t[0] = [a.to(orc.dtype) for a in t[0]]
    c = torch.from_numpy(np.ascontiguousarray(rgb)).to(orc.dtype).requires_grad_(True)
    feats = torch.cat([torch.from_numpy(np.ascontiguousarray(xyz0, np.float32)).to(orc.dtype), c], -1)
    logits = orc.forward(feats, *t)
    loss = colper_loss(logits, torch.from_numpy(np.asarray(labels)).long())
    loss.backward()
    return float(loss.detach()), logits.detach().numpy(), c.grad.numpy()


def bim_step(xs, xs_adv, grad, eps, alpha, metric="l_inf", x_min=0.0, x_max=1.0):
    """One BIM update on the flattened colours of one cloud (bim.py:84-98, goal 'ut': ascent)."""
    xs, xs_adv, grad = (np.asarray(a, np.float32) for a in (xs, xs_adv, grad))
    if metric == "l_inf":
        out = np.clip(xs_adv + np.float32(alpha) * np.sign(grad), xs - np.float32(eps), xs + np.float32(eps))
    else:
        unit = grad / max(1e-12, float(np.sqrt((grad.astype(np.float64) ** 2).sum())))
        delta = xs_adv - xs + np.float32(alpha) * unit.astype(np.float32)
        nrm = float(np.sqrt((delta.astype(np.float64) ** 2).sum()))
        if nrm > eps:                                     # tf.clip_by_norm
            delta = delta * np.float32(eps / nrm)
        out = xs + delta
    return np.clip(out, x_min, x_max).astype(np.float32)
