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


# ---- the other attacks of the reference's tester (tester_S3DIS.py:36-44): TBIM / tar_NBattack, NUattack, tar_NUattack.
# Source-read restatements like everything above (PARITY UNPINNED).

def colper_loss_masked(logits, ys, mask=None):
    """TBIM.colperloss / tar_NUattack.NUloss (bim.py:393-397, tar_NUattack.py:105-110): the hinge of colper_loss against
    the labels `ys` (the target class on the origin points), each point's term multiplied by mask."""
    onehot = F.one_hot(ys, logits.shape[1]).to(logits.dtype)
    real = (onehot * logits).sum(-1)
    other = ((1 - onehot) * logits).max(-1)[0]
    loss = torch.clamp(other - real, min=0)
    if mask is not None:
        loss = loss * mask.to(logits.dtype)
    return loss.sum()


def score_and_grad(orc, xyz0, rgb, ys, pyr, mask=None):
    """-> (score, logits [N,13], d score / d rgb [N,3]) of the (masked) hinge for one cloud."""
    t = [[torch.from_numpy(np.ascontiguousarray(a)).long() if a.dtype.kind == "i" else torch.from_numpy(np.ascontiguousarray(a))
          for a in lst] for lst in pyr]
    t[0] = [a.to(orc.dtype) for a in t[0]]
    c = torch.from_numpy(np.ascontiguousarray(rgb)).to(orc.dtype).requires_grad_(True)
    feats = torch.cat([torch.from_numpy(np.ascontiguousarray(xyz0, np.float32)).to(orc.dtype), c], -1)
    logits = orc.forward(feats, *t)
    m = None if mask is None else torch.from_numpy(np.asarray(mask, np.float32))
    score = colper_loss_masked(logits, torch.from_numpy(np.asarray(ys)).long(), m)
    score.backward()
    return float(score.detach()), logits.detach().numpy(), c.grad.numpy()


NU_BOUND = np.float32(1.0 - 1e-6)


def nu_color(xs, dws, mask=None):
    """NUattack.py:29-31 (+ tar_NUattack.py:41): ws = atanh(2 b x - b) + d_ws, adv = (tanh(ws) + 1) / 2; masked variant
    adv = mask adv + (1 - mask) x.  float32 like the TF graph."""
    xs, dws = np.asarray(xs, np.float32), np.asarray(dws, np.float32)
    ws = np.arctanh(np.float32(2.0) * NU_BOUND * xs - NU_BOUND).astype(np.float32) + dws
    adv = (np.float32(0.5) * (np.tanh(ws) + np.float32(1.0))).astype(np.float32)
    if mask is not None:
        adv = np.where(np.asarray(mask, bool)[:, None], adv, xs)
    return adv, ws


def nu_adam_step(xs, dws, m, v, t, dscore_drgb, c, lr, mask=None, b1=0.9, b2=0.999, eps=1e-8):
    """One tf.train.AdamOptimizer step on d_ws of loss = |adv - x|_2 + c * score (NUattack.py:55-59): the gradient of the
    distance term is (adv - x) / |adv - x|_2, the chain through tanh is (1 - tanh^2) / 2 (x mask), Adam in TF1's form
    lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), d_ws -= lr_t m / (sqrt(v) + eps).  Returns (d_ws, m, v, dist)."""
    adv, ws = nu_color(xs, dws, mask)
    d = (adv - np.asarray(xs, np.float32)).astype(np.float64)
    dist = float(np.sqrt((d ** 2).sum()))
    g = (d / dist if dist > 0 else np.zeros_like(d)) + float(c) * np.asarray(dscore_drgb, np.float64)
    g = g * 0.5 * (1.0 - np.tanh(ws.astype(np.float64)) ** 2)
    if mask is not None:
        g = g * np.asarray(mask, np.float64)[:, None]
    g = g.astype(np.float32)
    m = (np.float32(b1) * m + np.float32(1 - b1) * g).astype(np.float32)
    v = (np.float32(b2) * v + np.float32(1 - b2) * g * g).astype(np.float32)
    lr_t = np.float32(lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t))
    dws = (np.asarray(dws, np.float32) - lr_t * m / (np.sqrt(v) + np.float32(eps))).astype(np.float32)
    return dws, m, v, dist


def mean_iou(y_pred, y_true):
    """compute_iou of the attack classes (bim.py:153-165): sklearn's confusion_matrix over the labels that occur, mean of
    intersection / union."""
    y_pred, y_true = np.asarray(y_pred).ravel(), np.asarray(y_true).ravel()
    labs = np.unique(np.concatenate([y_pred, y_true]))
    iou = []
    for l in labs:
        inter = np.sum((y_pred == l) & (y_true == l))
        union = np.sum(y_true == l) + np.sum(y_pred == l) - inter
        iou.append(inter / np.float32(union))
    return float(np.mean(iou))
