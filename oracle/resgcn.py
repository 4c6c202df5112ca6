"""CPU oracle for the ResGCN-28 (dense DeepGCN) sem-seg attack path (numpy driver over oracle/psg_oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/pn2.py).  Restates, relative to /root/reference/ResGCN:
  gcn_lib/dense/torch_edge.py:32-79    pairwise_distance, dense_knn_matrix, DenseDilatedKnnGraph
  gcn_lib/dense/torch_nn.py:55-98      BasicConv (Conv -> ReLU -> BatchNorm), batched_index_select
  gcn_lib/dense/torch_vertex.py:23-100 EdgeConv2d (cat[x_i, x_j - x_i] -> conv -> max over k), ResDynBlock2d
  gcn_lib/dense/torch_vertex.py:8-20, 74-85, 103-115   MRConv2d, PlainDynBlock2d, DenseDynBlock2d (the `conv` /
                                       `block` switches of architecture.py:26-39)
  sem_seg_dense/architecture.py:58-68  DenseDeepGCN.forward
  sem_seg_dense/attacks/torchattacks/attacks/colper.py:17-39  NB_attack
and the input-gradient backward autograd derives (kNN graphs are constants: computed under no_grad).
EdgeConv is evaluated edge by edge exactly as the reference writes it (no algebraic split), so the HIP
path's split-GEMM identity is checked against the un-split form.
Parity status: pinned by tests/golden/gcn_*.npz (generated from the reference, tests/golden/make_golden.py).
"""
import ctypes

import numpy as np

from . import pn2
from .pn2 import _c, _fp, _ip, lib, linear_bwd, linear_fwd

F = np.float32
BN_EPS = 1e-5
K = 16


def knn_dilated(x, d, k=K):
    x = _c(x)
    out = np.empty((x.shape[0], k), np.int32)
    lib().orc_knn_dilated(_fp(x), x.shape[0], x.shape[1], k, int(d), _ip(out))
    return out


def pairwise_distance(x):
    x = _c(x)
    out = np.empty((x.shape[0], x.shape[0]), np.float32)
    lib().orc_pairwise_distance(_fp(x), x.shape[0], x.shape[1], _fp(out))
    return out


def _bn(sd, name):
    s = np.asarray(sd[name + ".weight"], np.float64) / np.sqrt(np.asarray(sd[name + ".running_var"], np.float64) + BN_EPS)
    t = np.asarray(sd[name + ".bias"], np.float64) - np.asarray(sd[name + ".running_mean"], np.float64) * s
    return s.astype(F), t.astype(F)


def _conv(sd, name):
    w = np.asarray(sd[name + ".weight"], F)
    return _c(w.reshape(w.shape[0], -1)), _c(np.asarray(sd[name + ".bias"], F))


class GCNOracle:
    def __init__(self, sd, n_blocks=28, block="res", conv="edge"):
        assert block in ("res", "plain", "dense") and conv in ("edge", "mr")
        self.n_blocks, self.block, self.conv = n_blocks, block, conv
        self.edge = []
        for e in range(n_blocks):
            base = "head.gconv.nn" if e == 0 else "backbone.%d.body.gconv.nn" % (e - 1)
            w, b = _conv(sd, base + ".0")
            s, t = _bn(sd, base + ".2")
            self.edge.append((w, b, s, t))
        self.fusion = _conv(sd, "fusion_block.0") + _bn(sd, "fusion_block.2")
        self.p1 = _conv(sd, "prediction.0.0") + _bn(sd, "prediction.0.2")
        self.p2 = _conv(sd, "prediction.1.0") + _bn(sd, "prediction.1.2")
        self.p3 = _conv(sd, "prediction.3.0")

    @staticmethod
    def _basic(x, w, b, s, t):
        z = linear_fwd(x, w, b, False)                 # Conv
        act = z > 0
        return (np.where(act, z, F(0)) * s + t).astype(F), act   # ReLU -> BatchNorm (torch_nn.py:57-63)

    def edge_conv(self, x, nbr, e):
        """EdgeConv2d.forward (torch_vertex.py:31-35): max_k BasicConv(cat[x_i, x_j - x_i])."""
        w, b, s, t = self.edge[e]
        n, c = x.shape
        xi = np.repeat(x[:, None, :], K, axis=1)
        xj = x[nbr]
        rows = _c(np.concatenate([xi, xj - xi], axis=2).reshape(n * K, 2 * c))
        y, act = self._basic(rows, w, b, s, t)
        y = y.reshape(n, K, -1)
        arg = y.argmax(axis=1)                          # first index on ties, like torch.max
        out = np.take_along_axis(y, arg[:, None, :], axis=1)[:, 0, :]
        return out.astype(F), (arg, act.reshape(n, K, -1))

    def mr_conv(self, x, nbr, e):
        """MRConv2d.forward (torch_vertex.py:16-20): BasicConv(cat[x, max_k (x_j - x_i)])."""
        w, b, s, t = self.edge[e]
        rel = x[nbr] - x[:, None, :]                    # [N,K,C]
        arg = rel.argmax(axis=1)                        # first index on ties, like torch.max
        mx = np.take_along_axis(rel, arg[:, None, :], axis=1)[:, 0, :]
        y, act = self._basic(_c(np.concatenate([x, mx], axis=1)), w, b, s, t)
        return y, (arg, act)

    def forward(self, x, graphs=None):
        """x [N,9] -> (logits [N,13], cache).  `graphs`: optional precomputed neighbour tables (teacher forcing)."""
        x = _c(x)
        cache = {"x": x, "nbr": [], "ec": [], "inp": []}
        feats = []
        cur = None
        for e in range(self.n_blocks):
            inp = x if e == 0 else cur
            if graphs is not None:
                nbr = graphs[e]
            else:   # dilation 1 + i for res / dense (architecture.py:23,28), 1 for plain (:36)
                nbr = knn_dilated(x[:, :3] if e == 0 else cur, 1 if e == 0 or self.block == "plain" else e)
            y, aux = (self.edge_conv if self.conv == "edge" else self.mr_conv)(inp, nbr, e)
            if e == 0 or self.block == "plain":
                cur = y                                   # PlainDynBlock2d: body(x) (torch_vertex.py:84-85)
            elif self.block == "res":
                cur = (y + cur).astype(F)                 # ResDynBlock2d: body(x) + x (torch_vertex.py:99-100)
            else:
                cur = _c(np.concatenate([cur, y], axis=1))   # DenseDynBlock2d: cat(x, body(x)) (torch_vertex.py:113-115)
            cache["nbr"].append(nbr)
            cache["ec"].append(aux)
            cache["inp"].append(inp)
            feats.append(cur)
        cache["widths"] = [f.shape[1] for f in feats]
        feats = _c(np.concatenate(feats, axis=1))
        fused, fact = self._basic(feats, *self.fusion)
        farg = fused.argmax(axis=0)
        fmax = fused[farg, np.arange(fused.shape[1])]
        cat = _c(np.concatenate([np.repeat(fmax[None, :], x.shape[0], axis=0), feats], axis=1))
        h1, a1 = self._basic(cat, *self.p1)
        h2, a2 = self._basic(h1, *self.p2)
        logits = linear_fwd(h2, self.p3[0], self.p3[1], False)
        cache.update(feats=feats, fact=fact, farg=farg, a1=a1, a2=a2, n=x.shape[0])
        return logits, cache

    def backward(self, cache, dlogits):
        """d loss / d x [N,9] given d loss / d logits [N,13]."""
        n = cache["n"]
        g2 = (_c(dlogits) @ self.p3[0]).astype(F)
        g2 = (g2 * self.p2[2] * cache["a2"]).astype(F)
        g1 = (g2.astype(np.float64) @ self.p2[0].astype(np.float64)).astype(F)
        g1 = (g1 * self.p1[2] * cache["a1"]).astype(F)
        dcat = (g1.astype(np.float64) @ self.p1[0].astype(np.float64)).astype(F)
        dfeats = dcat[:, 1024:].copy()
        gf = dcat[:, :1024].sum(axis=0, dtype=np.float64).astype(F)
        farg, fact = cache["farg"], cache["fact"]
        cols = np.arange(1024)
        gz = np.where(fact[farg, cols], gf * self.fusion[2], F(0)).astype(np.float64)
        np.add.at(dfeats, farg, (gz[:, None] * self.fusion[0].astype(np.float64)).astype(F))
        if self.block != "res" or self.conv != "edge":
            return self._backward_alt(cache, dfeats)
        G = dfeats[:, -64:].copy()
        dx = None
        for e in range(self.n_blocks - 1, -1, -1):
            dprev = self._edge_conv_bwd(cache, e, G)
            if e > 0:
                G = (dfeats[:, 64 * (e - 1):64 * e] + G + dprev).astype(F)
            else:
                dx = dprev
        return dx

    def _edge_conv_bwd(self, cache, e, G):
        """d/d(input of EdgeConv e) given G = d/d(its output) [N,64]."""
        n = cache["n"]
        w, b, s, t = self.edge[e]
        arg, act = cache["ec"][e]
        nbr = cache["nbr"][e]
        c = w.shape[1] // 2
        win_act = np.take_along_axis(act, arg[:, None, :], axis=1)[:, 0, :]
        gz = np.where(win_act, G * s, F(0)).astype(np.float64)            # [N,64] at the winning edge
        # d/d(cat[x_i, x_j - x_i]) of the winning edge of every (vertex, channel)
        dxi = np.zeros((n, c), np.float64)
        dxj = np.zeros((n, c), np.float64)
        w1, w2 = w[:, :c].astype(np.float64), w[:, c:].astype(np.float64)
        for k in range(K):
            gk = np.where(arg == k, gz, 0.0)                                # [N,64]
            if not gk.any():
                continue
            d1, d2 = gk @ w1, gk @ w2
            dxi += d1 - d2
            np.add.at(dxj, nbr[:, k], d2)
        return (dxi + dxj).astype(F)

    def _mr_conv_bwd(self, cache, e, G):
        """d/d(input of MRConv e): through BatchNorm, ReLU, the conv, then cat[x, max-relative]."""
        w, b, s, t = self.edge[e]
        arg, act = cache["ec"][e]
        nbr = cache["nbr"][e]
        c = w.shape[1] // 2
        gz = np.where(act, G * s, F(0)).astype(np.float64)
        dcat = gz @ w.astype(np.float64)                                    # [N,2C]
        dx = dcat[:, :c] - dcat[:, c:]
        j = nbr[np.arange(nbr.shape[0])[:, None], arg]                      # [N,C] winning neighbour per channel
        np.add.at(dx, (j, np.broadcast_to(np.arange(c), j.shape)), dcat[:, c:])
        return dx.astype(F)

    def _backward_alt(self, cache, dfeats_cat):
        """Backbone backward for the alternative blocks: walk the blocks downwards, d cur_e complete when reached."""
        widths = cache["widths"]
        offs = np.concatenate([[0], np.cumsum(widths)])
        dcur = [dfeats_cat[:, offs[e]:offs[e + 1]].astype(np.float64) for e in range(self.n_blocks)]
        conv_bwd = self._edge_conv_bwd if self.conv == "edge" else self._mr_conv_bwd
        dx = None
        for e in range(self.n_blocks - 1, -1, -1):
            g = dcur[e].astype(F)
            if self.block == "dense" and e > 0:
                gy, gpass = g[:, -64:], g[:, :-64]
            else:
                gy, gpass = g, (g if (self.block == "res" and e > 0) else None)
            dprev = conv_bwd(cache, e, _c(gy))
            if e == 0:
                dx = dprev
            else:
                dcur[e - 1] += dprev
                if gpass is not None:
                    dcur[e - 1] += gpass
        return dx


def ce_mean_grad(logits, labels):
    """nn.CrossEntropyLoss() (mean) on logits (colper.py:24,32): returns (d/dlogits, cost)."""
    z = logits - logits.max(axis=1, keepdims=True)
    lp = z - np.log(np.exp(z).sum(axis=1, keepdims=True, dtype=F))
    n = logits.shape[0]
    g = np.exp(lp).astype(F)
    g[np.arange(n), labels] -= F(1)
    return (g / F(n)).astype(F), float(-lp[np.arange(n), labels].mean(dtype=np.float64))


def nb_step(orc, x, state, ori, labels, alpha, eps, last, graphs=None):
    """One colper.NB_attack iteration for a single room: x [N,9] (colour replaced by `state` [N,3])."""
    xa = x.copy()
    xa[:, 3:6] = state
    logits, cache = orc.forward(xa, graphs=graphs)
    dl, cost = ce_mean_grad(logits, labels)
    g = orc.backward(cache, dl)[:, 3:6]
    stepped = (state + F(alpha) * np.sign(g)).astype(F)
    proj = np.clip(ori + np.clip(stepped - ori, -F(eps), F(eps)), 0, 1).astype(F)
    return (stepped if last else proj), g, logits, cost


# ------------------------------------------------------------------------------------------------
# ResGCN NU attacks: colper.py:42-120 (NU_attack), tcolper.py:51-170 (tar_NU_attack)
# ------------------------------------------------------------------------------------------------
def f_loss_grad_logits(z, y, mode, kappa, tsign, mask=None, target=None):
    """f-loss on raw logits with the reference's one-hot masking (a 0 takes part in every max).
    mode 0: NU f over all points; 1: tar_NU non_f, 2: tar_NU tar_f (class `target`), both over `mask` only.
    Returns (sum f, d sum f / d z)."""
    n, ncls = z.shape
    rows = np.arange(n)
    yy = np.full(n, int(target), np.int64) if mode == 2 else np.asarray(y, np.int64)
    others = z.copy()
    others[rows, yy] = -np.inf
    oi = others.argmax(axis=1)
    oth = others[rows, oi]
    oth_live = oth > 0
    oth = np.where(oth_live, oth, F(0))
    own = z[rows, yy]
    own_live = np.ones(n, bool)
    if mode == 0:
        own_live = own > 0
        own = np.where(own_live, own, F(0))
    jv, iv = (oth, own) if mode == 2 else (own, oth)
    val = F(tsign) * (jv - iv)
    passed = val >= -F(kappa)
    counted = np.ones(n, bool) if mode == 0 else np.asarray(mask, bool)
    g = np.zeros_like(z)
    gs = np.where(passed & counted, F(tsign), F(0))
    if mode == 2:
        g[rows, oi] += np.where(oth_live, gs, F(0))
        g[rows, yy] -= gs
    else:
        g[rows, yy] += np.where(own_live, gs, F(0))
        g[rows, oi] -= np.where(oth_live, gs, F(0))
    fsum = float(np.where(passed, val, -F(kappa))[counted].sum(dtype=np.float64))
    return fsum, g.astype(F)


def smooth_self_loss_grad(a, nb):
    """smooth(adv, adv) (colper.py:115-120): k smallest cdist(adv, adv) entries per row; gradient through BOTH
    arguments of cdist."""
    d, idx = pn2.smooth_knn(a, a, nb)
    diff = a[:, None, :].astype(F) - a[idx].astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = np.where(d[:, :, None] > 0, diff / d[:, :, None], F(0)).astype(np.float64)
    g = u.sum(axis=1)
    np.add.at(g, idx.reshape(-1), -u.reshape(-1, 3))
    return float(d.sum(dtype=np.float64)), g.astype(F)


def nu_step(orc, x_orig, x_base, w, m, v, t, labels, c, kappa, lr, nb, targeted_variant=False, mask=None, target=None,
            tsign=1.0, graphs=None):
    """One ResGCN NU / tar_NU optimisation step for a single room.  x_orig/x_base [N,9]; w,m,v [3,M] (reference
    layout, M = N or mask count).  Returns dict(cost, f, smooth, l2, grad_w, w, m, v, adv)."""
    from .attacks import adam_update, tanh_space
    sel = slice(None) if mask is None else np.asarray(mask, bool)
    adv = x_base.copy()
    col = adv[:, 3:6].T.copy()            # [3,N]
    col[:, sel] = tanh_space(w)
    adv[:, 3:6] = col.T
    logits, cache = orc.forward(adv, graphs=graphs)
    mode = 0 if not targeted_variant else (2 if target is not None else 1)
    fsum, dz = f_loss_grad_logits(logits, labels, mode, kappa, tsign, mask=mask, target=target)
    c_f, c_l2 = (c, 1.0) if not targeted_variant else (1.0, c)
    g_color = orc.backward(cache, (dz * F(c_f)).astype(F))[:, 3:6].T.copy()     # [3,N]
    sm, sg = smooth_self_loss_grad(np.ascontiguousarray(adv[:, 3:6]), nb)
    g_color += F(1e-4) * sg.T
    diff = (adv - x_orig).astype(F)
    l2 = float((diff.astype(np.float64) ** 2).sum())
    g_color += F(c_l2) * F(2) * diff[:, 3:6].T
    th = np.tanh(w.astype(F))
    g_w = (g_color[:, sel] * F(0.5) * (F(1) - th * th)).astype(F)
    w2, m2, v2 = adam_update(w, m, v, g_w, lr, t)
    return dict(cost=c_f * fsum + 1e-4 * sm + c_l2 * l2, f=fsum, smooth=sm, l2=l2, grad_w=g_w, w=w2, m=m2, v=v2, adv=adv,
                pred=logits.argmax(axis=1))
