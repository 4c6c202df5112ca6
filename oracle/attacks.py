"""CPU oracle for the colour-perturbation attack loops (numpy, float32 op-for-op).

TEST INFRASTRUCTURE ONLY (see oracle/pn2.py).  Restates, relative to /root/reference:
  PointNet/attacks/torchattacks/attacks/nontarget.py:18-42   NB_attack.forward
  PointNet/attacks/torchattacks/attacks/target.py:18-45      tar_NB_attack.forward
FPS start draws (`torch.randint`, pointnet_util.py:75) are passed in explicitly as
starts[iteration][level][room] so the stream can be replayed without the reference.
"""
import numpy as np

from . import pn2

F = np.float32


def model_forward(orc, images, starts):
    """images [B,9,N] channel-major; starts [4][B].  Returns (logp [B,N,13], caches)."""
    logps, caches = [], []
    for b in range(images.shape[0]):
        x = np.ascontiguousarray(images[b].T)
        geom = orc.geometry(x[:, :3], [int(starts[l][b]) for l in range(4)])
        lp, cache = orc.forward(x, geom)
        logps.append(lp)
        caches.append(cache)
    return np.stack(logps), caches


def nb_attack(orc, images, labels, eps, alpha, iters, starts, record=None):
    """NB_attack.forward (nontarget.py:18-42).  Returns adv_images [B,9,N] whose colour is the
    UN-projected last step (the projection lands only in `color`)."""
    images = np.asarray(images, F)
    labels = np.asarray(labels).astype(np.int64)
    B, _, N = images.shape
    color = images[:, 3:6].copy()
    ori = color.copy()
    adv = images.copy()
    for it in range(iters):
        adv[:, 3:6] = color
        logp, caches = model_forward(orc, adv, starts[it])
        grad = np.empty_like(color)
        for b in range(B):
            # cost = CE_sum(all rooms)/N  (nontarget.py:34: outputs.size(1) == N)
            dlogp, _ = pn2.nll_logp_grad(logp[b], labels[b], 1.0 / N)
            grad[b] = orc.backward_color(caches[b], dlogp).T
        adv[:, 3:6] = adv[:, 3:6] + F(alpha) * np.sign(grad)
        eta = np.clip(adv[:, 3:6] - ori, -F(eps), F(eps))
        color = np.clip(ori + eta, F(0), F(1)).astype(F)
        if record is not None:
            record(it, adv, color, grad, logp)
    return adv


def tar_nb_attack(orc, images, labels, eps, alpha, iters, target, mask, starts, record=None):
    """tar_NB_attack.forward (target.py:18-45): only colours under `mask` [N] move (on every batch
    row), loss = CE(mean)(outputs[0], target) -- batch row 0 only -- descent step."""
    images = np.asarray(images, F)
    mask = np.asarray(mask, bool)
    B, _, N = images.shape
    color = images[:, 3:6][:, :, mask].copy()
    ori = color.copy()
    adv = images.copy()
    tgt = np.full(N, int(target), np.int64)
    for it in range(iters):
        adv[:, 3:6][:, :, mask] = color
        logp, caches = model_forward(orc, adv, starts[it])
        grad = np.zeros((B, 3, N), F)
        dlogp, _ = pn2.nll_logp_grad(logp[0], tgt, 1.0 / N)  # CrossEntropyLoss() = mean over N points
        grad[0] = orc.backward_color(caches[0], dlogp).T
        g = grad[:, :, mask]
        adv[:, 3:6][:, :, mask] = adv[:, 3:6][:, :, mask] - F(alpha) * np.sign(g)
        eta = np.clip(adv[:, 3:6][:, :, mask] - ori, -F(eps), F(eps))
        color = np.clip(ori + eta, F(0), F(1)).astype(F)
        if record is not None:
            record(it, adv, color, grad, logp)
    return adv


# ------------------------------------------------------------------------------------------------
# NU (Adam in tanh space) attacks: nontarget.py:52-135, target.py:62-175
# ------------------------------------------------------------------------------------------------
BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-8


def tanh_space(w):
    return (F(0.5) * (np.tanh(w.astype(F)) + F(1))).astype(F)


def inverse_tanh_space(c):
    x = c.astype(F) * F(2) - F(1)
    return (F(0.5) * np.log((F(1) + x) / (F(1) - x))).astype(F)


def f_loss_grad(logp, y, kappa, tsign):
    """f = clamp(tsign*(p_y - max_{k!=y} p_k), min=-kappa) on p = softmax(logp) (nontarget.py:119-128);
    returns (sum f, d sum f / d logp)."""
    n = logp.shape[0]
    z = logp - logp.max(axis=1, keepdims=True)
    p = np.exp(z)
    p = (p / p.sum(axis=1, keepdims=True, dtype=F)).astype(F)
    rows = np.arange(n)
    others = p.copy()
    others[rows, y] = 0.0
    oi = others.argmax(axis=1)
    val = F(tsign) * (p[rows, y] - others[rows, oi])
    passed = val >= -F(kappa)
    g = np.zeros_like(p)
    g[rows, y] = np.where(passed, F(tsign), F(0))
    g[rows, oi] += np.where(passed, -F(tsign), F(0))
    dot = (g * p).sum(axis=1, keepdims=True, dtype=F)
    return float(np.where(passed, val, -F(kappa)).sum(dtype=np.float64)), (p * (g - dot)).astype(F)


def smooth_loss_grad(adv_color, ref_color, nb):
    """Sum of the nb smallest Euclidean distances per adversarial colour + gradient (row 0 only)."""
    d, idx = pn2.smooth_knn(adv_color, ref_color, nb)
    # cdist backward: grad = diff / dist with the (noisy, expansion-based) forward distance, 0 where dist == 0
    diff = adv_color[:, None, :].astype(F) - ref_color[idx].astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.where(d[:, :, None] > 0, diff / d[:, :, None], F(0)).sum(axis=1)
    return float(d.sum(dtype=np.float64)), g.astype(F)


def adam_update(w, m, v, g, lr, t):
    """torch.optim.Adam single-tensor step (fp32 tensors, python-double scalars)."""
    m = (m + (g - m) * F(1 - BETA1)).astype(F)
    v = (v * F(BETA2) + F(1 - BETA2) * (g * g)).astype(F)
    bc1, bc2 = 1.0 - BETA1 ** t, 1.0 - BETA2 ** t
    denom = (np.sqrt(v) / F(np.sqrt(bc2)) + F(ADAM_EPS)).astype(F)
    w = (w + F(-(lr / bc1)) * (m / denom)).astype(F)
    return w, m, v


def nu_step(orc, images, x_base, w, m, v, t, labels, starts_t, c, kappa, lr, neighbour, mask=None, target=None,
            tsign=1.0):
    """One NU / tar_NU optimisation step.  images = originals [B,9,N]; x_base = current best_adv_images.
    Returns dict(adv, cost, f, smooth, l2, grad_w, w, m, v, pred)."""
    B, _, N = images.shape
    sel = slice(None) if mask is None else mask
    adv = x_base.copy()
    color = tanh_space(w)                       # [B,3,M]
    adv[:, 3:6][:, :, sel] = color
    logp, caches = model_forward(orc, adv, starts_t)
    f_sum, g_color = 0.0, np.zeros((B, 3, N), F)
    for b in range(B):
        y = labels[b] if target is None else np.full(N, int(target), np.int64)
        fs, dlogp = f_loss_grad(logp[b], y, kappa, tsign)
        f_sum += fs
        g_color[b] = orc.backward_color(caches[b], dlogp).T
    sm, sg = smooth_loss_grad(np.ascontiguousarray(adv[0, 3:6].T), np.ascontiguousarray(images[0, 3:6].T), neighbour)
    g_color[0] += F(c) * sg.T
    diff = (adv - images).astype(F)
    l2 = float((diff.astype(np.float64) ** 2).sum())
    g_color += F(c) * F(2) * diff[:, 3:6]
    th = np.tanh(w.astype(F))
    g_w = (g_color[:, :, sel] * F(0.5) * (F(1) - th * th)).astype(F)
    w2, m2, v2 = adam_update(w, m, v, g_w, lr, t)
    return dict(adv=adv, cost=f_sum + c * sm + c * l2, f=f_sum, smooth=sm, l2=l2, grad_w=g_w, w=w2, m=m2, v=v2,
                pred=logp.argmax(axis=2), logp=logp)
