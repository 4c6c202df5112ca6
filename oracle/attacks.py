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
