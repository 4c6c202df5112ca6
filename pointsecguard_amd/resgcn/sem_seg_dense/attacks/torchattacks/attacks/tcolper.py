"""Targeted colour attack on DenseDeepGCN (reference: .../attacks/tcolper.py: tar_NB_attack :7-46): only colours
under `mask` move, CrossEntropyLoss (mean over ALL rows) towards `target`, early exit when more than 90 % of the
masked points are classified as the target, projected colours written back at the end."""
import numpy as np
import torch

from pointsecguard_amd import _lib, runtime

from ..attack import Attack
from .colper import _gcn


class tar_NB_attack(Attack):
    def __init__(self, model, eps=0.3, alpha=2 / 255, iters=40, target=None, mask=None):
        super(tar_NB_attack, self).__init__("tar_NB_attack", model)
        self.model, self.eps, self.alpha, self.iters, self.target, self.mask = model, eps, alpha, iters, target, mask

    def forward(self, images, labels):
        net = _gcn(self.model)
        images = images.detach().to(self.device).float()
        B, C, N, _ = images.shape
        mask_np = self.mask.detach().cpu().numpy() if isinstance(self.mask, torch.Tensor) else np.asarray(self.mask)
        mask_d = torch.from_numpy(mask_np.astype(np.uint8)).to(self.device)
        mask_b = mask_d.bool()
        model, ws = net._packed(), net._workspace(B, N)
        net._generation += 1
        st = runtime.stream
        x0 = torch.empty(B, N, 9, device=self.device, dtype=torch.float32)
        _lib.call("psg_to_point_major", runtime.ptr(images[:, :, :, 0].contiguous()), B, 9, N, runtime.ptr(x0), st())
        ori = x0[:, :, 3:6].contiguous()
        dl = torch.empty(B, N, 13, device=self.device, dtype=torch.float32)
        out = torch.empty(B, 9, N, device=self.device, dtype=torch.float32)

        def snapshot():
            _lib.call("psg_to_channel_major", runtime.ptr(x0), B, 9, N, runtime.ptr(out), st())
            return out.unsqueeze(-1)

        for i in range(self.iters):
            net.consume_rng(1)
            logits = ws.forward(model, x0)
            pred = logits.argmax(dim=2)
            target_acc = pred[:, mask_b].eq(int(self.target)).sum().item() / float(mask_b.sum().item())
            if target_acc > 0.9:                                    # tcolper.py:37-38
                return snapshot()
            _lib.call("psg_ce_logp_grad", runtime.ptr(logits), None, int(self.target), B * N, B * N, 13, 1.0 / (B * N),
                      runtime.ptr(dl), None, st())
            dx0 = ws.backward(model, dl)
            # descent, projected colour kept in x0 (the reference writes the projected colour back, :45)
            _lib.call("psg_pgd_step", runtime.ptr(x0), runtime.ptr(dx0), runtime.ptr(ori), runtime.ptr(mask_d), B, N,
                      float(self.alpha), float(self.eps), -1.0, 0, st())
        return snapshot()


class tar_NU_attack(Attack):
    """Targeted norm-unbounded attack (tcolper.py:51-170): masked colours, cost = f + 1e-4*Smooth + c*L2, learning
    rate halved with a fresh optimiser every 50 steps, restart check every 10 steps."""

    def __init__(self, model, c=1e-4, kappa=0, steps=1000, lr=0.01, target=None, mask=None):
        super(tar_NU_attack, self).__init__("tar_NU_attack", model)
        self.c, self.kappa, self.steps, self.lr, self.target, self.mask = c, kappa, steps, lr, target, mask

    def forward(self, images, labels):
        from .nu import gcn_nu_attack
        return gcn_nu_attack(self, images, labels, mask=self.mask, target=self.target, neighbour=5, targeted_variant=True)
