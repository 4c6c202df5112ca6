"""The `random` baseline of the reference's ResGCN evaluation driver (ResGCN/sem_seg_dense/test.py:47-109): per room a
uniform colour noise of L2 norm 1, clean and noisy predictions, per-class intersection / union, and the TSV log.

Same call shape, `random_noise(model, loader, opt)`: `loader` yields objects with `.pos [B,N,3]`, `.x [B,N,6]`,
`.y [B,N]` (what torch_geometric's DenseDataLoader hands the reference); `opt` needs `.device`, `.n_classes`,
`.res_dir`.  Model forwards, arg-max / intersection / union counters and the L2 distance run on the MI355X
(`psg_gcn_forward`, `psg_seg_stats`, `psg_l2_dist`); the host keeps the reference's bookkeeping arrays and log format.
Returns the arrays the reference only logs, so callers and tests can read them."""
import os
import os.path as osp
import time

import numpy as np
import torch

from pointsecguard_amd import runtime
from pointsecguard_amd.harness import l2_distance


def _stats(logits_cn, gt, n_classes):
    """logits [B,C,N] -> (pred [B,N] int32, seen / inter / union int64 [n_classes]) on the device."""
    rows = logits_cn.transpose(1, 2).contiguous()
    counters, pred = runtime.seg_stats(rows, gt.to(torch.int32).contiguous(), n_classes)
    return pred, counters


def random_noise(model, loader, opt, noise_range=1.0, log=True):
    n = len(loader)
    Is, Us = np.empty((n, opt.n_classes)), np.empty((n, opt.n_classes))
    adv_Is, adv_Us = np.empty((n, opt.n_classes)), np.empty((n, opt.n_classes))
    dis, mious, adv_mious, acc, adv_acc = (np.empty(n) for _ in range(5))
    path = None
    if log:
        save_path = opt.res_dir + '/random/'
        os.makedirs(save_path, exist_ok=True)
        path = osp.join(save_path, 'log_' + time.strftime('%b-%d-%Y_%H%M', time.localtime()) + '.txt')
        with open(path, 'a+') as f:
            f.write("index\tl2dis\tadv_acc\tacc\tadv_miou\\miou\n")
    model.eval()
    with torch.no_grad():
        for i, data in enumerate(loader):
            pos, feat, gt = data.pos.to(opt.device), data.x.to(opt.device), data.y.to(opt.device)
            inputs = torch.cat((pos.transpose(2, 1).unsqueeze(3), feat.transpose(2, 1).unsqueeze(3)), 1).float().contiguous()
            pred, c = _stats(model(inputs), gt, opt.n_classes)
            noise = torch.empty_like(inputs[:, 3:6]).uniform_(0, 1)
            noise = noise / torch.norm(noise) * float(noise_range)
            adv_images = inputs.clone()
            adv_images[:, 3:6] += noise
            adv_pred, ca = _stats(model(adv_images), gt, opt.n_classes)
            d = l2_distance(inputs, adv_images)
            c, ca = c.cpu().numpy(), ca.cpu().numpy()
            acc[i] = float(c[1].sum()) / 4096          # correct points / 4096 like the reference (test.py:77)
            adv_acc[i] = float(ca[1].sum()) / 4096
            dis[i] = d.item()
            Is[i], Us[i], adv_Is[i], adv_Us[i] = c[1], c[2], ca[1], ca[2]
            with np.errstate(divide="ignore", invalid="ignore"):
                mious[i] = np.divide(np.sum(Is[i], 0), np.sum(Us[i], 0))
                adv_mious[i] = np.divide(np.sum(adv_Is[i], 0), np.sum(adv_Us[i], 0))
            mious[np.isnan(mious[i])] = 1
            adv_mious[np.isnan(adv_mious[i])] = 1
            if path:
                with open(path, 'a+') as f:
                    f.write("%d\t%.3f\t%.5f\t\t%.5f\t%.5f\t%.5f\n" % (i, dis[i], adv_acc[i], acc[i], np.mean(adv_mious[i]),
                                                                   np.mean(mious[i])))
    return dict(dis=dis, acc=acc, adv_acc=adv_acc, mious=mious, adv_mious=adv_mious, Is=Is, Us=Us, adv_Is=adv_Is,
                adv_Us=adv_Us, log=path)


def attack(model, test_loader, opt):
    """The `--attack` dispatcher of the reference's test.py:33-45."""
    from .attacks import NB_attack_exp, NU_attack_exp, tar_NB_attack_exp, tar_NU_attack_exp
    if opt.attack == 'random':
        return random_noise(model, test_loader, opt)
    if opt.attack == 'NU_attack':
        return NU_attack_exp(model, test_loader, opt)
    if opt.attack == 'tar_NU_attack':
        return tar_NU_attack_exp(model, test_loader, opt)
    if opt.attack == 'NB_attack':
        return NB_attack_exp(model, test_loader, opt)
    if opt.attack == 'tar_NB_attack':
        return tar_NB_attack_exp(model, test_loader, opt)
    raise ValueError("unknown attack %r" % (opt.attack,))
