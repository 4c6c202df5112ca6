"""The ResGCN attack experiment loops of the reference (ResGCN/sem_seg_dense/attacks.py:42-321; the reference keeps them
in a module `attacks.py` next to the directory `attacks/` that holds its torchattacks package -- here they live in that
package's __init__, so `from attacks import NB_attack_exp` and `attacks.torchattacks` both resolve) --
`NU_attack_exp`, `NB_attack_exp`, `tar_NU_attack_exp`, `tar_NB_attack_exp(model, test_loader, opt)` -- with the same
per-room protocol, attack hyper-parameters, skip rules, metrics and TSV log columns; forwards, attacks, the per-class
intersection / union counters (`psg_seg_stats`) and the L2 distance (`psg_l2_dist`) run on the MI355X.

`test_loader` yields objects with `.pos [B,N,3]`, `.x [B,N,6]`, `.y [B,N]` (torch_geometric's DenseDataLoader in the
reference); `opt` needs `.device`, `.n_classes`, `.res_dir` and, for the targeted loops, `.target`, `.origin`,
`.left_ratio`, `.att_type`.  `opt.attack_kwargs` (optional dict) overrides the attack's hyper-parameters, e.g. fewer
steps for a smoke run; without it the reference's values are used.  Every loop returns its bookkeeping arrays.

One deliberate difference: for the targeted loops the reference leaves the adversarial I / U entry of the attacked
class uninitialised (`np.empty`, attacks.py:303-304 `continue`), so its `other_miou` column contains whatever was in
memory; here that entry is 0 / 0 and is left out of the ratio.
"""
import os
import os.path as osp
import time

import numpy as np
import torch

from pointsecguard_amd import runtime
from pointsecguard_amd.harness import l2_distance


def _inputs(data, device):
    pos, feat = data.pos.to(device), data.x.to(device)
    return torch.cat((pos.transpose(2, 1).unsqueeze(3), feat.transpose(2, 1).unsqueeze(3)), 1).float().contiguous()


def _pred_stats(model, inputs, gt, n_classes):
    rows = model(inputs).transpose(1, 2).contiguous()                    # [B,N,C]
    counters, pred = runtime.seg_stats(rows, gt.to(torch.int32).contiguous(), n_classes)
    return pred, counters.cpu().numpy()


def _open_log(opt, sub, header, preamble=None):
    save_path = opt.res_dir + sub
    os.makedirs(save_path, exist_ok=True)
    path = osp.join(save_path, 'log_' + time.strftime('%b-%d-%Y_%H%M', time.localtime()) + '.txt')
    with open(path, 'w') as f:
        if preamble:
            f.write(preamble)
        f.write(header)
    return path


def _untargeted(model, test_loader, opt, sub, make_attack):
    n = len(test_loader)
    Is, Us, adv_Is, adv_Us = (np.empty((n, opt.n_classes)) for _ in range(4))
    dis, other_acc, acc, mious, adv_mious = (np.empty(n) for _ in range(5))
    path = _open_log(opt, sub, "index\tL2_dis\tother_acc\tacc\tadv_miou\tmiou\n")
    model.eval()
    for i, data in enumerate(test_loader):
        inputs = _inputs(data, opt.device)
        gt = data.y.to(opt.device)
        batch_size = inputs.shape[0]
        _, c = _pred_stats(model, inputs, gt, opt.n_classes)
        acc[i] = float(c[1].sum()) / 4096
        adv_images = make_attack()(inputs, gt)
        _, ca = _pred_stats(model, adv_images, gt, opt.n_classes)
        other_acc[i] = float(ca[1].sum()) / 4096
        dis[i] = l2_distance(inputs, adv_images.contiguous()).item() / batch_size
        Is[i], Us[i], adv_Is[i], adv_Us[i] = c[1], c[2], ca[1], ca[2]
        mious[i] = np.divide(np.sum(Is[i], 0), np.sum(Us[i], 0))
        adv_mious[i] = np.divide(np.sum(adv_Is[i], 0), np.sum(adv_Us[i], 0))
        with open(path, 'a+') as f:
            f.write("%d\t%.3f\t%.5f\t%.5f\t%.5f\t%.5f\n" % (i, dis[i], other_acc[i], acc[i], np.mean(adv_mious[i]), np.mean(mious[i])))
    return dict(dis=dis, other_acc=other_acc, acc=acc, mious=mious, adv_mious=adv_mious, Is=Is, Us=Us, adv_Is=adv_Is,
                adv_Us=adv_Us, log=path)


def _kw(opt, **defaults):
    defaults.update(getattr(opt, "attack_kwargs", None) or {})
    return defaults


def NU_attack_exp(model, test_loader, opt):
    from . import torchattacks
    return _untargeted(model, test_loader, opt, '/NU_attack_exp/',
                       lambda: torchattacks.NU_attack(model, **_kw(opt, c=1e-1, kappa=0, steps=1000, lr=0.1)))


def NB_attack_exp(model, test_loader, opt):
    from . import torchattacks
    return _untargeted(model, test_loader, opt, '/NB_attack_exp/',
                       lambda: torchattacks.NB_attack(model, **_kw(opt, eps=0.3, alpha=2 / 255, iters=50)))


def _targeted(model, test_loader, opt, sub, make_attack):
    target, ori = opt.target, opt.origin
    n = len(test_loader)
    Is, Us = np.empty((n, opt.n_classes)), np.empty((n, opt.n_classes))
    adv_Is, adv_Us = np.zeros((n, opt.n_classes)), np.zeros((n, opt.n_classes))
    dis, other_acc, acc, target_acc, mious, other_mious = (np.empty(n) for _ in range(6))
    skipped = np.zeros(n, bool)
    path = _open_log(opt, sub + opt.att_type + str(target) + '_' + str(ori) + '/',
                     "index\tcount\tL2_dis\ttarget_acc\tother_acc\tacc\tother_miou\tmiou\n",
                     preamble="left_ratio=" + str(opt.left_ratio) + '\n')
    model.eval()
    for i, data in enumerate(test_loader):
        inputs = _inputs(data, opt.device)
        gt = data.y.to(opt.device)
        batch_size = inputs.shape[0]
        pred, c = _pred_stats(model, inputs, gt, opt.n_classes)
        acc[i] = float(c[1].sum()) / 4096
        mask = (gt == ori)[0]
        count = int(mask.sum().item())
        if count <= 500:                                   # attacks.py:274-275: too few points of the attacked class
            skipped[i] = True
            continue
        temp_acc = float((pred.view_as(gt)[:, mask] == gt[:, mask]).sum().item()) / count
        if temp_acc < 0.5:                                 # :278-279: the class is not recognised to begin with
            skipped[i] = True
            continue
        adv_images = make_attack(target, mask)(inputs, gt)
        adv_pred, _ = _pred_stats(model, adv_images, gt, opt.n_classes)
        adv_pred = adv_pred.view_as(gt)
        other_acc[i] = float((adv_pred[0, ~mask] == gt[0, ~mask]).sum().item()) / 4096
        target_acc[i] = float((adv_pred[0, mask] == target).sum().item()) / count
        dis[i] = l2_distance(inputs, adv_images.contiguous()).item() / batch_size
        Is[i], Us[i] = c[1], c[2]
        ap, tg = adv_pred[0].cpu().numpy(), gt[0].cpu().numpy()
        for cl in range(opt.n_classes):
            if cl == ori:
                continue
            adv_Is[i, cl] = np.sum(np.logical_and(ap == cl, tg == cl), dtype=np.float32)
            adv_Us[i, cl] = np.sum(np.logical_or(ap == cl, tg == cl), dtype=np.float32)
        mious[i] = np.divide(np.sum(Is[i], 0), np.sum(Us[i], 0))
        other_mious[i] = np.divide(np.sum(adv_Is[i], 0), np.sum(adv_Us[i], 0))
        with open(path, 'a+') as f:
            f.write("%d\t%d\t%.3f\t%.3f\t%.5f\t%.5f\t%.5f\t%.5f\n" % (i, count, dis[i], target_acc[i], other_acc[i], acc[i],
                                                                    np.mean(other_mious[i]), np.mean(mious[i])))
    return dict(dis=dis, other_acc=other_acc, target_acc=target_acc, acc=acc, mious=mious, other_mious=other_mious,
                skipped=skipped, log=path)


def tar_NU_attack_exp(model, test_loader, opt):
    from . import torchattacks
    return _targeted(model, test_loader, opt, '/tar_NU_attack/',
                     lambda target, mask: torchattacks.tar_NU_attack(
                         model, **_kw(opt, c=1e-1, kappa=0, steps=1000, lr=0.1), target=target, mask=mask))


def tar_NB_attack_exp(model, test_loader, opt):
    from . import torchattacks
    return _targeted(model, test_loader, opt, '/tar_NB_attack/',
                     lambda target, mask: torchattacks.tar_NB_attack(
                         model, **_kw(opt, eps=0.4, alpha=0.04, iters=50), target=target, mask=mask))
