"""Host loop of the NU (norm-unbounded, Adam in tanh space) attacks over the libpsg kernels.

Reference: PointNet/attacks/torchattacks/attacks/nontarget.py:52-135 (NU_attack.forward/f/smooth) and
target.py:62-175 (tar_NU_attack.forward/non_f/tar_f/smooth).  Per step:
  colour = 1/2 (tanh w + 1) -> model forward -> f-loss on softmax(log-probs) + c*Smooth + c*L2 ->
  input-gradient backward -> Adam step on w.
Everything numeric runs in HIP kernels (psg_nu_*, psg_smooth_knn, psg_pn2_forward/backward); the host
only evaluates the reference's control flow (early exits on accuracy, learning-rate halving with a fresh
optimiser every 50 steps, the every-10-steps restart), which needs one scalar read-back per step exactly
like the reference's `.item()` calls.

Reference behaviours kept on purpose: the Smooth term and `other_acc`
use batch row 0 only; `acc` divides by 4096 whatever the batch; the returned image lags the optimiser by
one step; `_targeted` stays +1 unless set_attack_mode() is called; the restart adds uniform noise to the
masked colours (overwritten by tanh_space on the next step) and clamps ALL nine channels to [0,1], which
moves the xyz coordinates -- the geometry plan is rebuilt from the current coordinates after it.
"""
import numpy as np
import torch

from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.models.pointnet2_sem_seg import draw_fps_starts

from ._common import labels_to_device, mask_to_device, psg_model

BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-8
CHUNK = 10  # geometry plan horizon: restarts can only happen after steps that are multiples of 10


def nu_attack(atk, images, labels, mask, target, neighbour, targeted_variant=False, trace=None):
    net = psg_model(atk.model)
    dev = atk.device
    images = images.detach().to(dev).float().contiguous()
    B, C, N = images.shape
    labels_d = labels_to_device(labels, dev)
    mask_d = mask_to_device(mask, N, dev) if mask is not None else None
    mask_b = mask_d.bool() if mask_d is not None else None
    model = net._packed()
    net._generation += 1
    ws = net._workspace(B, N, CHUNK + 1)
    st = runtime.stream

    x0 = torch.empty(B, N, 9, device=dev, dtype=torch.float32)
    _lib.call("psg_to_point_major", runtime.ptr(images), B, 9, N, runtime.ptr(x0), st())
    ori = x0[:, :, 3:6].contiguous()
    x0_orig = x0.clone()
    extra_l2 = 0.0  # (adv - images)^2 over the non-colour channels: non-zero only after a restart clamped them
    w = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
    _lib.call("psg_nu_inverse_tanh", runtime.ptr(x0), B, N, runtime.ptr(w), st())
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    dlogp = torch.empty(B, N, 13, device=dev, dtype=torch.float32)
    dx0 = torch.empty(B, N, 9, device=dev, dtype=torch.float32)
    sgrad = torch.empty(N, 3, device=dev, dtype=torch.float32)
    pred = torch.empty(B, N, device=dev, dtype=torch.int32)
    scal = torch.zeros(3, device=dev, dtype=torch.float32)  # f, smooth, l2
    lr, adam_t = float(atk.lr), 0
    prev_cost = [1e10] * atk.steps
    tsign = float(atk._targeted)
    use_target = targeted_variant and target is not None
    planned_until = 0
    n_mask = int(mask_b.sum().item()) if mask_b is not None else 0   # once: the loop reads back ONE tensor per step
    out = torch.empty_like(images)

    def snapshot():
        _lib.call("psg_to_channel_major", runtime.ptr(x0), B, 9, N, runtime.ptr(out), st())
        return out

    for step in range(atk.steps):
        _lib.call("psg_nu_tanh_color", runtime.ptr(w), runtime.ptr(mask_d), B, N, runtime.ptr(x0), st())
        if step >= planned_until:
            # Geometry windows end after steps 10, 20, 30, ..: a restart (which draws from the RNG and may move xyz) can
            # only follow a step that is a multiple of 10 greater than 10, so the FPS draws of a window come out of the
            # generator in the reference's order.  The first forward is planned ALONE: the accuracy exits can end the
            # attack right after it (at batch 32 the targeted one always does, target.py:105-121), and a plan for
            # eleven forwards would then be ten too many; the rest of the first window follows with step 1.
            window_end = 1 if step == 0 else ((step - 1) // CHUNK + 1) * CHUNK + 1      # [0], [1..10], [11..20], ..
            n_plan = min(window_end - step, atk.steps - step)
            starts = draw_fps_starts(B, N, n_plan).to(dev)
            ws.plan_build(x0, starts, n_plan)
            plan_base, planned_until = step, step + n_plan
        slot = step - plan_base
        logp = ws.forward(model, slot, x0)
        scal.zero_()
        _lib.call("psg_nu_f_loss_grad", runtime.ptr(logp), None if use_target else runtime.ptr(labels_d),
                  int(target) if use_target else 0, B * N, 13, float(atk.kappa), tsign, runtime.ptr(dlogp),
                  runtime.ptr(scal[0:1]), runtime.ptr(pred), st())
        ws.backward(model, slot, dlogp, dx0)
        _lib.call("psg_smooth_knn", ctypes_off(x0, 3), 9, runtime.ptr(ori), 3, N, int(neighbour),
                  runtime.ptr(scal[1:2]), runtime.ptr(sgrad), st())
        adam_t += 1
        _lib.call("psg_nu_adam_step", runtime.ptr(w), runtime.ptr(m), runtime.ptr(v), runtime.ptr(mask_d),
                  runtime.ptr(dx0), runtime.ptr(x0), runtime.ptr(ori), runtime.ptr(sgrad), float(atk.c), float(atk.c),
                  lr, BETA1, BETA2, ADAM_EPS, adam_t, B, N, runtime.ptr(scal[2:3]), st())
        # ---- control flow of the reference (one read-back per step, like its .item() calls)
        correct = pred.eq(labels_d)
        if targeted_variant:
            # masked count as a logical AND (boolean-mask indexing would synchronise and launch a nonzero every step)
            tgt_hits = ((pred.eq(int(target)) if use_target else correct) & mask_b.unsqueeze(0)).sum()
            stats = torch.stack([correct.sum().float(), tgt_hits.float(), scal[0], scal[1], scal[2]]).cpu()
        else:
            stats = torch.stack([correct.sum().float(), scal[0], scal[0], scal[1], scal[2]]).cpu()
        n_correct, n_tgt, f_loss, sm_loss, l2_loss = (float(z) for z in stats)
        l2_loss += extra_l2
        cost = f_loss + float(atk.c) * sm_loss + float(atk.c) * l2_loss
        prev_cost[step] = cost
        if trace is not None:
            trace(step=step, cost=cost, f=f_loss, smooth=sm_loss, l2=l2_loss, w=w, m=m, v=v, dx0=dx0, x0=x0, pred=pred)
        if not targeted_variant:
            if n_correct / 4096 < 1 / 13:          # nontarget.py:87,95-96
                return snapshot()
            continue
        target_acc = n_tgt / float(n_mask)
        if (not use_target and target_acc < 1 / 13) or (use_target and target_acc > 0.9):   # target.py:116-121
            return snapshot()
        if step > 0 and step % 50 == 0:             # target.py:123-125: halve lr, NEW optimiser (moments reset)
            atk.lr = atk.lr / 2
            lr, adam_t = float(atk.lr), 0
            m.zero_()
            v.zero_()
        if step > 10 and step % 10 == 0 and cost >= prev_cost[step - 10]:   # target.py:127-132
            noise = torch.empty(B, 3, n_mask, device=dev, dtype=torch.float32).uniform_(0, 1)
            col = x0[:, :, 3:6].transpose(1, 2)      # view [B,3,N]
            col[:, :, mask_b] = col[:, :, mask_b] + noise
            x0.clamp_(min=0, max=1)                 # ALL channels, like the reference
            d = x0 - x0_orig
            extra_l2 = float((d[:, :, 0:3] ** 2).sum().item() + (d[:, :, 6:9] ** 2).sum().item())
            planned_until = step + 1               # xyz may have moved: rebuild the plan before the next forward
    return snapshot()


def ctypes_off(t, n_floats):
    """Device pointer to element `n_floats` of a float32 tensor."""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * n_floats)
