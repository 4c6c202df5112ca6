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


def nu_attack(atk, images, labels, mask, target, neighbour, targeted_variant=False, trace=None, starts_fn=None):
    """`starts_fn(step, n_plan)` (tests) supplies the [n_plan, 4, B] FPS start indices of a geometry window instead of the
    generator draws."""
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
    nn_state = torch.empty(N, int(neighbour), device=dev, dtype=torch.int32)   # Smooth term: last step's neighbours of row 0
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
            starts = (draw_fps_starts(B, N, n_plan) if starts_fn is None else starts_fn(step, n_plan)).to(dev)
            ws.plan_build(x0, starts, n_plan)
            plan_base, planned_until = step, step + n_plan
        slot = step - plan_base
        logp = ws.forward(model, slot, x0)
        scal.zero_()
        _lib.call("psg_nu_f_loss_grad", runtime.ptr(logp), None if use_target else runtime.ptr(labels_d),
                  int(target) if use_target else 0, B * N, 13, float(atk.kappa), tsign, runtime.ptr(dlogp),
                  runtime.ptr(scal[0:1]), runtime.ptr(pred), st())
        ws.backward(model, slot, dlogp, dx0)
        # (the rooms entry with one room: it keeps the neighbour lists, and every step after the first starts its scan from
        # the rigorous bound they give - same rows as psg_smooth_knn)
        _lib.call("psg_smooth_knn_rooms", ctypes_off(x0, 3), 9, N * 9, runtime.ptr(ori), 3, N * 3, 1, N, int(neighbour),
                  runtime.ptr(scal[1:2]), runtime.ptr(sgrad), runtime.ptr(nn_state), 1 if step > 0 else 0, st())
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


def nu_attack_rooms(atk, images, labels, masks, target, neighbour, targeted_variant=False, trace=None, starts_fn=None):
    """R independent ONE-ROOM attacks advanced in lockstep: what `nu_attack` does when it is called once per room
    (`images[r:r+1]`, `labels[r:r+1]`, `masks[r]`), with one launch per operation for all rooms instead of R
    (psg_*_rooms entry points; the network kernels simply see a batch of R).  This is how BASELINE configs[2] is applied
    per room: the reference's tar_NU semantics are batch-of-one (target.py:62-133: Smooth term, exit
    accuracies and the mask all read batch row 0).

    Per room exactly the per-call arithmetic: its own mask row, Smooth term, f / Smooth / L2 sums, cost history, exit test
    and restart; a room that has left the loop keeps the image it returned and is skipped by the optimiser.  Rooms share
    the step counter, so all of them re-plan their geometry at the same steps ([0], [1..10], [11..20], ..: where a single
    call re-plans too).  Two things differ from R sequential calls and are why this is a separate entry: (1) the FPS
    start indices of a window are drawn for all rooms at once ([n_plan, 4, R] from the CPU generator) instead of room
    after room, and restart noise is drawn room by room at the step it happens - the same distributions, another order of
    consumption; (2) the reference halves `self.lr` every 50 steps and leaves it halved for the NEXT call
    (target.py:123-125), which makes later rooms depend on how long earlier ones ran: that cannot be advanced in lockstep,
    so more than 50 steps are refused here (call `nu_attack` per room).

    images [R, 9, N], labels [R, N], masks [R, N] bool (None for the non-targeted variant).  Returns (adv [R, 9, N],
    steps_run [R] int64 numpy: the optimiser steps each room executed).
    """
    net = psg_model(atk.model)
    dev = atk.device
    images = images.detach().to(dev).float().contiguous()
    R, C, N = images.shape
    if R < 2:
        raise ValueError("nu_attack_rooms advances several rooms in lockstep; call nu_attack for one room")
    if atk.steps > 50:
        raise ValueError("nu_attack_rooms: more than 50 steps would need the reference's learning-rate halving, whose state "
                         "leaks from one call into the next (target.py:123-125); call nu_attack per room")
    labels_d = labels_to_device(labels, dev)
    if masks is not None:
        mk = masks.detach().to(torch.bool).cpu().numpy() if isinstance(masks, torch.Tensor) else np.asarray(masks).astype(bool)
        if mk.shape != (R, N):
            raise ValueError("masks must be boolean [%d, %d], got shape %s" % (R, N, mk.shape))
        mask_d = torch.from_numpy(mk.astype(np.uint8)).to(dev)
        mask_b = mask_d.bool()
        n_mask = mk.sum(axis=1).astype(np.float64)
    else:
        if targeted_variant:
            raise ValueError("the targeted variant needs one mask per room")
        mask_d = mask_b = None
        n_mask = np.zeros(R)
    model = net._packed()
    net._generation += 1
    ws = net._workspace(R, N, CHUNK + 1)
    st = runtime.stream

    x0 = torch.empty(R, N, 9, device=dev, dtype=torch.float32)
    _lib.call("psg_to_point_major", runtime.ptr(images), R, 9, N, runtime.ptr(x0), st())
    ori = x0[:, :, 3:6].contiguous()
    x0_orig = x0.clone()
    w = torch.empty(R, N, 3, device=dev, dtype=torch.float32)
    _lib.call("psg_nu_inverse_tanh", runtime.ptr(x0), R, N, runtime.ptr(w), st())
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    dlogp = torch.empty(R, N, 13, device=dev, dtype=torch.float32)
    dx0 = torch.empty(R, N, 9, device=dev, dtype=torch.float32)
    sgrad = torch.empty(R, N, 3, device=dev, dtype=torch.float32)
    pred = torch.empty(R, N, device=dev, dtype=torch.int32)
    scal = torch.zeros(3, R, device=dev, dtype=torch.float32)      # rows: f, smooth, l2
    nn_state = torch.empty(R, N, int(neighbour), device=dev, dtype=torch.int32)   # Smooth term: last step's neighbours (start bound)
    active_d = torch.ones(R, device=dev, dtype=torch.uint8)
    active = np.ones(R, bool)
    steps_run = np.zeros(R, np.int64)
    extra_l2 = np.zeros(R)
    prev_cost = np.full((atk.steps, R), 1e10)
    lr, adam_t = float(atk.lr), 0
    tsign = float(atk._targeted)
    use_target = targeted_variant and target is not None
    out = torch.empty_like(images)
    planned_until = 0

    def snapshot(rooms):
        for r in rooms:
            out[r].copy_(x0[r].t())

    for step in range(atk.steps):
        # (rooms that are done keep their w: their colours are rewritten with the same values)
        _lib.call("psg_nu_tanh_color_rooms", runtime.ptr(w), runtime.ptr(mask_d), R, N, runtime.ptr(x0), st())
        if step >= planned_until:
            window_end = 1 if step == 0 else ((step - 1) // CHUNK + 1) * CHUNK + 1      # [0], [1..10], [11..20], ..
            n_plan = min(window_end - step, atk.steps - step)
            starts = (draw_fps_starts(R, N, n_plan) if starts_fn is None else starts_fn(step, n_plan)).to(dev)
            ws.plan_build(x0, starts, n_plan)
            plan_base, planned_until = step, step + n_plan
        slot = step - plan_base
        logp = ws.forward(model, slot, x0)
        scal.zero_()
        _lib.call("psg_nu_f_loss_grad_rooms", runtime.ptr(logp), None if use_target else runtime.ptr(labels_d),
                  int(target) if use_target else 0, R, N, 13, float(atk.kappa), tsign, runtime.ptr(dlogp),
                  runtime.ptr(scal[0]), runtime.ptr(pred), st())
        ws.backward(model, slot, dlogp, dx0)
        _lib.call("psg_smooth_knn_rooms", ctypes_off(x0, 3), 9, N * 9, runtime.ptr(ori), 3, N * 3, R, N, int(neighbour),
                  runtime.ptr(scal[1]), runtime.ptr(sgrad), runtime.ptr(nn_state), 1 if step > 0 else 0, st())
        adam_t += 1
        _lib.call("psg_nu_adam_step_rooms", runtime.ptr(w), runtime.ptr(m), runtime.ptr(v), runtime.ptr(mask_d),
                  runtime.ptr(dx0), runtime.ptr(x0), runtime.ptr(ori), runtime.ptr(sgrad), float(atk.c), float(atk.c),
                  lr, BETA1, BETA2, ADAM_EPS, adam_t, R, N, runtime.ptr(active_d), runtime.ptr(scal[2]), st())
        # ---- the reference's control flow, per room (one read-back per step for all rooms)
        correct = pred.eq(labels_d)
        n_correct_d = correct.sum(dim=1).float()
        if targeted_variant:
            tgt_hits = ((pred.eq(int(target)) if use_target else correct) & mask_b).sum(dim=1).float()
        else:
            tgt_hits = n_correct_d
        stats = torch.cat([n_correct_d[None], tgt_hits[None], scal]).cpu().numpy().astype(np.float64)   # [5, R]
        n_correct, n_tgt, f_loss, sm_loss = stats[0], stats[1], stats[2], stats[3]
        l2_loss = stats[4] + extra_l2
        cost = f_loss + float(atk.c) * sm_loss + float(atk.c) * l2_loss
        prev_cost[step] = np.where(active, cost, prev_cost[step])
        steps_run[active] += 1
        if trace is not None:
            trace(step=step, cost=cost, f=f_loss, smooth=sm_loss, l2=l2_loss, w=w, m=m, v=v, dx0=dx0, x0=x0, pred=pred,
                  active=active.copy())
        if not targeted_variant:
            done = active & (n_correct / 4096 < 1 / 13)                     # nontarget.py:87,95-96
        else:
            with np.errstate(divide="ignore", invalid="ignore"):
                target_acc = n_tgt / n_mask
            done = active & ((target_acc > 0.9) if use_target else (target_acc < 1 / 13))   # target.py:116-121
        if done.any():
            snapshot(np.nonzero(done)[0])
            active &= ~done
            active_d.copy_(torch.from_numpy(active.astype(np.uint8)))
            if not active.any():
                return out, steps_run
        if targeted_variant and step > 10 and step % 10 == 0:               # target.py:127-132, room by room
            again = np.nonzero(active & (cost >= prev_cost[step - 10]))[0]
            for r in again:                                                  # the noise draws stay per room, in room order
                k = int(n_mask[r])
                noise = torch.empty(1, 3, k, device=dev, dtype=torch.float32).uniform_(0, 1)
                col = x0[r:r + 1, :, 3:6].transpose(1, 2)                     # view [1, 3, N]
                col[:, :, mask_b[r]] = col[:, :, mask_b[r]] + noise
            if len(again):
                idx = torch.from_numpy(again).to(dev)
                clamped = x0[idx].clamp_(min=0, max=1)                       # ALL channels, like the reference
                x0[idx] = clamped
                d = clamped - x0_orig[idx]
                sums = torch.stack([(d[:, :, 0:3] ** 2).sum(dim=(1, 2)), (d[:, :, 6:9] ** 2).sum(dim=(1, 2))]).cpu().numpy()
                extra_l2[again] = sums[0].astype(np.float64) + sums[1].astype(np.float64)   # one read-back for all rooms
            # (xyz may have moved: the next window starts with the next step anyway - restarts only follow steps 20, 30, ..)
    snapshot(np.nonzero(active)[0])
    return out, steps_run


def ctypes_off(t, n_floats):
    """Device pointer to element `n_floats` of a float32 tensor."""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * n_floats)
