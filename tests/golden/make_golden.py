"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself (build container only).

Usage (from the repo root, needs /root/reference, never runs on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [weights|room|nb|tarnb|all]

Outputs are plain numeric .npz files (inputs + expected outputs); no reference source is copied.
  pn2_weights.npz   state_dict of the reference get_model(13) after a short supervised fit on
                    rule-labelled synthetic rooms (default-init weights predict a single class)
  pn2_room.npz      one room: FPS / ball-query / 3-NN indices, interpolation weights, raw
                    square_distance bit patterns, per-layer activations, log-probs, d cost/d colour
  pn2_nb.npz        NB_attack(eps=.05, alpha=2/255, iters=40), B=2: the colour state the attack feeds
                    to the model at selected iterations (recorded through an instrumented model
                    argument; the attack code itself is unmodified), the returned adversarial colours,
                    the recorded torch.randint stream, clean/adv log-probs and metrics
  pn2_tarnb.npz     tar_NB_attack(eps=.5, alpha=.1, iters=10, target=6, mask=label==11), B=2, same layout
  pn2_nu.npz        NU_attack(c=.1, lr=.01, 6 steps), B=1: per Adam step w before, gradient, w/m/v after, cost
                    (recorded by instrumenting torch.optim.Adam.step / Tensor.backward / torch.randint)
  pn2_tarnu.npz     tar_NU_attack(c=1, lr=.01, 23 steps, target=6, mask=label==11): same, steps 0-2 and 19-22
                    (around the first restart at step 20)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/PointNet"
sys.path[:0] = [ROOT, REF, REF + "/models", REF + "/attacks"]
sys.dont_write_bytecode = True

from pointsecguard_amd.synthetic import make_rooms, rule_labels  # noqa: E402

from models.pointnet2_sem_seg import get_model  # noqa: E402  (reference)
import torchattacks  # noqa: E402  (reference)

LEVEL_N = (4096, 1024, 256, 64)


def load_model():
    sd = np.load(os.path.join(HERE, "pn2_weights.npz"))
    m = get_model(13)
    m.load_state_dict({k: torch.from_numpy(sd[k]) for k in sd.files})
    return m.eval()


def draw_starts(seed, n_forward, batch):
    """Replay of the CPU-generator stream: 4 torch.randint draws per forward (pointnet_util.py:75)."""
    torch.manual_seed(seed)
    out = np.zeros((n_forward, 4, batch), np.int32)
    for f in range(n_forward):
        for lvl, n in enumerate(LEVEL_N):
            out[f, lvl] = torch.randint(0, n, (batch,), dtype=torch.long).numpy()
    return out


def fit_weights(steps=400):
    torch.manual_seed(1234)
    torch.set_num_threads(8)
    m = get_model(13).train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    t0 = time.time()
    for step in range(steps):
        rooms = make_rooms(4, 10_000 + step)
        y = torch.from_numpy(rule_labels(rooms))
        x = torch.from_numpy(rooms).transpose(2, 1).contiguous()
        logp, _ = m(x)
        loss = torch.nn.functional.nll_loss(logp.reshape(-1, 13), y.reshape(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 20 == 0 or step == steps - 1:
            acc = (logp.argmax(2) == y).float().mean().item()
            print("fit step %d loss %.4f acc %.3f ncls %d  %.0fs" % (
                step, loss.item(), acc, logp.argmax(2).unique().numel(), time.time() - t0), flush=True)
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "pn2_weights.npz"), **sd)


def metrics(pred, gt):
    """acc + per-batch mIoU of NB_nontarget_test_semseg.py:188-211."""
    acc = float((pred == gt).sum()) / pred.size
    inter = np.array([np.sum((pred == l) & (gt == l)) for l in range(13)])
    union = np.array([np.sum((pred == l) | (gt == l)) for l in range(13)])
    seen = np.array([np.sum(gt == l) for l in range(13)])
    iou = inter / (union.astype(np.float64) + 1e-6)
    return acc, float(np.mean(iou[seen != 0])), inter, union, seen


def gen_room():
    torch.set_num_threads(1)
    m = load_model()
    seed_room, seed_rng = 7, 0
    room = make_rooms(1, seed_room)
    labels = rule_labels(room)
    x = torch.from_numpy(room).transpose(2, 1).contiguous()
    starts = draw_starts(seed_rng, 1, 1)[0, :, 0]
    out = {"room": room[0], "labels": labels[0].astype(np.int16), "starts": starts}

    # geometry, straight from the reference functions
    from models import pointnet_util as pu
    xyz = torch.from_numpy(room[:, :, :3].copy())
    cfg = ((1024, 0.1), (256, 0.2), (64, 0.4), (16, 0.8))
    torch.manual_seed(seed_rng)
    lv = [xyz]
    for lvl, (npoint, radius) in enumerate(cfg):
        fi = pu.farthest_point_sample(lv[lvl], npoint)
        assert int(fi[0, 0]) == starts[lvl]
        new_xyz = pu.index_points(lv[lvl], fi)
        gi = pu.query_ball_point(radius, 32, lv[lvl], new_xyz)
        out["fps%d" % lvl] = fi[0].numpy().astype(np.int16)
        out["group%d" % lvl] = gi[0].numpy().astype(np.int16)
        lv.append(new_xyz)
    for lvl in range(4):
        d = pu.square_distance(lv[lvl], lv[lvl + 1])
        ds, idx = d.sort(dim=-1)
        ds, idx = ds[:, :, :3], idx[:, :, :3]
        rc = 1.0 / (ds + 1e-8)
        w = rc / torch.sum(rc, dim=2, keepdim=True)
        out["nn_idx%d" % lvl] = idx[0].numpy().astype(np.int16)
        out["nn_w%d" % lvl] = w[0].numpy()
    out["sqd_src"] = lv[2][0, :64].numpy()
    out["sqd_dst"] = lv[1][0, :256].numpy()
    out["sqd_bits"] = pu.square_distance(lv[2][:, :64], lv[1][:, :256])[0].numpy().view(np.uint32)

    # network forward with hooks (same RNG stream -> same geometry), then d cost / d colour
    acts = {}
    hooks = []
    for name in ("sa1", "sa2", "sa3", "sa4", "fp4", "fp3", "fp2", "fp1"):
        def hook(mod, inp, outp, name=name):
            t = outp[1] if isinstance(outp, tuple) else outp
            acts[name] = t.detach()[0].T.contiguous().numpy()
        hooks.append(getattr(m, name).register_forward_hook(hook))
    color = x[:, 3:6].clone().requires_grad_(True)
    adv = x.clone()
    adv[:, 3:6] = color
    torch.manual_seed(seed_rng)
    logp, l4 = m(adv)
    y = torch.from_numpy(labels)
    cost = torch.nn.CrossEntropyLoss(reduction="sum")(logp.reshape(-1, 13), y.reshape(-1)) / logp.size(1)
    cost.backward()
    for h in hooks:
        h.remove()
    for name, a in acts.items():
        out["act_" + name] = a[::16] if name == "fp1" else a  # fp1 is [4096,128]: keep every 16th point
    out["logp"] = logp.detach()[0].numpy()
    out["cost"] = np.float64(cost.item())
    out["dcolor"] = color.grad[0].T.contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, "pn2_room.npz"), **out)
    print("room: cost %.6f, |g|max %.3e, nonzero %d" % (cost.item(), color.grad.abs().max().item(),
                                                       int((color.grad != 0).sum())))


class Recorder(torch.nn.Module):
    """Instrumented model argument for the UNMODIFIED reference attacks: records the colour channels
    the attack feeds to the model at every iteration (= the projected colour state entering it)."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.seen = []
        self.seen_full = []

    def forward(self, x):
        self.seen.append(x.detach()[:, 3:6].clone().numpy())
        self.seen_full.append(x.detach().clone().numpy())
        return self.inner(x)


NB_KEEP = (0, 1, 2, 3, 5, 6, 10, 11, 20, 21, 39)


def gen_nb():
    torch.set_num_threads(1)
    m = load_model()
    B, seed_room, seed_rng, iters = 2, 21, 3, 40
    eps, alpha = 0.05, 2 / 255
    room = make_rooms(B, seed_room)
    labels = rule_labels(room)
    x = torch.from_numpy(room).transpose(2, 1).contiguous()
    out = {"rooms": room, "labels": labels.astype(np.int16), "eps": eps, "alpha": alpha, "iters": iters,
           "starts": draw_starts(seed_rng, iters + 2, B), "seed_rng": seed_rng}
    # RNG stream: [clean forward] + `iters` attack forwards + [adversarial forward]
    torch.manual_seed(seed_rng)
    with torch.no_grad():
        clean_logp, _ = m(x)
    rec = Recorder(m).eval()
    atk = torchattacks.NB_attack(rec, eps=eps, alpha=alpha, iters=iters)
    adv = atk(x, labels.astype(np.float64)).detach()
    with torch.no_grad():
        adv_logp, _ = m(adv)
    assert len(rec.seen) == iters
    for t in NB_KEEP:
        out["state_it%d" % t] = rec.seen[t]          # colour entering attack iteration t (after t steps)
    out["adv_color_final"] = adv[:, 3:6].numpy()      # un-projected last step (reference return value)
    out["clean_logp"] = clean_logp.numpy()
    out["adv_logp"] = adv_logp.numpy()
    pred, apred = clean_logp.argmax(2).numpy(), adv_logp.argmax(2).numpy()
    acc, miou, inter, union, seen = metrics(pred, labels)
    aacc, amiou, ainter, aunion, _ = metrics(apred, labels)
    out.update(acc=acc, miou=miou, adv_acc=aacc, adv_miou=amiou, inter=inter, union=union, seen=seen,
               adv_inter=ainter, adv_union=aunion, l2_dis=float(torch.dist(adv, x).item()))
    np.savez_compressed(os.path.join(HERE, "pn2_nb.npz"), **out)
    print("nb: acc %.4f -> %.4f, miou %.4f -> %.4f" % (acc, aacc, miou, amiou))


TAR_KEEP = (0, 1, 2, 5, 6, 9)


def gen_tarnb():
    torch.set_num_threads(1)
    m = load_model()
    seed_room, seed_rng, iters = 33, 5, 10
    eps, alpha, target, origin = 0.5, 0.1, 6, 11
    room = make_rooms(2, seed_room)  # two rooms: the loss uses batch row 0 only (target.py:36)
    labels = rule_labels(room)
    mask = labels[0] == origin
    x = torch.from_numpy(room).transpose(2, 1).contiguous()
    out = {"rooms": room, "labels": labels.astype(np.int16), "mask": mask, "eps": eps, "alpha": alpha,
           "target": target, "iters": iters, "starts": draw_starts(seed_rng, iters, 2), "seed_rng": seed_rng}
    torch.manual_seed(seed_rng)
    rec = Recorder(m).eval()
    atk = torchattacks.tar_NB_attack(rec, eps=eps, alpha=alpha, iters=iters, target=target, mask=mask)
    adv = atk(x, labels.astype(np.float64)).detach()
    for t in TAR_KEEP:
        out["state_it%d" % t] = rec.seen[t]
    out["adv_color_final"] = adv[:, 3:6].numpy()
    with torch.no_grad():
        adv_logp, _ = m(adv)
    pred = adv_logp.argmax(2).numpy()
    out["target_acc"] = float((pred[0][mask] == target).sum()) / float(mask.sum())  # NB_target_test_semseg.py:190
    np.savez_compressed(os.path.join(HERE, "pn2_tarnb.npz"), **out)
    print("tarnb: mask count", int(mask.sum()), "target_acc", out["target_acc"])


class Instrument:
    """Records, around the UNMODIFIED reference NU attacks, what their internals compute: every
    torch.randint draw (FPS starts), every Adam step (w before, grad, w/m/v after) and the scalar
    each .backward() is called on (the attack cost).  Only torch itself is patched."""

    def __init__(self):
        self.randints, self.adam, self.costs = [], [], []

    def __enter__(self):
        self._randint, self._step, self._backward = torch.randint, torch.optim.Adam.step, torch.Tensor.backward
        inst = self

        def randint(*a, **k):
            out = inst._randint(*a, **k)
            inst.randints.append(out.numpy().copy())
            return out

        def step(opt, *a, **k):
            p = opt.param_groups[0]["params"][0]
            rec = {"w_before": p.detach().numpy().copy(), "grad": p.grad.detach().numpy().copy(),
                   "lr": opt.param_groups[0]["lr"]}
            r = inst._step(opt, *a, **k)
            st = opt.state[p]
            rec.update(w_after=p.detach().numpy().copy(), m=st["exp_avg"].numpy().copy(),
                       v=st["exp_avg_sq"].numpy().copy(), t=int(st["step"]))
            inst.adam.append(rec)
            return r

        def backward(t, *a, **k):
            if t.dim() == 0:
                inst.costs.append(float(t.item()))
            return inst._backward(t, *a, **k)

        torch.randint, torch.optim.Adam.step, torch.Tensor.backward = randint, step, backward
        return self

    def __exit__(self, *exc):
        torch.randint, torch.optim.Adam.step, torch.Tensor.backward = self._randint, self._step, self._backward

    def starts(self, batch):
        """[n_forward][4][batch] from the recorded randint stream."""
        r = [x for x in self.randints if x.shape == (batch,)]
        n = len(r) // 4
        return np.stack([np.stack(r[4 * f:4 * f + 4]) for f in range(n)]).astype(np.int32)


def _pack_adam(out, inst, keep):
    for t in keep:
        rec = inst.adam[t]
        for k in ("w_before", "grad", "w_after", "m", "v"):
            out["s%d_%s" % (t, k)] = rec[k]
        out["s%d_lr" % t] = rec["lr"]
        out["s%d_t" % t] = rec["t"]
    out["costs"] = np.array(inst.costs, np.float64)


def gen_nu():
    torch.set_num_threads(1)
    m = load_model()
    seed_room, seed_rng, steps = 41, 8, 6
    c, kappa, lr = 0.1, 0, 0.01
    room = make_rooms(1, seed_room)
    labels = rule_labels(room)
    x = torch.from_numpy(room).transpose(2, 1).contiguous()
    torch.manual_seed(seed_rng)
    rec = Recorder(m).eval()
    with Instrument() as inst:
        adv = torchattacks.NU_attack(rec, c=c, kappa=kappa, steps=steps, lr=lr)(x, labels.astype(np.float64)).detach()
    out = {"rooms": room, "labels": labels.astype(np.int16), "c": c, "kappa": kappa, "lr": lr, "steps": steps,
           "starts": inst.starts(1), "adv_final": adv.numpy(), "n_steps_run": len(inst.adam)}
    _pack_adam(out, inst, range(len(inst.adam)))
    np.savez_compressed(os.path.join(HERE, "pn2_nu.npz"), **out)
    print("nu: steps run", len(inst.adam), "costs", inst.costs)


def gen_tarnu():
    torch.set_num_threads(1)
    m = load_model()
    seed_room, seed_rng, steps = 33, 6, 43
    c, kappa, lr, target, origin = 1, 0, 0.01, 6, 11
    room = make_rooms(1, seed_room)
    labels = rule_labels(room)
    mask = labels[0] == origin
    x = torch.from_numpy(room).transpose(2, 1).contiguous()
    torch.manual_seed(seed_rng)
    rec = Recorder(m).eval()
    with Instrument() as inst:
        atk = torchattacks.tar_NU_attack(rec, c=c, kappa=kappa, steps=steps, lr=lr, target=target, mask=mask)
        adv = atk(x, labels.astype(np.float64)).detach()
    keep = [t for t in (0, 1, 2, 19, 20, 21, 29, 30, 31, 39, 40, 41) if t < len(inst.adam)]
    out = {"rooms": room, "labels": labels.astype(np.int16), "mask": mask, "c": c, "kappa": kappa, "lr": lr,
           "steps": steps, "target": target, "starts": inst.starts(1), "adv_final": adv.numpy(),
           "n_steps_run": len(inst.adam), "keep": np.array(keep),
           "xyz_step21": rec.seen_full[21] if len(rec.seen_full) > 21 else np.zeros(0)}
    _pack_adam(out, inst, keep)
    np.savez_compressed(os.path.join(HERE, "pn2_tarnu.npz"), **out)
    print("tarnu: steps run", len(inst.adam), "mask", int(mask.sum()), "costs", np.round(inst.costs, 2).tolist())


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("weights", "all"):
        fit_weights()
    if what in ("room", "all"):
        gen_room()
    if what in ("nb", "all"):
        gen_nb()
    if what in ("tarnb", "all"):
        gen_tarnb()
    if what in ("nu", "all"):
        gen_nu()
    if what in ("tarnu", "all"):
        gen_tarnu()
