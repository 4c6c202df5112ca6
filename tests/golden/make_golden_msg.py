"""Generate tests/golden/pn2msg_room.npz and pn2msg_nb.npz by running the REFERENCE MSG network itself
(build container only; never runs on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_msg.py

Weights are NOT stored: both sides rebuild them from pointsecguard_amd.synthetic.msg_state_dict(MSG_SEED); here they
are loaded into the reference's get_model with load_state_dict(strict=True), which also checks the key layout.
  pn2msg_room.npz  one room: FPS indices per level, ball-query groups per (level, scale), level outputs l1..l4,
                   fp4..fp2 outputs, log-probs, cost and d cost / d colour
  pn2msg_nb.npz    NB_attack(eps=.05, alpha=2/255, iters=6), B=2 through the unmodified reference attack: colour
                   state entering every iteration, returned adversarial colours, clean / adversarial log-probs
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/PointNet"
sys.path[:0] = [ROOT, REF, REF + "/models", REF + "/attacks"]
sys.dont_write_bytecode = True

from pointsecguard_amd.synthetic import make_rooms, msg_state_dict, rule_labels  # noqa: E402

from models.pointnet2_sem_seg_msg import get_model  # noqa: E402  (reference)
from models import pointnet_util as pu  # noqa: E402  (reference)
import torchattacks  # noqa: E402  (reference)

MSG_SEED = 77
LEVEL_N = (4096, 1024, 256, 64)
CFG = ((1024, (0.05, 0.1), (16, 32)), (256, (0.1, 0.2), (16, 32)), (64, (0.2, 0.4), (16, 32)), (16, (0.4, 0.8), (16, 32)))


def load_model():
    m = get_model(13)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in msg_state_dict(MSG_SEED).items()}, strict=True)
    return m.eval()


def draw_starts(seed, n_forward, batch):
    torch.manual_seed(seed)
    out = np.zeros((n_forward, 4, batch), np.int32)
    for f in range(n_forward):
        for lvl, n in enumerate(LEVEL_N):
            out[f, lvl] = torch.randint(0, n, (batch,), dtype=torch.long).numpy()
    return out


def gen_room():
    torch.set_num_threads(1)
    m = load_model()
    seed_room, seed_rng = 9, 5
    room = make_rooms(1, seed_room)
    labels = rule_labels(room)
    x = torch.from_numpy(room).transpose(2, 1).contiguous()
    starts = draw_starts(seed_rng, 1, 1)[0, :, 0]
    out = {"room": room[0], "labels": labels[0].astype(np.int16), "starts": starts, "msg_seed": MSG_SEED}
    xyz = torch.from_numpy(room[:, :, :3].copy())
    torch.manual_seed(seed_rng)
    lv = [xyz]
    for lvl, (npoint, radii, ks) in enumerate(CFG):
        fi = pu.farthest_point_sample(lv[lvl], npoint)
        assert int(fi[0, 0]) == starts[lvl]
        new_xyz = pu.index_points(lv[lvl], fi)
        out["fps%d" % lvl] = fi[0].numpy().astype(np.int16)
        for i, (r, k) in enumerate(zip(radii, ks)):
            out["group%d_%d" % (lvl, i)] = pu.query_ball_point(r, k, lv[lvl], new_xyz)[0].numpy().astype(np.int16)
        lv.append(new_xyz)
    acts = {}
    hooks = []
    for name in ("sa1", "sa2", "sa3", "sa4", "fp4", "fp3", "fp2"):
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
        out["act_" + name] = a
    out["logp"] = logp.detach()[0].numpy()
    out["l4"] = l4.detach()[0].numpy()
    out["cost"] = np.float64(cost.item())
    out["dcolor"] = color.grad[0].T.contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, "pn2msg_room.npz"), **out)
    print("room: cost %.6f |g|max %.3e nonzero %d classes %d" % (
        cost.item(), color.grad.abs().max().item(), int((color.grad != 0).sum()), logp.argmax(2).unique().numel()))


class Recorder(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.seen = []

    def forward(self, x):
        self.seen.append(x.detach()[:, 3:6].clone().numpy())
        return self.inner(x)


def gen_nb():
    torch.set_num_threads(4)
    m = load_model()
    B, seed_room, seed_rng, iters = 2, 23, 11, 6
    eps, alpha = 0.05, 2 / 255
    room = make_rooms(B, seed_room)
    labels = rule_labels(room)
    x = torch.from_numpy(room).transpose(2, 1).contiguous()
    out = {"rooms": room, "labels": labels.astype(np.int16), "eps": eps, "alpha": alpha, "iters": iters,
           "starts": draw_starts(seed_rng, iters + 2, B), "seed_rng": seed_rng, "msg_seed": MSG_SEED}
    torch.manual_seed(seed_rng)
    with torch.no_grad():
        clean_logp, _ = m(x)
    rec = Recorder(m).eval()
    atk = torchattacks.NB_attack(rec, eps=eps, alpha=alpha, iters=iters)
    adv = atk(x, labels.astype(np.float64)).detach()
    with torch.no_grad():
        adv_logp, _ = m(adv)
    assert len(rec.seen) == iters
    for t in range(iters):
        out["state_it%d" % t] = rec.seen[t]
    out["adv_color_final"] = adv[:, 3:6].numpy()
    out["clean_logp"] = clean_logp.numpy()
    out["adv_logp"] = adv_logp.numpy()
    np.savez_compressed(os.path.join(HERE, "pn2msg_nb.npz"), **out)
    print("nb: clean acc %.3f adv acc %.3f" % ((clean_logp.argmax(2).numpy() == labels).mean(),
                                               (adv_logp.argmax(2).numpy() == labels).mean()))


if __name__ == "__main__":
    gen_room()
    gen_nb()
