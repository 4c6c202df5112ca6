"""cProfile of one tar_NU attack call (32 rooms, 60 steps): which host lines cost time."""
import cProfile, pstats, os, sys, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointsecguard_amd.attacks import torchattacks
from pointsecguard_amd.attacks.torchattacks.attacks import nu as nu_mod
from pointsecguard_amd.models.pointnet2_sem_seg import get_model
from pointsecguard_amd.synthetic import make_rooms, rule_labels
sd = dict(np.load("tests/golden/pn2_weights.npz"))
net = get_model(13); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net = net.cuda().eval()
rooms = make_rooms(32, 7000, structured=True); labels = rule_labels(rooms)
images = torch.from_numpy(np.ascontiguousarray(rooms.transpose(0, 2, 1))).cuda()
mask = labels[0] == 2
def run(steps):
    atk = torchattacks.tar_NU_attack(net, c=1, kappa=0, steps=steps, lr=0.01, target=None, mask=mask)
    nu_mod.nu_attack(atk, images, labels.astype(np.float64), mask, None, 5, targeted_variant=True)
    torch.cuda.synchronize()
run(12)
pr = cProfile.Profile(); pr.enable(); run(60); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
