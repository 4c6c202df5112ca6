"""Feasibility probe (GPU box): a RIGOROUS warm start of block e's kNN from the previous PGD iteration's top-KK set.
Two consecutive NB iterations of the fitted ResGCN-28 (eps = 0.3, alpha = 2/255: the harness values): for block e the set S
of the KK = 15 e + 1 nearest candidates of every point at iteration t; at iteration t + 1 the bound b_i = max_{j in S_i} D_ij is
an upper bound of the KK-th smallest distance (S_i holds KK distinct candidates), so a kernel that starts from it needs no open
phase and cannot run short.  Reported: count(D <= b) / KK per row - how many keys such a row would carry (1.0 = nothing to cut)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import runtime
from pointsecguard_amd.synthetic import gcn28_state_dict, make_rooms, rule_labels

B, N = 1, 4096
model = runtime.GCNModel(gcn28_state_dict(), 28)
rooms = make_rooms(B, 5000, structured=os.environ.get("ROOMS", "uniform") == "structured")
labels = torch.from_numpy(rule_labels(rooms).astype(np.int64)).cuda()
x = torch.from_numpy(np.ascontiguousarray(rooms)).cuda()
ws = runtime.GCNWorkspace(B, N, 28)
it0 = int(os.environ.get("IT0", "0"))
feats = []
for it in range(it0 + 2):
    logits = ws.forward(model, x)
    if it >= it0:
        feats.append(ws.feats().clone())
    z = logits.detach().clone().requires_grad_(True)
    torch.nn.functional.cross_entropy(z.reshape(-1, 13), labels.reshape(-1), reduction="mean").backward()
    dx0 = ws.backward(model, z.grad.contiguous())
    x = x.clone()
    x[:, :, 3:6] = (x[:, :, 3:6] + (2.0 / 255.0) * dx0[:, :, 3:6].sign()).clamp(0, 1)
print("iterations %d -> %d" % (it0, it0 + 1))
print("block   KK |  count / KK: median   p90    p99    max  | rows above 1024 keys %")
for e in [int(v) for v in os.environ.get("BLOCKS", "1,2,3,5,9,14,20,27").split(",")]:
    KK = 15 * e + 1
    def dist(f):
        xb = f[0, :, 64 * (e - 1):64 * e]
        sq = (xb * xb).sum(-1)
        return (sq[:, None] - 2.0 * xb @ xb.t()) + sq[None, :]
    D0, D1 = dist(feats[0]), dist(feats[1])
    S = torch.topk(D0, KK, dim=1, largest=False).indices
    b = torch.gather(D1, 1, S).max(dim=1).values
    cnt = (D1 <= b[:, None]).sum(dim=1).float()
    r = (cnt / KK).cpu().numpy()
    print("%5d %4d |             %6.2f %6.2f %6.2f %6.2f  | %5.2f" % (e, KK, np.median(r), np.percentile(r, 90), np.percentile(r, 99), r.max(),
                                                                     100.0 * float((cnt > 1024).float().mean())), flush=True)
