"""Feasibility probe (GPU box): could block e's kNN start from a row threshold derived from block e - 1's graph?
For blocks e >= 2 of the fitted ResGCN-28: tau0(i) = c * max_k |x[i] - x[nbr_{e-1}(i, k)]|^2 over the 16 dilated neighbours
block e - 1 found, evaluated in block e's input features x; per c the share of rows with fewer than KK = 15 e + 1 candidates
below tau0 (a warm start would have to redo those) and the median / 99th percentile of count / KK (rows above CAP / KK
overflow their buffer and need a cut anyway)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import runtime
from pointsecguard_amd.synthetic import gcn28_state_dict, make_rooms

B, N = 2, 4096
model = runtime.GCNModel(gcn28_state_dict(), 28)
structured = os.environ.get("ROOMS", "uniform") == "structured"
rooms = make_rooms(B, 5000, structured=structured)
x0 = torch.from_numpy(np.ascontiguousarray(rooms)).cuda()
ws = runtime.GCNWorkspace(B, N, 28)
ws.forward(model, x0)
feats = ws.feats()                                                   # [B, N, 64 * 28]
cs = [float(v) for v in os.environ.get("CS", "1.0,1.2,1.5,2.0,3.0").split(",")]
print("block  KK | " + " | ".join("c=%.1f: under%%  med  p99 (count/KK)" % c for c in cs))
for e in [int(v) for v in os.environ.get("BLOCKS", "2,3,5,9,14,20,27").split(",")]:
    x = feats[:, :, 64 * (e - 1):64 * e]                             # input of block e
    prev = ws.edges(e - 1).long()                                    # [B, N, 16] graph of block e - 1 (room-local)
    KK = 15 * e + 1
    line = "%5d %4d | " % (e, KK)
    for c in cs:
        under, ratios = 0, []
        for b in range(B):
            xb = x[b]
            sq = (xb * xb).sum(-1)
            D = (sq[:, None] - 2.0 * xb @ xb.t()) + sq[None, :]      # [N, N]
            dn = torch.gather(D, 1, prev[b])                         # distances to the previous block's neighbours
            tau = c * dn.max(dim=1).values
            cnt = (D <= tau[:, None]).sum(dim=1)
            under += int((cnt < KK).sum())
            ratios.append((cnt.float() / KK).cpu().numpy())
        r = np.concatenate(ratios)
        line += "%5.2f %5.2f %5.2f | " % (100.0 * under / (B * N), np.median(r), np.percentile(r, 99))
    print(line, flush=True)
