"""Diagnostic (GPU box): error distribution of the teacher-forced 28-block forward against the reference fixture."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import runtime
from pointsecguard_amd.synthetic import gcn28_state_dict
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gcn28_room.npz")))
model, ws = runtime.GCNModel(gcn28_state_dict(), 28), runtime.GCNWorkspace(1, 4096, 28)
ws.set_graphs(torch.from_numpy(g["graphs"].astype(np.int32)[:, None]).cuda().contiguous())
logits = ws.forward(model, torch.from_numpy(g["room"][None]).cuda())
feats = ws.feats()[0].cpu().numpy()
for e in (0, 1, 14, 27):
    ref = g["feat%d" % e]; err = np.abs(feats[:, 64 * e:64 * e + 64] - ref) / np.abs(ref).max()
    print("feat%d: max|ref| %.2f rel err max %.2e p99.99 %.2e p99.9 %.2e median %.2e; entries > 2e-4: %d (rows %d)" % (
        e, np.abs(ref).max(), err.max(), np.quantile(err, 0.9999), np.quantile(err, 0.999), np.median(err), (err > 2e-4).sum(), (err > 2e-4).any(1).sum()))
ref = g["logits"]; err = np.abs(logits[0].cpu().numpy() - ref) / np.abs(ref).max()
print("logits: max|ref| %.2f rel err max %.2e p99.9 %.2e" % (np.abs(ref).max(), err.max(), np.quantile(err, 0.999)))
# free-running: graph overlap with the reference's tables per block
ws.set_graphs(None)
ws.forward(model, torch.from_numpy(g["room"][None]).cuda())
torch.cuda.synchronize()
feats2 = ws.feats()[0].cpu().numpy()
line = []
for e in range(28):
    got = ws.edges(e)[0].cpu().numpy(); ref = g["graphs"][e]
    line.append("%d:%.4f" % (e, np.mean([len(set(a) & set(b)) / 16.0 for a, b in zip(got, ref)])))
print("free-running edge overlap per block:", " ".join(line))
for e in (1, 14, 27):
    ref = g["feat%d" % e]; err = np.abs(feats2[:, 64 * e:64 * e + 64] - ref) / np.abs(ref).max()
    print("free-running feat%d rel err max %.2e p99 %.2e median %.2e" % (e, err.max(), np.quantile(err, 0.99), np.median(err)))
