"""GPU box: us per 4-room call of the prefilter kNN kernel on the fitted ResGCN-28's own block inputs, for the library named by
PSG_LIBRARY_OVERRIDE (timing-only variant builds: tools/build_variant.sh ablN psg_knn "-DKB_ABLATE=N": 1 = final ranking ends
after the collect pass, 2 = after the window search, 3 = no final ranking at all, 4 = no rank pass; their tables are wrong)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PSG_GCN_KNN_BF_MAXD", "27")
from pointsecguard_amd import runtime
from pointsecguard_amd.synthetic import gcn28_state_dict, make_rooms
B, N = 4, 4096
model = runtime.GCNModel(gcn28_state_dict(), 28)
x0 = torch.from_numpy(np.ascontiguousarray(make_rooms(B, 5000))).cuda()
ws = runtime.GCNWorkspace(B, N, 28)
ws.forward(model, x0)
feats = ws.feats()
out = []
for e in [int(v) for v in os.environ.get("BLOCKS", "1,6,12,20,27").split(",")]:
    x = feats[:, :, 64 * (e - 1):64 * e].contiguous()
    ws.knn(x, e)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        ws.knn(x, e)
    t1.record(); torch.cuda.synchronize()
    out.append("d=%d %.1f" % (e, t0.elapsed_time(t1) * 50))
print(os.environ.get("PSG_LIBRARY_OVERRIDE", "default library") + ": us per call  " + "  ".join(out), flush=True)
