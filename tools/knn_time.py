"""Timing (GPU box): the stand-alone dilated kNN entry point (prep + fused kernel) per dilation, normal build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import runtime
rng = np.random.default_rng(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
f = torch.from_numpy(rng.standard_normal((B, 4096, 64)).astype(np.float32)).cuda()
ws = runtime.GCNWorkspace(B, 4096, 28)
out = []
for d in [int(x) for x in os.environ.get("KNN_D", "1,4,9,17,27").split(",")]:
    ws.knn(f, d)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        ws.knn(f, d)
    t1.record(); torch.cuda.synchronize()
    out.append("d=%d %.1f" % (d, t0.elapsed_time(t1) * 50))
print("us per knn call (B=%d): " % B + "  ".join(out), flush=True)
