"""GPU box: the bf16-prefilter kNN kernel against the exact fused kernel (PSG_GCN_KNN=f32) on the same features - equal
tables, counters of the prefilter kernel (exact-path tiles, finalists per row, cuts) and time per call."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointsecguard_amd import runtime

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kind = sys.argv[3] if len(sys.argv) > 3 else "gauss"
rng = np.random.default_rng(3)
if kind == "gauss":
    f = (rng.standard_normal((B, N, 64)) * rng.uniform(0.2, 3.0, (1, 1, 64))).astype(np.float32)
elif kind == "relu":
    f = np.maximum(rng.standard_normal((B, N, 64)) + 0.3 * rng.standard_normal((B, 1, 64)), 0).astype(np.float32)
elif kind == "clusters":
    cen = rng.standard_normal((B, 24, 64)) * 3
    f = (cen[:, rng.integers(0, 24, N)][np.arange(B)[:, None], np.arange(N)[None]] if False else
         np.stack([cen[b][rng.integers(0, 24, N)] for b in range(B)]) + 0.4 * rng.standard_normal((B, N, 64))).astype(np.float32)
else:
    raise SystemExit("kind")
x = torch.from_numpy(f).cuda()
os.environ["PSG_GCN_KNN_STATS"] = "1"
bf = runtime.GCNWorkspace(B, N, 28)
os.environ["PSG_GCN_KNN"] = "f32"
ex = runtime.GCNWorkspace(B, N, 28)
del os.environ["PSG_GCN_KNN"]
for d in [int(v) for v in os.environ.get("KNN_D", "1,2,4,9,14,17,22,27").split(",")]:
    if 16 * d > N:
        continue
    bf.knn_stats()
    a = bf.knn(x, d)
    st = bf.knn_stats()
    b = ex.knn(x, d)
    torch.cuda.synchronize()
    bad = int((a != b).any(dim=-1).sum())
    t = []
    for ws in (bf, ex):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            ws.knn(x, d)
        t1.record(); torch.cuda.synchronize()
        t.append(t0.elapsed_time(t1) * 100)
    rows = max(st["rows"], 1)
    print("d=%2d rows differing %d / %d | bf16 %.1f us, f32 %.1f us per call | tiles %d exact %d, finalists/row %.1f, cuts/row %.2f, entries/row %.0f"
          % (d, bad, B * N, t[0], t[1], st["tiles"], st["exact_tiles"], st["finalists"] / rows, st["cuts"] / rows,
             st["entries"] / rows), flush=True)
