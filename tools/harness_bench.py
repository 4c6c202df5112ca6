"""Per-batch cost of the whole-scene harness bookkeeping (8 blocks x 4096 points): device kernels (vote pool x2,
per-batch counters x2, L2 distance) against the reference-style host code (the add_vote Python double loop,
NB_nontarget_test_semseg.py:55-62, restated by oracle/harness.py's loop-free np.add.at and timed both ways).
Prints one JSON line; run on the GPU box:  python tools/harness_bench.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointsecguard_amd import harness, runtime
from oracle import harness as oh

B, N, P = 8, 4096, 200000
rng = np.random.default_rng(0)
idx = rng.integers(0, P, (B, N)).astype(np.int32)
logp = np.log(rng.dirichlet(np.ones(13), (B, N))).astype(np.float32)
w = (rng.random((B, N)) > 0.05).astype(np.float32)
gt = rng.integers(0, 13, (B, N)).astype(np.int32)
d_idx, d_logp, d_w, d_gt = (torch.from_numpy(a).cuda() for a in (idx, logp, w, gt))
pool = torch.zeros(P, 13, dtype=torch.int32, device="cuda")
adv = d_logp + 0.01


def device_batch():
    harness.add_vote(pool, d_idx, d_logp, d_w)
    harness.add_vote(pool, d_idx, adv, d_w)
    runtime.seg_stats(d_logp, d_gt)
    runtime.seg_stats(adv, d_gt)
    harness.l2_distance(d_logp, adv)


for _ in range(5):
    device_batch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    device_batch()
torch.cuda.synchronize()
dev_us = (time.perf_counter() - t0) / 50 * 1e6

pred = logp.argmax(2)
t0 = time.perf_counter()
for _ in range(5):
    oh.add_vote(np.zeros((P, 13)), idx, pred, w)
np_ms = (time.perf_counter() - t0) / 5 * 1e3

hp = np.zeros((P, 13))
t0 = time.perf_counter()
for b in range(1):                       # one of the 8 blocks, extrapolated x8 (and x2 for the adversarial pool)
    for n in range(N):
        if w[b, n]:
            hp[int(idx[b, n]), int(pred[b, n])] += 1
loop_ms = (time.perf_counter() - t0) * 1e3 * B
print(json.dumps({"batch": "8 x 4096 points, 13 classes", "device_us_per_batch_all_bookkeeping": round(dev_us, 1),
                  "numpy_add_at_ms_per_pool": round(np_ms, 2), "python_double_loop_ms_per_pool": round(loop_ms, 1),
                  "speedup_vs_reference_loop_two_pools": round(2 * loop_ms * 1e3 / dev_us, 1)}))
