"""Time the RandLA-Net index pyramid (main_S3DIS.py:198-207: 5 levels of knn(xyz, xyz, 16) + knn(sub, xyz, 1) on a
40 960-point cloud, val_batch_size = 1) on the MI355X against the reference's own nanoflann + OpenMP code
(oracle/_ref/libknn_ref.so, compiled from the reference's knn_.cxx) on this box's host cores.

    python tools/randla_knn_bench.py [--batch 1] [--reps 20] > profiles/<round>_randla_knn.json

Inputs are resident in HBM when the timed region starts; timing is by HIP events on the launch stream.
Algorithmic work per cloud: sum over levels of N_l^2 (self k-NN) + N_l * N_l/ratio (up-sampling) candidate pairs, each
3 subtractions + 3 multiplies + 2 adds + 1 compare: 9 lane-operations.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import randla  # noqa: E402  (CPU baseline leg only)
from pointsecguard_amd.randla.helper_tool import tf_map_indices  # noqa: E402

RATIOS = (4, 4, 4, 4, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--points", type=int, default=40960)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    xyz = (rng.random((args.batch, args.points, 3), dtype=np.float32) * np.array([8, 6, 3], np.float32)).astype(np.float32)
    d = torch.from_numpy(xyz).cuda()
    for _ in range(3):
        tf_map_indices(d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        out = tf_map_indices(d)
    e1.record()
    torch.cuda.synchronize()
    gpu_ms = e0.elapsed_time(e1) / args.reps
    pairs, n = 0, args.points
    for r in RATIOS:
        pairs += n * n + n * (n // r)
        n //= r
    res = {"metric": "RandLA-Net index pyramid (5 levels, k=16 + up-sampling k=1)", "points": args.points,
           "batch": args.batch, "gpu_ms_per_batch": gpu_ms, "clouds_per_s": args.batch / gpu_ms * 1e3,
           "candidate_pairs_per_cloud": pairs,
           "lane_ops_per_s": 9.0 * pairs * args.batch / gpu_ms * 1e3,
           "valu_peak_lane_ops_per_s": 256 * 4 * 16 * 2.4e9,
           "note": "brute force: 9 VALU lane-operations per candidate pair + 1 broadcast LDS read per pair per wave"}
    res["valu_frac"] = res["lane_ops_per_s"] / res["valu_peak_lane_ops_per_s"]
    if randla.have_ref():
        t0 = time.perf_counter()
        ref = randla.pyramid(xyz, knn=randla.knn_ref)
        cpu_s = time.perf_counter() - t0
        same = all(np.array_equal(np.sort(a.cpu().numpy(), axis=2), np.sort(b, axis=2)) for a, b in zip(out[1], ref[1]))
        res["cpu_baseline"] = {"value": args.batch / cpu_s, "unit": "clouds/s", "cores": os.cpu_count(), "kind": "reference",
                               "sample": "the same batch through the reference's cpp_knn_batch_omp (nanoflann kd-tree, "
                                         "OpenMP over the batch only: one core per cloud), %.2f s" % cpu_s,
                               "neighbour_sets_equal": bool(same)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
