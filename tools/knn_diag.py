"""Diagnostic (GPU box): mismatch statistics of the ResGCN kNN paths against the CPU oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import resgcn  # noqa: E402
from pointsecguard_amd import runtime  # noqa: E402


def report(tag, got, ref, f):
    bad_rows = np.nonzero((got != ref).any(1))[0]
    print("%s: %d of %d rows differ" % (tag, len(bad_rows), got.shape[0]), flush=True)
    for r in bad_rows[:3]:
        cols = np.nonzero(got[r] != ref[r])[0]
        d_got = ((f[r] - f[got[r]]) ** 2).sum(1)
        d_ref = ((f[r] - f[ref[r]]) ** 2).sum(1)
        print("   row %d cols %s got %s ref %s  d_got %s d_ref %s" % (r, cols.tolist(), got[r][cols].tolist(), ref[r][cols].tolist(),
                                                                     np.round(d_got[cols], 5).tolist(), np.round(d_ref[cols], 5).tolist()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rng = np.random.default_rng(1)
    f = rng.standard_normal((n, 64)).astype(np.float32)
    xyz = rng.random((n, 3)).astype(np.float32)
    ws = runtime.GCNWorkspace(1, n, 28)
    os.environ["PSG_GCN_KNN"] = "matrix"
    legacy = runtime.GCNWorkspace(1, n, 28)
    del os.environ["PSG_GCN_KNN"]
    for d in (1, 2, 3, 4, 9, 10, 17, 18, 27):
        ref = resgcn.knn_dilated(f, d)
        for tag, w in (("fused", ws), ("matrix", legacy)):
            got = w.knn(torch.from_numpy(f[None]).cuda(), d)[0].cpu().numpy()
            report("C=64 d=%d %s" % (d, tag), got, ref, f)
    ref = resgcn.knn_dilated(xyz, 1)
    got = ws.knn(torch.from_numpy(xyz[None]).cuda(), 1)[0].cpu().numpy()
    report("xyz d=1", got, ref, xyz)


if __name__ == "__main__":
    main()
