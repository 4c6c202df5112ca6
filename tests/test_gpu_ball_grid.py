"""The grid ball query (ball_query_grid_kernel, psg_geometry.hip; reference: query_ball_point, PointNet/models/
pointnet_util.py:87-107) returns what the full scan returns - the first `nsample` indices, ascending, whose expansion distance
is not above float32(r^2), padded with the first - bit for bit against the oracle, on the cases where pruning by cells could
go wrong: radii from far below to far above the cell structure, clustered and planar clouds (thousands of points per cell,
empty cells), exact duplicates, centroids that are not cloud points (outside the bounding box, empty balls), coordinates with
a large offset (the expansion's rounding error grows with |x|^2: the cell edge follows it), non-finite coordinates (one
cell = the scan), and against the scan kernel itself in a child process (PSG_BALL_QUERY=scan)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def clouds(kind, n, rng):
    if kind == "uniform":
        x = rng.random((n, 3), dtype=np.float32) * np.array([1, 1, 3], np.float32) - np.array([0.5, 0.5, 0], np.float32)
    elif kind == "planes":                                  # walls and a floor: dense sheets, empty volume
        x = rng.random((n, 3), dtype=np.float32) * np.array([1, 1, 3], np.float32)
        x[: n // 3, 0] = 0.0
        x[n // 3: 2 * n // 3, 2] = 0.0
        x[2 * n // 3:, 1] = 1.0
    elif kind == "cluster":                                 # 90 % of the points in a 5 cm blob
        x = rng.random((n, 3), dtype=np.float32)
        x[: 9 * n // 10] = 0.5 + 0.05 * rng.random((9 * n // 10, 3), dtype=np.float32)
    elif kind == "offset":                                  # a room far from the origin: |x|^2 ~ 1e4
        x = rng.random((n, 3), dtype=np.float32) + np.array([70, -60, 40], np.float32)
    elif kind == "line":                                    # degenerate box: one cell row
        x = np.zeros((n, 3), np.float32)
        x[:, 0] = rng.random(n, dtype=np.float32) * 5
    else:
        raise ValueError(kind)
    x[n // 2] = x[n // 3]                                   # exact duplicates
    x[n // 2 + 1] = x[n // 3]
    return x.astype(np.float32)


@pytest.mark.parametrize("kind,n,s,radius,k", [
    ("uniform", 4096, 1024, 0.1, 32), ("uniform", 4096, 1024, 0.2, 16), ("uniform", 2048, 256, 0.2, 32),
    ("uniform", 4096, 300, 0.02, 32), ("uniform", 4096, 300, 2.5, 32), ("planes", 4096, 1024, 0.1, 32),
    ("cluster", 4096, 512, 0.1, 32), ("cluster", 3000, 333, 0.03, 8), ("offset", 4096, 512, 0.1, 32),
    ("line", 2560, 100, 0.1, 32)])
def test_grid_ball_query_equals_oracle(kind, n, s, radius, k):
    from oracle import pn2
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(hash((kind, n, s)) % 1000)
    P = 2
    xyz = np.stack([clouds(kind, n, rng) for _ in range(P)])
    new = np.stack([x[rng.choice(n, s, replace=False)] for x in xyz])
    new[:, 0] = xyz[:, 0] + np.float32(3.0 * radius)          # centroids that are no cloud points: near, ..
    new[:, 1] = xyz.min(axis=1) - np.float32(0.4 * radius)    # .. just outside the box, ..
    new[:, 2] = xyz.max(axis=1) + np.float32(50.0)            # .. far outside (empty ball: every slot = N)
    got = runtime.ball_query(radius, k, dev(xyz), dev(new)).cpu().numpy()
    for p in range(P):
        want = pn2.ball_query(radius, k, xyz[p], new[p])
        assert np.array_equal(got[p], want), (kind, p, int((got[p] != want).any(1).sum()))
    assert (got[:, 2] == n).all()


def test_grid_ball_query_non_finite_cloud_is_the_scan():
    from oracle import pn2
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(3)
    xyz = clouds("uniform", 2048, rng)[None].copy()       # (2048: the smallest cloud the grid kernel takes)
    xyz[0, 17, 1] = np.inf
    xyz[0, 900, 0] = np.nan
    new = xyz[:, rng.choice(2048, 64, replace=False)].copy()
    got = runtime.ball_query(0.15, 32, dev(xyz), dev(new)).cpu().numpy()
    assert np.array_equal(got[0], pn2.ball_query(0.15, 32, xyz[0], new[0]))


def test_grid_ball_query_arbitrary_centroids_equal_the_scan():
    """psg_ball_query takes ARBITRARY new_xyz (advisor, round 4): a NaN / infinite centroid passes `!(d > r^2)` for every point
    in the reference's test (indices 0 .. K-1), and so does a finite centroid so far away that the expansion -2ab + |a|^2 +
    |b|^2 overflows to inf - inf (3e38: the dot product and |c|^2 are both infinite) - neither is something a 27-cell walk
    would find.  (Merely distant centroids pass nothing: the expansion's error is relative to |c|^2, and so is the distance.)
    Centroids that are non-finite or more than a cell edge outside the box take the whole cloud as candidates inside the
    grid kernel: same rows as the oracle's scan."""
    from oracle import pn2
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(11)
    n, radius, k = 4096, 0.1, 32
    xyz = clouds("uniform", n, rng)[None].copy()
    new = xyz[:, rng.choice(n, 96, replace=False)].copy()
    new[0, 0] = np.nan
    new[0, 1] = (np.inf, 0.0, 1.0)
    new[0, 2] = (0.1, -np.inf, 1.0)
    new[0, 3] = (0.2, 0.1, np.nan)
    for i, far in enumerate((1e3, 1e5, 3e6, 1e20, 3e38)):                 # far away, up to overflow of the expansion
        new[0, 4 + 2 * i] = xyz[0, 50 + i] + np.float32(far)
        new[0, 5 + 2 * i] = xyz[0, 80 + i] * np.array([1, 1, -1], np.float32) - np.array([0, far, 0], np.float32)
    new[0, 14] = xyz[0].min(axis=0) - np.float32(1.5 * radius)           # just beyond one cell edge outside the box
    new[0, 15] = xyz[0].max(axis=0) + np.float32(0.9 * radius)           # inside one cell edge: the border-cell walk
    got = runtime.ball_query(radius, k, dev(xyz), dev(new)).cpu().numpy()
    with np.errstate(all="ignore"):
        want = pn2.ball_query(radius, k, xyz[0], new[0])
    assert np.array_equal(got[0], want), np.nonzero((got[0] != want).any(1))[0]
    assert np.array_equal(got[0, 0], np.arange(k))                       # the NaN centroid: the first K points
    assert (want[12] != n).all() and want[12, 0] == 0 and (want[4] == n).all()   # overflow: a full row of far points; merely far: nothing


def test_grid_equals_scan_kernel_on_the_network_plan():
    """The whole geometry plan of an attack (4 levels x the reference's FPS starts) under PSG_BALL_QUERY=scan in a child
    process against the default (grid for levels 0 and 1): identical group tables."""
    code = """
import numpy as np, torch, sys
sys.path.insert(0, %r)
from pointsecguard_amd import runtime
from pointsecguard_amd.synthetic import make_rooms
rooms = make_rooms(3, 77, structured=True)
x0 = torch.from_numpy(rooms).cuda()
rng = np.random.default_rng(5)
starts = torch.from_numpy(np.stack([rng.integers(0, n, (2, 3)) for n in (4096, 1024, 256, 64)], axis=1).astype(np.int32)).cuda()
ws = runtime.PN2Workspace(3, 4096, 2)
ws.plan_build(x0, starts, 2)
torch.cuda.synchronize()
out = [ws.plan_tensor(1, l, f, b).cpu().numpy() for l in range(4) for f in range(2) for b in range(3)]
np.save(sys.argv[1], np.concatenate([o.reshape(-1) for o in out]))
""" % ROOT
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        res = []
        for mode in ("grid", "scan"):
            env = dict(os.environ)
            env.pop("PSG_BALL_QUERY", None)
            if mode == "scan":
                env["PSG_BALL_QUERY"] = "scan"
            path = os.path.join(tmp, mode + ".npy")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res.append(np.load(path))
        assert res[0].size > 100000 and np.array_equal(res[0], res[1])
