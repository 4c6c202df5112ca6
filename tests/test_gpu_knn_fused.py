"""The matrix-free feature-space kNN (psg_knn_fused.cuh; reference: ResGCN/gcn_lib/dense/torch_edge.py:32-59, 19-29)
against the CPU oracle (bit-exact: same fp32 distance order, ties -> lowest index) at every buffer class, for
adversarial candidate orders (every candidate enters the running top list: exercises pruning and the overflow
roll-back), massive ties, several rooms per launch and ragged sizes, and against the round-1 path (distance matrix in
HBM + selection kernel, PSG_GCN_KNN=matrix) on the same inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda().contiguous()


def oracle_rooms(f, d):
    from oracle import resgcn
    return np.stack([resgcn.knn_dilated(f[b], d) for b in range(f.shape[0])])


@pytest.mark.parametrize("d", [1, 2, 3, 4, 9, 10, 17, 18, 27])
def test_every_buffer_class_4096(d):
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(100 + d)
    f = (rng.standard_normal((1, 4096, 64)) * rng.uniform(0.2, 3.0, (1, 1, 64))).astype(np.float32)
    ws = runtime.GCNWorkspace(1, 4096, 28)
    got = ws.knn(dev(f), d).cpu().numpy()
    assert np.array_equal(got, oracle_rooms(f, d))


def test_adversarial_orders_and_ties():
    """Points ordered so that EVERY later candidate is nearer to the late queries than all earlier ones (each one enters
    the running top list: N appends per row instead of ~KK ln N), the reverse order, and 3000 exact duplicates."""
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(7)
    n = 2048
    u = rng.standard_normal(64).astype(np.float32)
    shrink = (1.0 - np.arange(n, dtype=np.float32) / n)[:, None] * u[None] * 4 + rng.standard_normal((n, 64)).astype(np.float32) * 1e-3
    ws = runtime.GCNWorkspace(1, n, 28)
    for name, f in (("shrinking", shrink), ("growing", shrink[::-1].copy())):
        for d in (1, 5, 27):
            got = ws.knn(dev(f[None]), d).cpu().numpy()
            assert np.array_equal(got, oracle_rooms(f[None], d)), (name, d)
    f = rng.standard_normal((n, 64)).astype(np.float32)
    dup = rng.permutation(n)[:1500]
    f[dup] = f[dup[0]]
    for d in (1, 12, 27):
        got = ws.knn(dev(f[None]), d).cpu().numpy()
        assert np.array_equal(got, oracle_rooms(f[None], d)), ("ties", d)


@pytest.mark.parametrize("n,batch", [(448, 3), (1040, 2), (4096, 2)])
def test_rooms_and_ragged_sizes(n, batch):
    """Several rooms per launch (neighbours never cross rooms) and N not a multiple of the 32-candidate step."""
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(n)
    f = rng.standard_normal((batch, n, 64)).astype(np.float32)
    ws = runtime.GCNWorkspace(batch, n, 28)
    for d in (1, 7, 27):
        if 16 * d > n:
            continue
        got = ws.knn(dev(f), d).cpu().numpy()
        assert got.max() < n
        assert np.array_equal(got, oracle_rooms(f, d)), d


def test_matches_round1_matrix_path():
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(5)
    f = rng.standard_normal((1, 4096, 64)).astype(np.float32)
    fused = runtime.GCNWorkspace(1, 4096, 28)
    os.environ["PSG_GCN_KNN"] = "matrix"
    try:
        legacy = runtime.GCNWorkspace(1, 4096, 28)
    finally:
        del os.environ["PSG_GCN_KNN"]
    for d in (1, 13, 27):
        a, b = fused.knn(dev(f), d).cpu().numpy(), legacy.knn(dev(f), d).cpu().numpy()
        assert np.array_equal(a, b), d
