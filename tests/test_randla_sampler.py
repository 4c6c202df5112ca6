"""The possibility-based crop sampler of the RandLA-Net pipeline (pointsecguard_amd/randla/sampler.py over
csrc/psg_randla_sampler.hip; reference: RandLA-Net/main_S3DIS.py:116-187) against the numpy restatement of the generator
(oracle/randla.py: CropSamplerOracle, source-read - the reference module imports TensorFlow): with the same numpy seed the two
visit the same crops, point for point, and end with the same possibilities."""
import numpy as np
import pytest

from oracle import randla


def clouds(seed, sizes):
    rng = np.random.default_rng(seed)
    out = []
    for m in sizes:
        pts = (rng.random((m, 3)) * np.array([12.0, 9.0, 3.0])).astype(np.float32)
        out.append((pts, rng.integers(0, 255, (m, 3)).astype(np.uint8), rng.integers(0, 13, m).astype(np.uint8)))
    return out


def test_oracle_generator_properties():
    """The restatement alone (no GPU): crops have num_points points, are centred on the jittered pick point, possibilities
    only grow, and the least-visited cloud is served next."""
    cl = clouds(1, (3000, 5000))
    np.random.seed(3)
    orc = randla.CropSamplerOracle(cl, num_points=1024)
    before = [p.copy() for p in orc.possibility]
    seen = []
    for _ in range(6):
        xyz, col, lab, idx, ci = orc.next_crop()
        assert xyz.shape == (1024, 3) and col.shape == (1024, 3) and lab.shape == (1024,) and idx.dtype == np.int32
        assert len(np.unique(idx)) == 1024 and np.abs(xyz).max() < 12
        seen.append(int(ci[0]))
    assert set(seen) == {0, 1}
    assert all((a >= b).all() for a, b in zip(orc.possibility, before))
    # a cloud smaller than the crop: every point once, then random duplicates (DP.data_aug)
    np.random.seed(4)
    small = randla.CropSamplerOracle(clouds(2, (700,)), num_points=1024)
    xyz, col, lab, idx, ci = small.next_crop()
    assert xyz.shape == (1024, 3) and len(np.unique(idx)) == 700 and np.array_equal(np.sort(idx[:700]), np.arange(700))


def test_oracle_query_is_the_sklearn_tree_query():
    """The reference's neighbour source is `sklearn.neighbors.KDTree(sub_xyz)` (utils/data_prepare_s3dis.py:62) queried with
    the float64 pick point (main_S3DIS.py:141-153).  sklearn is installed here, so the restatement's two assumptions are
    checked against it directly: the tree's copy of a float32 cloud is float64 (which makes `points`, the pick point and
    the centring float64), and `query(pick, k)` returns the k smallest float64 squared distances in ascending order."""
    KDTree = pytest.importorskip("sklearn.neighbors").KDTree
    cl = clouds(5, (6000,))
    tree = KDTree(cl[0][0])
    assert np.asarray(tree.data).dtype == np.float64 and np.array_equal(np.asarray(tree.data), cl[0][0].astype(np.float64))
    np.random.seed(21)
    orc = randla.CropSamplerOracle(cl, num_points=1500)
    state = np.random.get_state()
    point_ind = np.argmin(orc.possibility[0])
    points = np.asarray(tree.data)
    center = points[point_ind, :].reshape(1, -1)
    pick = center + np.random.normal(scale=orc.noise_init / 10, size=center.shape).astype(center.dtype)
    want = tree.query(pick, k=1500)[1][0]
    perm = np.arange(1500)
    np.random.shuffle(perm)
    np.random.set_state(state)
    xyz, _, _, idx, _ = orc.next_crop()
    assert np.array_equal(idx, want[perm].astype(np.int32))                       # same points, same order (no exact ties here)
    assert np.array_equal(xyz, (points[want[perm]] - pick).astype(np.float32))    # float64 difference, rounded once


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,num_points,n_crops", [((30000, 41000, 25000), 8192, 9), ((60000,), 40960, 3), ((900, 5000), 2048, 5)])
def test_gpu_sampler_equals_restatement(sizes, num_points, n_crops):
    from pointsecguard_amd.randla import sampler
    cl = clouds(7 + len(sizes), sizes)
    np.random.seed(11)
    orc = randla.CropSamplerOracle(cl, num_points=num_points)
    want = [orc.next_crop() for _ in range(n_crops)]
    np.random.seed(11)
    dev = sampler.CropSampler(cl, num_points=num_points)
    got = list(dev.spatially_regular_gen(n_crops))
    for c, (g, w) in enumerate(zip(got, want)):
        assert int(g[4][0]) == int(w[4][0]), c
        assert np.array_equal(g[3], w[3]), (c, int((g[3] != w[3]).sum()))
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]) and np.array_equal(g[2], w[2])
    for i in range(len(sizes)):
        assert np.array_equal(dev.possibility(i), orc.possibility[i]), i           # float64, bit for bit
    assert dev.min_possibility == orc.min_possibility
