"""RandLA-Net pipeline k-NN (first piece).

CPU: the numpy oracle (oracle/randla.py: knn_brute) against tests/golden/randla_knn.npz, which holds the output of
the reference's own nanoflann code (compiled from its sources into oracle/_ref by `make -C oracle ref`), and against
that library directly when it is present.  GPU: psg_knn_points through the C ABI against the same fixtures, against the
oracle, and at the reference's full size (40 960 points) through properties.

Equality bar: the sorted squared distances of the returned neighbours are bit-equal and the index sets are equal;
where indices differ in ORDER the distances there are exactly tied (nanoflann's tie order follows its tree traversal)."""
import os

import numpy as np
import pytest

from oracle import randla

GOLD = os.path.join(os.path.dirname(__file__), "golden", "randla_knn.npz")


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(GOLD))


def assert_same_knn(got, ref, support, query):
    """got / ref [B,N2,k] indices into support [B,N1,3] for query [B,N2,3]."""
    assert got.shape == ref.shape
    for b in range(got.shape[0]):
        qa = query[b][:, None, :].astype(np.float32)
        def dist(idx):
            d = qa - support[b][idx]
            sq = d * d
            return (sq[..., 0] + sq[..., 1]) + sq[..., 2]
        dg, dr = dist(got[b]), dist(ref[b])
        assert np.array_equal(dg.view(np.uint32), dr.view(np.uint32)), "neighbour distances differ"
        assert np.all(np.diff(dg, axis=1) >= 0), "not sorted by distance"
        diff = got[b] != ref[b]
        if diff.any():     # only inside runs of exactly equal distance, and as sets nothing may change ...
            rows = np.where(diff.any(axis=1))[0]
            for r in rows:
                cols = np.where(diff[r])[0]
                tied = [np.sum(dg[r] == dg[r, c]) > 1 or c == got.shape[2] - 1 for c in cols]
                assert all(tied), (b, r)
                # ... except for a tie that straddles the k-th place
                assert len(set(got[b, r]) ^ set(ref[b, r])) <= 2 * int(np.sum(dg[r] == dg[r, -1]))


CASES = (("a", "a_support", "a_support", 16), ("b", "b_support", "b_query", 5), ("d", "d_support", "d_support", 16))


@pytest.mark.parametrize("tag,sup,qry,k", CASES)
def test_oracle_vs_reference_fixture(fx, tag, sup, qry, k):
    got = randla.knn_brute(fx[sup], fx[qry], k)
    assert_same_knn(got, fx[tag + "_idx"], fx[sup], fx[qry])


def test_oracle_upsample_index_vs_reference_fixture(fx):
    room = fx["a_support"]
    sub = np.ascontiguousarray(room[:, :1024])
    assert_same_knn(randla.knn_brute(sub, room, 1), fx["c_idx"], sub, room)


@pytest.mark.skipif(not randla.have_ref(), reason="oracle/_ref/libknn_ref.so not built (make -C oracle ref)")
def test_oracle_vs_reference_library():
    rng = np.random.default_rng(17)
    s = rng.random((2, 2500, 3), dtype=np.float32)
    q = rng.random((2, 600, 3), dtype=np.float32)
    for k in (1, 3, 16):
        assert_same_knn(randla.knn_brute(s, q, k), randla.knn_ref(s, q, k), s, q)


def test_pyramid_shapes():
    xyz = np.random.default_rng(2).random((1, 2048, 3), dtype=np.float32)
    pts, neigh, pools, ups = randla.pyramid(xyz, num_layers=3, ratios=(4, 4, 2))
    assert [p.shape[1] for p in pts] == [2048, 512, 128]
    assert neigh[1].shape == (1, 512, 16) and pools[1].shape == (1, 128, 16) and ups[1].shape == (1, 512, 1)
    assert np.array_equal(neigh[0][0, :, 0], np.arange(2048))       # every point is its own nearest neighbour


# ------------------------------------------------------------------------------------------------ GPU
def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,sup,qry,k", CASES)
def test_gpu_vs_reference_fixture(fx, tag, sup, qry, k):
    from pointsecguard_amd.randla.helper_tool import knn_points
    got = knn_points(dev(fx[sup]), dev(fx[qry]), k).cpu().numpy()
    assert_same_knn(got, fx[tag + "_idx"], fx[sup], fx[qry])
    assert np.array_equal(got, randla.knn_brute(fx[sup], fx[qry], k))     # canonical (distance, index) order: exact


@pytest.mark.gpu
def test_gpu_reference_api_and_pyramid(fx):
    """DataProcessing.knn_search (numpy in / out like the reference) and the tf_map index pyramid."""
    import torch
    from pointsecguard_amd.randla.helper_tool import DP, tf_map_indices
    room = fx["a_support"]
    idx = DP.knn_search(room, room, 16)
    assert idx.dtype == np.int32 and idx.shape == (2, 4096, 16)
    assert_same_knn(idx, fx["a_idx"], room, room)
    pts, neigh, pools, ups = tf_map_indices(dev(room), num_layers=3, sub_sampling_ratio=(4, 4, 2))
    o_pts, o_neigh, o_pools, o_ups = randla.pyramid(room, num_layers=3, ratios=(4, 4, 2))
    for i in range(3):
        assert np.array_equal(pts[i].cpu().numpy(), o_pts[i])
        assert np.array_equal(neigh[i].cpu().numpy(), o_neigh[i])
        assert np.array_equal(pools[i].cpu().numpy(), o_pools[i])
        assert np.array_equal(ups[i].cpu().numpy(), o_ups[i])
    assert_same_knn(ups[0].cpu().numpy(), fx["c_idx"], room[:, :1024], room)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_gpu_full_size_properties():
    """BASELINE's RandLA size: 40 960 points, k = 16, batch 2.  Self first, sorted, and a random sample of queries
    equal to the oracle; k = 1 up-sampling of the 4x sub-sampled cloud likewise."""
    from pointsecguard_amd.randla.helper_tool import knn_points
    rng = np.random.default_rng(40960)
    xyz = (rng.random((2, 40960, 3), dtype=np.float32) * np.array([8, 6, 3], np.float32)).astype(np.float32)
    got = knn_points(dev(xyz), dev(xyz), 16).cpu().numpy()
    assert np.array_equal(got[:, :, 0], np.broadcast_to(np.arange(40960), (2, 40960)))
    pick = rng.choice(40960, 512, replace=False)
    for b in range(2):
        want = randla.knn_brute(xyz[b:b + 1], xyz[b:b + 1, pick], 16)[0]
        assert np.array_equal(got[b, pick], want)
    sub = np.ascontiguousarray(xyz[:, :10240])
    up = knn_points(dev(sub), dev(xyz), 1).cpu().numpy()
    assert np.array_equal(up[:, :10240, 0], np.broadcast_to(np.arange(10240), (2, 10240)))
    want = randla.knn_brute(sub[:1], xyz[:1, pick], 1)[0]
    assert np.array_equal(up[0, pick], want)


@pytest.mark.gpu
def test_gpu_argument_checks():
    from pointsecguard_amd import _lib
    from pointsecguard_amd.randla.helper_tool import knn_points
    x = dev(np.zeros((1, 8, 3), np.float32))
    with pytest.raises(_lib.PsgError):
        knn_points(x, x, 17)
    with pytest.raises(_lib.PsgError):
        knn_points(x, x, 9)        # k > n_support
    import torch
    with pytest.raises(_lib.PsgError):
        knn_points(torch.zeros(1, 8, 3), torch.zeros(1, 8, 3), 2)   # CPU tensors: no CPU path
