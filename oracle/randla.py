"""CPU oracle for the RandLA-Net input pipeline's k-NN (first piece).

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/randla_knn_bench.py (CPU baseline) as the checker.

  knn_brute(support, query, k)   numpy restatement of what nearest_neighbors.knn_batch returns
                                 (RandLA-Net/utils/nearest_neighbors/knn_.cxx:103-134 -> nanoflann exact k-NN, results
                                 sorted by ascending squared L2 distance, nanoflann.hpp L2_Adaptor arithmetic
                                 ((dx*dx) + dy*dy) + dz*dz in fp32); ties broken by lowest index (nanoflann's tie order
                                 depends on its tree traversal and is not a contract)
  knn_ref(support, query, k)     the REFERENCE ITSELF: cpp_knn_batch_omp of oracle/_ref/libknn_ref.so, compiled by
                                 `make -C oracle ref` from the reference's own knn_.cxx where it lies (nothing copied)
  pyramid(xyz, ...)              the index pyramid of the reference's tf.data map function (main_S3DIS.py:198-207)
Parity status: knn_brute is pinned by tests/golden/randla_knn.npz, generated with knn_ref (tests/golden/
make_golden_randla.py), and checked against knn_ref directly whenever oracle/_ref/libknn_ref.so is present.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = None
REF_PATH = os.path.join(_HERE, "_ref", "libknn_ref.so")


def have_ref():
    return os.path.exists(REF_PATH)


def knn_ref(support, query, k):
    """[B,N1,3], [B,N2,3] float32 -> [B,N2,k] int32 through the reference's cpp_knn_batch_omp (knn_.cxx:103)."""
    global _REF
    if _REF is None:
        _REF = ctypes.CDLL(REF_PATH)
    fn = getattr(_REF, "_Z17cpp_knn_batch_ompPKfmmmS0_mmPl")   # void cpp_knn_batch_omp(const float*, size_t x3, const float*, size_t x2, long*)
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                   ctypes.c_size_t, ctypes.c_void_p]
    s = np.ascontiguousarray(support, np.float32)
    q = np.ascontiguousarray(query, np.float32)
    out = np.zeros((s.shape[0], q.shape[1], k), np.int64)
    fn(s.ctypes.data, s.shape[0], s.shape[1], 3, q.ctypes.data, q.shape[1], k, out.ctypes.data)
    return out.astype(np.int32)


def sqdist(support, query):
    """[N1,3], [N2,3] -> [N2,N1] fp32 squared distances with nanoflann's evaluation order."""
    d = query[:, None, :].astype(np.float32) - support[None, :, :].astype(np.float32)
    sq = d * d
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def knn_brute(support, query, k, chunk=1024):
    support = np.ascontiguousarray(support, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    out = np.empty((support.shape[0], query.shape[1], k), np.int32)
    for b in range(support.shape[0]):
        for q0 in range(0, query.shape[1], chunk):
            d = sqdist(support[b], query[b, q0:q0 + chunk])
            out[b, q0:q0 + chunk] = np.argsort(d, axis=1, kind="stable")[:, :k]    # stable: lowest index on ties
    return out


def pyramid(xyz, num_layers=5, k_n=16, ratios=(4, 4, 4, 4, 2), knn=knn_brute):
    """tf_map of main_S3DIS.py:198-207: per layer (points, neighbour idx, pool idx, up-sample idx)."""
    pts, neigh, pools, ups = [], [], [], []
    cur = np.ascontiguousarray(xyz, np.float32)
    for i in range(num_layers):
        nb = knn(cur, cur, k_n)
        n_sub = cur.shape[1] // ratios[i]
        sub = np.ascontiguousarray(cur[:, :n_sub])
        pts.append(cur); neigh.append(nb); pools.append(nb[:, :n_sub]); ups.append(knn(sub, cur, 1))
        cur = sub
    return pts, neigh, pools, ups


class CropSamplerOracle:
    """numpy restatement of RandLA-Net/main_S3DIS.py:116-187 (get_batch_gen / spatially_regular_gen), source-read: the
    reference module imports TensorFlow and open3d and cannot be imported here.  The KDTree query is restated as what it
    returns: the k nearest points by float64 squared distance, ascending, exact ties in index order."""

    def __init__(self, clouds, num_points=40960, noise_init=3.5):
        self.num_points, self.noise_init = num_points, noise_init
        self.points = [np.ascontiguousarray(c[0], np.float32) for c in clouds]
        self.colors = [np.asarray(c[1]) for c in clouds]
        self.labels = [np.asarray(c[2]) for c in clouds]
        self.possibility = [np.random.rand(p.shape[0]) * 1e-3 for p in self.points]
        self.min_possibility = [float(np.min(p)) for p in self.possibility]

    def next_crop(self):
        cloud_idx = int(np.argmin(self.min_possibility))
        point_ind = np.argmin(self.possibility[cloud_idx])
        # main_S3DIS.py:141: `points` is the sklearn KDTree's copy of the cloud, which is float64 (tests/test_randla_sampler.py
        # checks that against sklearn itself): centre, jitter, pick point and the differences below are float64
        points = self.points[cloud_idx].astype(np.float64)
        center_point = points[point_ind, :].reshape(1, -1)
        noise = np.random.normal(scale=self.noise_init / 10, size=center_point.shape)
        pick_point = center_point + noise.astype(center_point.dtype)
        k = min(len(points), self.num_points)
        d = points - pick_point
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        queried_idx = np.argsort(d2, kind="stable")[:k]
        idx = np.arange(len(queried_idx))
        np.random.shuffle(idx)
        queried_idx = queried_idx[idx]
        queried_pc_xyz = points[queried_idx] - pick_point
        queried_pc_colors = self.colors[cloud_idx][queried_idx]
        queried_pc_labels = self.labels[cloud_idx][queried_idx]
        dists = np.sum(np.square((points[queried_idx] - pick_point).astype(np.float32)), axis=1)
        delta = np.square(1 - dists / np.max(dists))
        self.possibility[cloud_idx][queried_idx] += delta
        self.min_possibility[cloud_idx] = float(np.min(self.possibility[cloud_idx]))
        if len(points) < self.num_points:
            num_in = len(queried_pc_xyz)
            dup = np.random.choice(num_in, self.num_points - num_in)
            idx_dup = list(range(num_in)) + list(dup)
            queried_pc_xyz = np.concatenate([queried_pc_xyz, queried_pc_xyz[dup, ...]], 0)
            queried_pc_colors = np.concatenate([queried_pc_colors, queried_pc_colors[dup, ...]], 0)
            queried_idx = queried_idx[idx_dup]
            queried_pc_labels = queried_pc_labels[idx_dup]
        return (queried_pc_xyz.astype(np.float32), queried_pc_colors.astype(np.float32), queried_pc_labels,
                queried_idx.astype(np.int32), np.array([cloud_idx], dtype=np.int32))
