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
