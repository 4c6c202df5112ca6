"""`DataProcessing.knn_search` and the per-batch index pyramid of the reference's RandLA-Net pipeline, on the device.

Mirrors RandLA-Net/helper_tool.py:158-167 (`DataProcessing.knn_search(support_pts, query_pts, k)`: numpy in, int32
numpy out, like the reference, whose implementation is a nanoflann kd-tree per batch element on the host cores) and
the tf.data map function of main_S3DIS.py:189-214 (`tf_map`: for each of the num_layers encoder levels the points, their
k_n neighbours, the pooling indices of the sub-sampled points and the nearest sub-sampled point of every point).
Both run `psg_knn_points` (exact brute-force k-NN, hand-written HIP); there is no CPU path.

Only the indices are produced here; the network (RandLANet.py) is `randla/network.py`, the tester's attacks are
`randla/attack.py`, and the possibility-based crop sampler of main_S3DIS.py:116-187 is `randla/sampler.py`.
"""
import numpy as np
import torch

from pointsecguard_amd import _lib, runtime


def knn_points(support, query, k):
    """support [B,N1,3], query [B,N2,3] float32 CUDA tensors -> [B,N2,k] int32 neighbour indices into `support`,
    ascending (squared distance, index)."""
    runtime.require_cuda(support, "support", torch.float32)
    runtime.require_cuda(query, "query", torch.float32)
    if support.dim() != 3 or query.dim() != 3 or support.shape[2] != 3 or query.shape[2] != 3 or support.shape[0] != query.shape[0]:
        raise ValueError("expected support [B,N1,3] and query [B,N2,3], got %s and %s" % (tuple(support.shape), tuple(query.shape)))
    B, N1, _ = support.shape
    N2 = query.shape[1]
    out = torch.empty(B, N2, int(k), dtype=torch.int32, device=support.device)
    _lib.call("psg_knn_points", runtime.context(support.device), runtime.ptr(support), runtime.ptr(query), B, N1, N2, int(k),
              runtime.ptr(out), runtime.stream())
    return out


class DataProcessing:
    @staticmethod
    def knn_search(support_pts, query_pts, k):
        """
        :param support_pts: points you have, B*N1*3
        :param query_pts: points you want to know the neighbour index, B*N2*3
        :param k: Number of neighbours in knn search
        :return: neighbor_idx: neighboring points indexes, B*N2*k
        """
        s = torch.from_numpy(np.ascontiguousarray(support_pts, np.float32)).cuda()
        q = torch.from_numpy(np.ascontiguousarray(query_pts, np.float32)).cuda()
        return knn_points(s, q, k).cpu().numpy().astype(np.int32)


def tf_map_indices(batch_xyz, num_layers=5, k_n=16, sub_sampling_ratio=(4, 4, 4, 4, 2)):
    """The index part of tf_map (main_S3DIS.py:198-207) for a batch of clouds [B,N,3] resident on the device:
    returns (input_points, input_neighbors, input_pools, input_up_samples), each a list of num_layers tensors.
    Sub-sampling is the reference's: the first N // ratio points (its clouds arrive shuffled)."""
    runtime.require_cuda(batch_xyz, "batch_xyz", torch.float32)
    input_points, input_neighbors, input_pools, input_up_samples = [], [], [], []
    cur = batch_xyz
    for i in range(num_layers):
        neighbour_idx = knn_points(cur, cur, k_n)
        n_sub = cur.shape[1] // sub_sampling_ratio[i]
        sub_points = cur[:, :n_sub, :].contiguous()
        input_points.append(cur)
        input_neighbors.append(neighbour_idx)
        input_pools.append(neighbour_idx[:, :n_sub, :])
        input_up_samples.append(knn_points(sub_points, cur, 1))
        cur = sub_points
    return input_points, input_neighbors, input_pools, input_up_samples


DP = DataProcessing
