"""Possibility-based crop sampler of the reference's RandLA-Net pipeline on the device.

Mirrors `S3DIS.get_batch_gen(split)` / `spatially_regular_gen` of RandLA-Net/main_S3DIS.py:116-187: every crop starts at the
least-visited point of the least-visited cloud, jitters it (noise_init / 10), takes the `num_points` nearest points of that
cloud (the reference: a 40 960-neighbour query of an sklearn KDTree on the host, whose float64 copy of the cloud also makes the
pick point and the centring float64), shuffles them, centres them on the pick point and raises their possibility by
(1 - d / max d)^2.  Here the argmin, the nearest-point query (float64 distances + one radix
sort) and the possibility update run in libpsg (csrc/psg_randla_sampler.hip); the host draws exactly what the reference draws
from numpy's global generator, in its order (initial possibilities, jitter, shuffle, duplication for small clouds), so a seeded
run visits the same crops.  Equal distances come out in index order (a KDTree's order of exact ties is unspecified).
"""
import ctypes

import numpy as np
import torch

from pointsecguard_amd import _lib, runtime


class CropSampler:
    def __init__(self, clouds, num_points=40960, noise_init=3.5, device=None):
        """clouds: list of (points [M,3] float32, colours [M,3], labels [M]) - what the reference keeps per cloud as
        input_trees[split][i].data, input_colors[split][i], input_labels[split][i]."""
        self.ctx = runtime.context(device)
        self.num_points, self.noise_init = int(num_points), float(noise_init)
        self.points = [np.ascontiguousarray(c[0], np.float32) for c in clouds]
        self.colors = [np.asarray(c[1]) for c in clouds]
        self.labels = [np.asarray(c[2]) for c in clouds]
        self.handles, self.min_possibility = [], []
        # main_S3DIS.py:125-128: possibility = rand * 1e-3 per cloud, in cloud order, from numpy's global generator
        for pts in self.points:
            poss = np.random.rand(pts.shape[0]) * 1e-3
            h = ctypes.c_void_p()
            _lib.check(_lib.load().psg_rla_sampler_create(self.ctx, pts.ctypes.data_as(ctypes.c_void_p),
                                                          poss.ctypes.data_as(ctypes.c_void_p), pts.shape[0], ctypes.byref(h)),
                       "psg_rla_sampler_create")
            self.handles.append(h)
            self.min_possibility.append(float(np.min(poss)))
        self.device = torch.device("cuda", torch.cuda.current_device())

    def __del__(self):
        try:
            for h in getattr(self, "handles", []):
                _lib.load().psg_rla_sampler_destroy(h)
            self.handles = []
        except Exception:
            pass

    def possibility(self, cloud_idx):
        out = np.empty(self.points[cloud_idx].shape[0], np.float64)
        _lib.call("psg_rla_sampler_possibility", self.handles[cloud_idx], out.ctypes.data_as(ctypes.c_void_p))
        return out

    def _argmin(self, cloud_idx):
        i, v = ctypes.c_int(), ctypes.c_double()
        _lib.call("psg_rla_sampler_argmin", self.handles[cloud_idx], ctypes.byref(i), ctypes.byref(v), runtime.stream())
        return i.value, v.value

    def next_crop(self):
        """One iteration of spatially_regular_gen (main_S3DIS.py:131-179): (xyz [n,3] float32 centred on the pick point,
        colours [n,3] float32, labels [n], point indices [n] int32, [cloud index] int32)."""
        cloud_idx = int(np.argmin(self.min_possibility))
        h, points = self.handles[cloud_idx], self.points[cloud_idx]
        point_ind, _ = self._argmin(cloud_idx)
        # main_S3DIS.py:141 takes the points from the sklearn KDTree, whose copy of the cloud is float64: the centre, the
        # jitter and the pick point are float64, and so are the differences below before they are rounded to float32
        center_point = points[point_ind, :].astype(np.float64).reshape(1, -1)
        noise = np.random.normal(scale=self.noise_init / 10, size=center_point.shape)
        pick_point = center_point + noise.astype(center_point.dtype)
        k = min(len(points), self.num_points)
        pick = np.ascontiguousarray(pick_point[0], np.float64)
        idx_dev = torch.empty(k, dtype=torch.int32, device=self.device)
        _lib.call("psg_rla_sampler_query", h, pick.ctypes.data_as(ctypes.c_void_p), k, runtime.ptr(idx_dev), runtime.stream())
        queried_idx = idx_dev.cpu().numpy().astype(np.int64)
        perm = np.arange(len(queried_idx))                    # DP.shuffle_idx (helper_tool.py:183-187)
        np.random.shuffle(perm)
        queried_idx = queried_idx[perm]
        queried_pc_xyz = points[queried_idx].astype(np.float64) - pick_point
        queried_pc_colors = self.colors[cloud_idx][queried_idx]
        queried_pc_labels = self.labels[cloud_idx][queried_idx]
        scratch = torch.empty(k, dtype=torch.float32, device=self.device)
        _lib.call("psg_rla_sampler_update", h, runtime.ptr(idx_dev), k, pick.ctypes.data_as(ctypes.c_void_p), runtime.ptr(scratch),
                  runtime.stream())
        self.min_possibility[cloud_idx] = self._argmin(cloud_idx)[1]
        if len(points) < self.num_points:                     # DP.data_aug (helper_tool.py:170-180): duplicate random points
            num_in = len(queried_pc_xyz)
            dup = np.random.choice(num_in, self.num_points - num_in)
            idx_dup = list(range(num_in)) + list(dup)
            queried_pc_xyz = np.concatenate([queried_pc_xyz, queried_pc_xyz[dup, ...]], 0)
            queried_pc_colors = np.concatenate([queried_pc_colors, queried_pc_colors[dup, ...]], 0)
            queried_idx = queried_idx[idx_dup]
            queried_pc_labels = queried_pc_labels[idx_dup]
        return (queried_pc_xyz.astype(np.float32), queried_pc_colors.astype(np.float32), queried_pc_labels,
                queried_idx.astype(np.int32), np.array([cloud_idx], dtype=np.int32))

    def spatially_regular_gen(self, num_per_epoch):
        for _ in range(num_per_epoch):
            yield self.next_crop()
