"""PointNet++ point-set operators behind the reference's `models/pointnet_util.py` names.

Public names and argument meaning follow the reference (PointNet/models/pointnet_util.py):
square_distance :19, index_points :43, farthest_point_sample :63, query_ball_point :87,
sample_and_group :110, PointNetSetAbstraction :166, PointNetSetAbstractionMsg :210, PointNetFeaturePropagation :270.
Every function takes CUDA tensors and runs a hand-written gfx950 kernel through libpsg.so; there is
no PyTorch-op or CPU fallback.  Index results are int64 like the reference's.

The three nn.Module classes keep the reference's parameter layout (`mlp_convs.N`, `mlp_bns.N`, `conv_blocks.S.N`) so
that state_dicts are interchangeable.  Inside get_model their arithmetic is ONE fused kernel per module
(pointnet2_sem_seg.py), which is how the attack path uses them; called on their own (`module(xyz, points)`, the
reference's signatures and channel-major layouts) they run on the per-operator entry points of libpsg
(pointsecguard_amd/ops.py): eval mode only, gradients for the input features only.
"""
import torch
import torch.nn as nn

from pointsecguard_amd import ops, runtime


def _eval_only(module):
    if module.training:
        raise NotImplementedError("pointsecguard_amd implements the eval-mode attack path only; call .eval() "
                                  "(training-mode BatchNorm statistics are out of scope)")


def _point_major(t):
    """[B,C,N] (the reference's layout) -> contiguous float32 [B,N,C]; None stays None."""
    if t is None:
        return None
    runtime.require_cuda(t.contiguous(), "input", None)
    return t.permute(0, 2, 1).contiguous().float()


def square_distance(src, dst):
    """[B,N,3] x [B,M,3] -> [B,N,M] with the reference's expansion -2ab + |a|^2 + |b|^2 (bit-exact)."""
    return runtime.square_distance(src.contiguous().float(), dst.contiguous().float())


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] -> [B,S,C] / [B,S,K,C]."""
    shape = idx.shape
    flat = idx.reshape(shape[0], -1).to(torch.int32).contiguous()
    out = runtime.gather_points(points.contiguous().float(), flat)
    return out.view(*shape, points.shape[-1])


def farthest_point_sample(xyz, npoint):
    """xyz [B,N,3] -> centroid indices [B,npoint] int64.  The start index is drawn from the global
    CPU generator exactly like the reference (pointnet_util.py:75)."""
    B, N, _ = xyz.shape
    start = torch.randint(0, N, (B,), dtype=torch.long).to(torch.int32).to(xyz.device)
    return runtime.fps(xyz.contiguous().float(), npoint, start).long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    """First `nsample` indices (ascending) within `radius` of each query, padded with the first hit."""
    return runtime.ball_query(radius, nsample, xyz.contiguous().float(), new_xyz.contiguous().float()).long()


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    grouped_xyz_norm = grouped_xyz - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1)
    else:
        new_points = grouped_xyz_norm
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


class PointNetSetAbstraction(nn.Module):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super(PointNetSetAbstraction, self).__init__()
        if group_all:
            raise NotImplementedError("group_all set abstraction is not used by pointnet2_sem_seg (out of scope)")
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.group_all = group_all

    def forward(self, xyz, points):
        """xyz [B,3,N], points [B,D,N] or None -> new_xyz [B,3,S], new_points [B,C_last,S] (pointnet_util.py:181-207)."""
        _eval_only(self)
        xyz_pm, pts_pm = _point_major(xyz), _point_major(points)
        B, N, _ = xyz_pm.shape
        fps_idx = farthest_point_sample(xyz_pm, self.npoint)              # draws its start like the reference (:75)
        new_xyz = index_points(xyz_pm, fps_idx)
        gidx = query_ball_point(self.radius, self.nsample, xyz_pm, new_xyz).to(torch.int32).contiguous()
        ws, bs = zip(*[ops.fold_conv_bn(c, b) for c, b in zip(self.mlp_convs, self.mlp_bns)])
        out = ops.SetAbstractionMLP.apply(pts_pm, xyz_pm, new_xyz.contiguous(), gidx, list(ws), list(bs), False)
        return new_xyz.permute(0, 2, 1), out.permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):
    """Multi-scale grouping: one FPS sample, one ball query + shared MLP + max-pool per radius, channel concat
    (pointnet_util.py:210-267 of the reference).  Parameter layout conv_blocks.{scale}.{layer} / bn_blocks.*"""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super(PointNetSetAbstractionMsg, self).__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for widths in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            last_channel = in_channel + 3
            for out_channel in widths:
                convs.append(nn.Conv2d(last_channel, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last_channel = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points):
        """xyz [B,3,N], points [B,D,N] -> new_xyz [B,3,S], concatenated scale outputs [B,sum C,S] (pointnet_util.py:229-267;
        one FPS sample for all radii, grouped rows [features, rel_xyz])."""
        _eval_only(self)
        xyz_pm, pts_pm = _point_major(xyz), _point_major(points)
        fps_idx = farthest_point_sample(xyz_pm, self.npoint)
        new_xyz = index_points(xyz_pm, fps_idx).contiguous()
        outs = []
        for i, radius in enumerate(self.radius_list):
            gidx = query_ball_point(radius, self.nsample_list[i], xyz_pm, new_xyz).to(torch.int32).contiguous()
            ws, bs = zip(*[ops.fold_conv_bn(c, b) for c, b in zip(self.conv_blocks[i], self.bn_blocks[i])])
            outs.append(ops.SetAbstractionMLP.apply(pts_pm, xyz_pm, new_xyz, gidx, list(ws), list(bs), True))
        return new_xyz.permute(0, 2, 1), torch.cat(outs, dim=2).permute(0, 2, 1)


class PointNetFeaturePropagation(nn.Module):
    def __init__(self, in_channel, mlp):
        super(PointNetFeaturePropagation, self).__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,3,N] (fine), xyz2 [B,3,S] (coarse), points1 [B,D1,N] or None, points2 [B,D2,S] -> [B,C_last,N]
        (pointnet_util.py:281-320)."""
        _eval_only(self)
        x1, x2 = _point_major(xyz1), _point_major(xyz2)
        if x2.shape[1] < 3:
            raise NotImplementedError("fewer than 3 coarse points (the S == 1 branch, pointnet_util.py:298-299) is not reachable "
                                      "from pointnet2_sem_seg and not implemented")
        idx, w = runtime.three_nn(x1, x2)
        ws, bs = zip(*[ops.fold_conv_bn(c, b) for c, b in zip(self.mlp_convs, self.mlp_bns)])
        out = ops.FeaturePropagationMLP.apply(_point_major(points1), _point_major(points2), idx, w, list(ws), list(bs))
        return out.permute(0, 2, 1)
