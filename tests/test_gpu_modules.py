"""Stand-alone forwards of the reference's public modules (the import surface of SURVEY.md 8b) against
tests/golden/modules.npz, which tests/golden/make_golden_modules.py produced by running the reference's own classes
(PointNet/models/pointnet_util.py:166-320, ResGCN/gcn_lib/dense/torch_vertex.py:8-115) with fixture weights.

Bars: sampled coordinates bit-equal (FPS / gather); pooled / interpolated outputs within 1e-4 of the output's largest
magnitude; input gradients within 1e-3 of the gradient's largest magnitude on >= 99.9 % of entries (a max-pool that picks
the other of two values equal to rounding moves one entry by its full cotangent) and median relative error < 1e-3;
pairwise_distance bit-equal; the dynamic block's graph equals the reference's except at exact ties."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(GOLDEN, "modules.npz")))


@pytest.fixture(scope="module")
def room():
    r = np.load(os.path.join(GOLDEN, "pn2_room.npz"))["room"]
    return torch.from_numpy(np.ascontiguousarray(r.T))[None].cuda()        # [1, 9, 4096]


def sub_state(sd, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k.startswith(prefix)}


def close(got, ref, tol=1e-4):
    return np.abs(got - ref).max() <= tol * np.abs(ref).max()


def grad_ok(got, ref):
    err = np.abs(got - ref)
    assert (err <= 1e-3 * np.abs(ref).max()).mean() >= 0.999, (err > 1e-3 * np.abs(ref).max()).mean()
    nz = ref != 0
    assert np.array_equal(got != 0, nz) or ((got != 0) != nz).mean() <= 1e-3
    assert np.median(err[nz] / np.abs(ref[nz])) < 1e-3


def test_set_abstraction_standalone(g, room, weights_sd):
    from pointsecguard_amd.models.pointnet_util import PointNetSetAbstraction
    sa = PointNetSetAbstraction(1024, 0.1, 32, 9 + 3, [32, 32, 64], False)
    sa.load_state_dict(sub_state(weights_sd, "sa1."), strict=True)
    sa = sa.cuda().eval()
    pts = room.clone().requires_grad_(True)
    torch.manual_seed(5)                                              # the reference's FPS start draw
    new_xyz, new_pts = sa(room[:, :3].contiguous(), pts)
    assert np.array_equal(new_xyz.detach().cpu().numpy(), g["sa_new_xyz"])
    assert close(new_pts.detach().cpu().numpy(), g["sa_out"])
    (new_pts * torch.from_numpy(g["sa_cot"]).cuda()).sum().backward()
    grad_ok(pts.grad.cpu().numpy(), g["sa_dpoints"])
    with pytest.raises(NotImplementedError):
        sa.train()(room[:, :3].contiguous(), pts)


def test_set_abstraction_msg_standalone(g, room):
    from pointsecguard_amd.models.pointnet_util import PointNetSetAbstractionMsg
    from pointsecguard_amd.synthetic import msg_state_dict
    msa = PointNetSetAbstractionMsg(1024, [0.05, 0.1], [16, 32], 9, [[16, 16, 32], [32, 32, 64]])
    msa.load_state_dict(sub_state(msg_state_dict(77), "sa1."), strict=True)
    msa = msa.cuda().eval()
    pts = room.clone().requires_grad_(True)
    torch.manual_seed(6)
    new_xyz, new_pts = msa(room[:, :3].contiguous(), pts)
    assert np.array_equal(new_xyz.detach().cpu().numpy(), g["msa_new_xyz"])
    assert close(new_pts.detach().cpu().numpy(), g["msa_out"])
    (new_pts * torch.from_numpy(g["msa_cot"]).cuda()).sum().backward()
    grad_ok(pts.grad.cpu().numpy(), g["msa_dpoints"])


def test_feature_propagation_standalone(g, room, weights_sd):
    from pointsecguard_amd.models.pointnet_util import PointNetFeaturePropagation
    fp = PointNetFeaturePropagation(320, [256, 128])
    fp.load_state_dict(sub_state(weights_sd, "fp2."), strict=True)
    fp = fp.cuda().eval()
    xyz = room[:, :3]
    p1 = torch.from_numpy(g["fp_p1"]).cuda().requires_grad_(True)
    p2 = torch.from_numpy(g["fp_p2"]).cuda().requires_grad_(True)
    y = fp(xyz[:, :, :1024].contiguous(), xyz[:, :, :256].contiguous(), p1, p2)
    assert close(y.detach().cpu().numpy(), g["fp_out"])
    (y * torch.from_numpy(g["fp_cot"]).cuda()).sum().backward()
    grad_ok(p1.grad.cpu().numpy(), g["fp_dp1"])
    grad_ok(p2.grad.cpu().numpy(), g["fp_dp2"])
    # no skip features (fp1 of the network): points1 = None
    y2 = fp.__class__(256, [128]).cuda().eval()(xyz[:, :, :1024].contiguous(), xyz[:, :, :256].contiguous(), None, p2.detach())
    assert y2.shape == (1, 128, 1024) and torch.isfinite(y2).all()


def test_graph_convolutions_standalone(g, gcn_weights_sd):
    from pointsecguard_amd.resgcn.gcn_lib.dense import EdgeConv2d, MRConv2d, ResDynBlock2d, PlainDynBlock2d, DenseDynBlock2d
    sd = sub_state(gcn_weights_sd, "backbone.2.")
    blk = ResDynBlock2d(64, 16, 3, "edge", "relu", "batch", True, True, 0.0)
    blk.load_state_dict(sd, strict=True)
    blk = blk.cuda().eval()
    x = torch.from_numpy(g["gcn_x"]).cuda()
    edge = torch.from_numpy(g["gcn_edge"].astype(np.int64)).cuda()
    cot = torch.from_numpy(g["gcn_cot"]).cuda()
    # EdgeConv2d on the reference's own edge_index
    xin = x.clone().requires_grad_(True)
    ye = blk.body.gconv(xin, edge)
    assert close(ye.detach().cpu().numpy(), g["gcn_edgeconv_out"])
    (ye * cot).sum().backward()
    grad_ok(xin.grad.cpu().numpy(), g["gcn_edgeconv_dx"])
    # MRConv2d, same parameters (same layout)
    mr = MRConv2d(64, 64, "relu", "batch", True)
    mr.load_state_dict(blk.body.gconv.state_dict(), strict=True)
    mr = mr.cuda().eval()
    xin = x.clone().requires_grad_(True)
    ym = mr(xin, edge)
    assert close(ym.detach().cpu().numpy(), g["gcn_mr_out"])
    (ym * cot).sum().backward()
    grad_ok(xin.grad.cpu().numpy(), g["gcn_mr_dx"])
    # the dynamic residual block builds its own dilated graph (d = 3) from x
    torch.manual_seed(7)
    got_edge = blk.body.dilated_knn_graph(x)
    assert (got_edge[0].cpu().numpy() == g["gcn_edge"][0]).mean() >= 0.9998
    xin = x.clone().requires_grad_(True)
    yb = blk(xin)
    ref = g["gcn_block_out"]
    assert (np.abs(yb.detach().cpu().numpy() - ref) <= 1e-4 * np.abs(ref).max()).mean() >= 0.999
    (yb * cot).sum().backward()
    err = np.abs(xin.grad.cpu().numpy() - g["gcn_block_dx"])
    assert (err <= 1e-3 * np.abs(g["gcn_block_dx"]).max()).mean() >= 0.998
    # plain / dense blocks are compositions of the same body
    plain = PlainDynBlock2d(64, 16, 3, "edge", "relu", "batch", True, True, 0.0)
    plain.load_state_dict(sd, strict=True)
    yp = plain.cuda().eval()(x)
    assert torch.allclose(yp + x, blk(x), atol=1e-5)
    dense = DenseDynBlock2d(64, 64, 16, 3, "edge", "relu", "batch", True, True, 0.0)
    dense.load_state_dict(sd, strict=True)
    yd = dense.cuda().eval()(x)
    assert yd.shape == (1, 128, 1024, 1) and torch.equal(yd[:, :64], x) and torch.allclose(yd[:, 64:], yp, atol=1e-5)
    # an edge_index the kernels cannot honour is refused, not silently mis-read: a centre row that is not the vertex itself
    # (the reference gathers x_i through edge_index[1], torch_vertex.py:17,32) and neighbours outside the room
    bad_centre = edge.clone()
    bad_centre[1, 0, 5, 2] = 7
    with pytest.raises(NotImplementedError):
        blk.body.gconv(x, bad_centre)
    bad_nbr = edge.clone()
    bad_nbr[0, 0, 3, 0] = x.shape[2]
    with pytest.raises(ValueError):
        mr(x, bad_nbr)


def test_pairwise_distance_bits(golden_gcn_room):
    from pointsecguard_amd.resgcn.gcn_lib.dense.torch_edge import pairwise_distance
    f = torch.from_numpy(golden_gcn_room["feat0"][:64][None]).cuda()
    d = pairwise_distance(f)[0].cpu().numpy()
    assert np.array_equal(d.view(np.uint32), golden_gcn_room["pd_bits"])       # the reference's own bits (torch_edge.py:41-43)


def test_gemm_interior_tiles_equal_the_general_path():
    """psg_gemm.cuh stages the operands of INTERIOR tiles (whole tile inside both matrices, 16-byte aligned rows, K a multiple
    of 32) without bounds tests and reads their epilogue terms as float4 (round 4).  The same product through the general
    path - forced by a row stride that is not a multiple of 4, by a ragged K, by M / rows that leave edge tiles - must be
    bit-identical on the common outputs, ReLU bits included; values are checked against float64."""
    from pointsecguard_amd import _lib, runtime
    rng = np.random.default_rng(42)
    rows, K, M = 512, 96, 256
    x = rng.standard_normal((rows, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) * 0.2).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, M).astype(np.float32)
    sh = rng.standard_normal(M).astype(np.float32)
    dw, db, dsc, dsh = (torch.from_numpy(a).cuda() for a in (w, b, sc, sh))

    def run(x_np, ld, n_rows, k, m, w_t):
        buf = torch.zeros(n_rows, ld, device="cuda")
        buf[:, :x_np.shape[1]] = torch.from_numpy(x_np).cuda()
        out = torch.empty(n_rows, m, device="cuda")
        mk = torch.zeros(n_rows, (m + 31) // 32, dtype=torch.int32, device="cuda")
        _lib.call("psg_pw_mlp_fwd", runtime.ptr(buf), ld, n_rows, k, runtime.ptr(w_t), runtime.ptr(db[:m].contiguous()), 1, m,
                  runtime.ptr(out), m, runtime.ptr(mk), runtime.ptr(dsc[:m].contiguous()), runtime.ptr(dsh[:m].contiguous()), runtime.stream())
        torch.cuda.synchronize()
        return out.cpu().numpy(), mk.cpu().numpy()

    ref, ref_bits = run(x, K, rows, K, M, dw)                                   # every tile interior
    z = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + b, 0) * sc + sh
    assert np.abs(ref - z).max() < 2e-4
    got, bits = run(x, K + 1, rows, K, M, dw)                                   # odd row stride: general path everywhere
    assert np.array_equal(got, ref) and np.array_equal(bits, ref_bits)
    got, bits = run(x[:rows - 3], K, rows - 3, K, M, dw)                        # ragged rows: the last row tile is an edge tile
    assert np.array_equal(got, ref[:rows - 3]) and np.array_equal(bits, ref_bits[:rows - 3])
    m2 = M - 40                                                                 # ragged M: the last column tile is an edge tile
    got, bits = run(x, K, rows, K, m2, dw[:m2].contiguous())
    assert np.array_equal(got, ref[:, :m2])
    # ragged K (general path) against the interior path on the zero-padded operands
    k2 = K - 5
    w2 = dw[:, :k2].contiguous()
    got, _ = run(x[:, :k2], K, rows, k2, M, w2)
    xp = x.copy(); xp[:, k2:] = 0
    wp = w.copy(); wp[:, k2:] = 0
    pad, _ = run(xp, K, rows, K, M, torch.from_numpy(wp).cuda())
    assert np.array_equal(got, pad)
