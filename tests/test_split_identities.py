"""The two algebraic identities behind the round-5 kernels, stated with the ORACLE's own primitives (CPU, no GPU):

  SA split   W1 . [x_j - c_i ; f_j] + b = (W1f . f_j + b) gathered through the group table + W1x . (x_j - c_i)
             (pointsecguard_amd/csrc/psg_pn2_kernels.cuh: sa_layer1_split; reference pointnet_util.py:126-140, 200-205)
  FP split   W1 . [f1 ; interp(f2)] + b = W1a . f1 + interp(W1b . f2) + b: the 3-NN interpolation is linear, so the
             interpolated part of the first layer commutes with it and can run per COARSE point
             (fp_layer1_split; reference pointnet_util.py:296-319)

and their transposes (what the backward kernels apply: dZ1 rows gathered per coarse point, then W1b^T).  Both sides are
computed in fp32 by the oracle's C routines; they differ by the rounding order only."""
import numpy as np
import pytest

from oracle import pn2


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def geometry():
    rng = np.random.default_rng(5)
    fine = rng.random((1024, 3), dtype=np.float32)
    coarse = fine[pn2.fps(fine, 256, 7)]
    idx, w = pn2.three_nn(fine, coarse)
    return rng, fine, coarse, idx, w


def interp(f2, idx, w):
    return (f2[idx] * w[:, :, None]).sum(axis=1, dtype=np.float32).astype(np.float32)


def test_fp_first_layer_commutes_with_the_interpolation(geometry):
    rng, fine, coarse, idx, w = geometry
    c1, c2, cout = 64, 256, 256                                     # fp2 of the SSG network: 64 skip + 256 interpolated -> 256
    f1 = rng.standard_normal((fine.shape[0], c1)).astype(np.float32)
    f2 = rng.standard_normal((coarse.shape[0], c2)).astype(np.float32)
    W = (rng.standard_normal((cout, c1 + c2)) / np.sqrt(c1 + c2)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    whole = pn2.linear_fwd(np.concatenate([f1, interp(f2, idx, w)], axis=1), W, b, relu=False)
    zero = np.zeros(cout, np.float32)
    t = pn2.linear_fwd(f2, np.ascontiguousarray(W[:, c1:]), zero, relu=False)          # per COARSE point: 4 x fewer rows
    split = pn2.linear_fwd(f1, np.ascontiguousarray(W[:, :c1]), b, relu=False) + interp(t, idx, w)
    assert rel(split, whole) < 2e-6
    # fp1: no skip part at all - the layer is the interpolation of T plus the bias
    whole1 = pn2.linear_fwd(interp(f2, idx, w), np.ascontiguousarray(W[:, c1:]), b, relu=False)
    assert rel(interp(t, idx, w) + b, whole1) < 2e-6


def test_fp_split_transposes(geometry):
    """Backward: the whole layer's input gradient is [W1a^T dZ ; W1b^T dZ] per FINE point, the second part then scattered to the
    coarse points with the interpolation weights.  Split: scatter dZ itself, then W1b^T per COARSE point."""
    rng, fine, coarse, idx, w = geometry
    c1, c2, cout = 64, 256, 256
    W = (rng.standard_normal((cout, c1 + c2)) / np.sqrt(cout)).astype(np.float32)
    dz = rng.standard_normal((fine.shape[0], cout)).astype(np.float32)

    def scatter(rows):
        out = np.zeros((coarse.shape[0], rows.shape[1]), np.float64)
        for k in range(3):
            np.add.at(out, idx[:, k], rows * w[:, k:k + 1])
        return out.astype(np.float32)

    y = np.zeros((fine.shape[0], cout), np.float32)
    whole = scatter(pn2.linear_bwd(dz, y, W, relu=False)[:, c1:])
    yc = np.zeros((coarse.shape[0], cout), np.float32)
    split = pn2.linear_bwd(scatter(dz), yc, np.ascontiguousarray(W[:, c1:]), relu=False)
    assert rel(split, whole) < 5e-6


def test_sa_first_layer_splits_into_a_per_point_product_and_an_xyz_chunk(geometry):
    rng, fine, coarse, _, _ = geometry
    d, cout, k = 64, 64, 32                                          # sa2 of the SSG network: 3 + 64 -> 64
    gidx = pn2.ball_query(0.2, k, fine, coarse)
    gidx = np.where(gidx == fine.shape[0], gidx[:, :1], gidx)
    feats = rng.standard_normal((fine.shape[0], d)).astype(np.float32)
    W = (rng.standard_normal((cout, 3 + d)) / np.sqrt(3 + d)).astype(np.float32)          # reference order: [rel_xyz, feats]
    b = rng.standard_normal(cout).astype(np.float32)
    rel_xyz = (fine[gidx] - coarse[:, None, :]).reshape(-1, 3)
    rows = np.concatenate([rel_xyz, feats[gidx].reshape(-1, d)], axis=1)
    whole = pn2.linear_fwd(rows, W, b, relu=False)
    t = pn2.linear_fwd(feats, np.ascontiguousarray(W[:, 3:]), b, relu=False)            # per POINT: 8 x fewer rows
    split = t[gidx].reshape(-1, cout) + pn2.linear_fwd(rel_xyz, np.ascontiguousarray(W[:, :3]), np.zeros(cout, np.float32), relu=False)
    assert rel(split, whole) < 2e-6
