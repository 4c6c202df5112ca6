"""CPU tests: the ResGCN oracle (oracle/resgcn.py) against fixtures generated from the reference
(tests/golden/make_golden_gcn.py)."""
import numpy as np
import pytest

from oracle import resgcn


def test_pairwise_distance_bits(golden_gcn_room):
    g = golden_gcn_room
    d = resgcn.pairwise_distance(g["feat0"][:64])
    assert np.array_equal(d.view(np.uint32), g["pd_bits"])      # torch_edge.py:41-43, fp32 order pinned


def test_knn_graphs(golden_gcn_room):
    g = golden_gcn_room
    # exact-distance ties are the only allowed differences (torch.topk leaves their order unspecified)
    assert (resgcn.knn_dilated(g["feat0"], 27) == g["nbr_d27"]).mean() >= 0.9998
    assert (resgcn.knn_dilated(g["room"][:, :3], 1) == g["nbr0"]).mean() >= 0.9998
    assert (resgcn.knn_dilated(g["feat0"], 1) == g["nbr1"]).mean() >= 0.9998


def test_forward_backward_teacher_forced(golden_gcn_room, gcn_oracle):
    g = golden_gcn_room
    graphs = [g["nbr%d" % e].astype(np.int32) for e in range(5)]
    logits, cache = gcn_oracle.forward(g["room"], graphs=graphs)
    for e in range(5):
        assert np.abs(cache["feats"][:, 64 * e:64 * e + 64] - g["feat%d" % e]).max() <= 1e-4
    assert np.abs(logits - g["logits"]).max() <= 1e-4
    dl, cost = resgcn.ce_mean_grad(logits, g["labels"].astype(np.int64))
    assert abs(cost - float(g["cost"])) <= 1e-5
    dx = gcn_oracle.backward(cache, dl)
    ref = g["dx"]
    assert (np.sign(dx) == np.sign(ref)).mean() >= 0.999
    assert np.abs(dx - ref).max() <= 1e-3 * np.abs(ref).max()


def test_forward_free_running_graph_overlap(golden_gcn_room, gcn_oracle):
    g = golden_gcn_room
    _, cache = gcn_oracle.forward(g["room"])
    for e in range(5):
        ov = np.mean([len(set(a) & set(b)) / 16.0 for a, b in zip(cache["nbr"][e], g["nbr%d" % e])])
        assert ov >= 0.99, (e, ov)


@pytest.mark.parametrize("t", [0, 3])
def test_nb_attack_step(golden_gcn_nb, gcn_oracle, t):
    g = golden_gcn_nb
    iters = int(g["iters"])
    xr = g["rooms"][0]
    graphs = [g["graphs_it%d" % t][e].astype(np.int32) for e in range(5)]
    nxt = g["adv_color_final"][0].T if t == iters - 1 else g["state_it%d" % (t + 1)][0].T
    got, _, _, _ = resgcn.nb_step(gcn_oracle, xr, np.ascontiguousarray(g["state_it%d" % t][0].T), xr[:, 3:6].copy(),
                                  g["labels"][0].astype(np.int64), float(g["alpha"]), float(g["eps"]), t == iters - 1,
                                  graphs=graphs)
    assert (np.ascontiguousarray(got).view(np.uint32) == np.ascontiguousarray(nxt).view(np.uint32)).mean() >= 0.999


def _gcn_nu_check(orc, g, t, nb, tv):
    x = g["rooms"][0]
    w = g["s%d_w_before" % t][0]
    if t == 0:
        m, v = np.zeros_like(w), np.zeros_like(w)
    else:
        m, v = g["s%d_m" % (t - 1)][0], g["s%d_v" % (t - 1)][0]
    graphs = [g["graphs_s%d" % t][e].astype(np.int32) for e in range(5)]
    r = resgcn.nu_step(orc, x, x, w, m, v, int(g["s%d_t" % t]), g["labels"][0].astype(np.int64), float(g["c"]),
                       float(g["kappa"]), float(g["s%d_lr" % t]), nb, targeted_variant=tv,
                       mask=g["mask"] if tv else None, target=int(g["target"]) if tv else None, graphs=graphs)
    ref = g["s%d_grad" % t][0]
    assert abs(r["cost"] - g["costs"][t]) <= 1e-5 * abs(g["costs"][t]) + 1e-3
    assert (np.abs(r["grad_w"] - ref) <= 1e-2 * np.abs(ref).max()).mean() >= 0.995
    assert (np.abs(r["w"] - g["s%d_w_after" % t][0]) <= 1e-4).mean() >= 0.995


@pytest.mark.parametrize("t", [0, 1, 2])
def test_gcn_nu_attack_step(golden_gcn_nu, gcn_oracle, t):
    """colper.NU_attack: c*f(logits) + 1e-4*Smooth(adv, adv) + L2, torch Adam; teacher-forced graphs."""
    _gcn_nu_check(gcn_oracle, golden_gcn_nu, t, 10, False)


@pytest.mark.parametrize("t", [0, 20, 22])
def test_gcn_tar_nu_attack_step(golden_gcn_tarnu, gcn_oracle, t):
    """tcolper.tar_NU_attack: masked tar_f on row 0 + 1e-4*Smooth + c*L2."""
    _gcn_nu_check(gcn_oracle, golden_gcn_tarnu, t, 5, True)


def test_oracle_knn_at_baseline_size_vs_reference_tables():
    """configs[3] size (4096 points, dilations up to 27): the oracle's kNN on the reference's own block outputs equals the
    reference's neighbour tables (tests/golden/gcn28_room.npz) except at exact distance ties."""
    import os
    from conftest import GOLDEN
    from oracle import resgcn
    g = np.load(os.path.join(GOLDEN, "gcn28_room.npz"))
    for e in (0, 1, 14):
        got = resgcn.knn_dilated(g["feat%d" % e], e + 1)
        ref = g["graphs"][e + 1].astype(np.int32)
        assert (got == ref).mean() >= 0.999, (e, (got == ref).mean())
    xyz = np.ascontiguousarray(g["room"][:, :3])
    assert (resgcn.knn_dilated(xyz, 1) == g["graphs"][0]).mean() >= 0.999
