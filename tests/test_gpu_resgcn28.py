"""BASELINE configs[3] at its real size: the 28-block ResGCN on a 4096-point room against tests/golden/gcn28_room.npz,
which tests/golden/make_golden_big.py produced by running the reference's own DenseDeepGCN
(/root/reference/ResGCN/sem_seg_dense/architecture.py:58-68) and colper.NB_attack (attacks/colper.py:17-39).

Bars: kNN graphs bit-exact against the CPU oracle on the fixture's features at dilation 1, 14, 27 (the oracle's
distance arithmetic is pinned to the reference's by gcn_room.npz) and >= 99.9 % of edges equal to the reference's own
tables (the rest are exact distance ties, which torch.topk orders arbitrarily); with the reference's 28 graphs
teacher-forced: block outputs and logits within 2e-4 relative to the tensor's largest magnitude, cost within 1e-4
relative, colour-gradient sign agreement >= 99.9 % with the misses below 1e-3 max|g|; two teacher-forced NB_attack
iterations bit-equal on >= 99.9 % of entries; free-running, every one of the 28 dynamic graphs (dilations 1..27, in
situ) equals the oracle's kNN of the GPU's own features."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
NB, N = 28, 4096


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda().contiguous()


@pytest.fixture(scope="module")
def g28():
    return dict(np.load(os.path.join(GOLDEN, "gcn28_room.npz")))


@pytest.fixture(scope="module")
def gcn28():
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import gcn28_state_dict
    return runtime.GCNModel(gcn28_state_dict(), NB), runtime.GCNWorkspace(1, N, NB)


def test_knn_on_reference_features(gcn28, g28):
    from oracle import resgcn
    _, ws = gcn28
    # block e + 1 builds its graph from block e's output with dilation e + 1: fixtures hold the outputs of e = 0, 1, 14, 27
    for e, d in ((0, 1), (1, 2), (14, 15), (27, 27), (14, 14)):
        f = g28["feat%d" % e]
        got = ws.knn(dev(f[None]), d)[0].cpu().numpy()
        assert np.array_equal(got, resgcn.knn_dilated(f, d)), (e, d)
        if e + 1 < NB and d == e + 1:
            ref = g28["graphs"][e + 1].astype(np.int32)
            assert (got == ref).mean() >= 0.999, (e, (got == ref).mean())
    xyz = np.ascontiguousarray(g28["room"][:, :3])
    got = ws.knn(dev(xyz[None]), 1)[0].cpu().numpy()
    assert np.array_equal(got, resgcn.knn_dilated(xyz, 1))
    assert (got == g28["graphs"][0]).mean() >= 0.999


def test_forward_backward_reference_graphs(gcn28, g28):
    from pointsecguard_amd import _lib, runtime
    model, ws = gcn28
    ws.set_graphs(dev(g28["graphs"].astype(np.int32)[:, None]))
    x0 = dev(g28["room"][None])
    logits = ws.forward(model, x0)
    feats = ws.feats()[0].cpu().numpy()
    for e in (0, 1, 14, 27):
        ref = g28["feat%d" % e]
        err = np.abs(feats[:, 64 * e:64 * e + 64] - ref) / np.abs(ref).max()
        # measured (tools/gcn28_diag.py): max 4.6e-7 / 4.0e-7 / 3.0e-5 / 7.7e-4 at blocks 0 / 1 / 14 / 27; the block-27 maximum
        # is ONE vertex (6 entries above 2e-4) whose max-pool picked another of two edges that agree to rounding
        # (EdgeConv maximises AFTER the BatchNorm affine: DESIGN.md 5d), everything else is below 6e-5
        assert np.quantile(err, 0.9999) <= 2e-4 and err.max() <= 2e-3, (e, err.max())
        assert (err > 2e-4).any(1).sum() <= 2, e
    ref = g28["logits"]
    assert np.abs(logits[0].cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
    labels = dev(g28["labels"].astype(np.int32)[None])
    dl = torch.empty_like(logits)
    cost = torch.zeros(1, device="cuda")
    _lib.call("psg_ce_logp_grad", runtime.ptr(logits), runtime.ptr(labels), 0, N, N, 13, 1.0 / N, runtime.ptr(dl),
              runtime.ptr(cost), runtime.stream())
    dx = ws.backward(model, dl)[0].cpu().numpy()
    ws.set_graphs(None)
    assert abs(cost.item() - float(g28["cost"])) <= 1e-4 * float(g28["cost"])
    ref = g28["dx"]
    gc, rc = dx[:, 3:6], ref[:, 3:6]
    agree = np.sign(gc) == np.sign(rc)
    assert agree.mean() >= 0.999, agree.mean()
    assert np.abs(rc[~agree]).max(initial=0.0) <= 1e-3 * np.abs(rc).max()
    nz = rc != 0
    assert np.median(np.abs(gc - rc)[nz] / np.abs(rc[nz])) < 1e-3


def test_two_nb_iterations_reference_graphs(gcn28, g28):
    from pointsecguard_amd import _lib, runtime
    model, ws = gcn28
    x0 = dev(g28["room"][None])
    ori = x0[:, :, 3:6].contiguous()
    labels = dev(g28["labels"].astype(np.int32)[None])
    for t, graphs, nxt in ((0, g28["graphs"], g28["state_it1"]), (1, g28["graphs_it1"], g28["adv_color_final"])):
        ws.set_graphs(dev(graphs.astype(np.int32)[:, None]))
        x0[:, :, 3:6] = dev(np.ascontiguousarray(g28["state_it%d" % t].transpose(0, 2, 1)))
        logits = ws.forward(model, x0)
        dl = torch.empty_like(logits)
        _lib.call("psg_ce_logp_grad", runtime.ptr(logits), runtime.ptr(labels), 0, N, N, 13, 1.0 / N, runtime.ptr(dl), None,
                  runtime.stream())
        dx = ws.backward(model, dl)
        _lib.call("psg_pgd_step", runtime.ptr(x0), runtime.ptr(dx), runtime.ptr(ori), None, 1, N, float(g28["alpha"]),
                  float(g28["eps"]), 1.0, 1 if t == 1 else 0, runtime.stream())
        torch.cuda.synchronize()
        got = np.ascontiguousarray(x0[:, :, 3:6].cpu().numpy().transpose(0, 2, 1))
        assert (got.view(np.uint32) == nxt.view(np.uint32)).mean() >= 0.999, t
    ws.set_graphs(None)


def test_free_running_graphs_in_situ(gcn28, g28):
    """All 28 dynamic graphs of one free-running forward (dilations 1..27 where the kernel actually runs them) equal the
    oracle's kNN of the features the GPU produced; the first blocks also stay close to the reference's tables."""
    from oracle import resgcn
    model, ws = gcn28
    ws.set_graphs(None)
    ws.forward(model, dev(g28["room"][None]))
    torch.cuda.synchronize()
    feats = ws.feats()[0].cpu().numpy()
    for e in range(NB):
        got = ws.edges(e)[0].cpu().numpy()
        src = g28["room"][:, :3] if e == 0 else np.ascontiguousarray(feats[:, 64 * (e - 1):64 * e])
        assert np.array_equal(got, resgcn.knn_dilated(src, 1 if e == 0 else e)), e
        if e <= 2:
            ref = g28["graphs"][e]
            overlap = np.mean([len(set(a) & set(b)) / 16.0 for a, b in zip(got, ref)])
            assert overlap >= 0.995, (e, overlap)


def test_fused_attack_invariants_28_blocks(gcn28, g28):
    """psg_gcn_nb_attack (hipGraph replay of the interior iterations) at configs[3]'s size: only colours move, by whole
    alpha steps inside the eps ball; the first iteration equals the teacher-free single step wherever graphs agree."""
    model, ws = gcn28
    images_np = np.ascontiguousarray(g28["room"][None].transpose(0, 2, 1))
    labels = dev(g28["labels"].astype(np.int32)[None])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        adv = ws.nb_attack(model, dev(images_np), labels, 0.3, 2 / 255, 5)
    side.synchronize()
    out = adv.cpu().numpy()
    assert np.array_equal(out[:, :3], images_np[:, :3]) and np.array_equal(out[:, 6:], images_np[:, 6:])
    steps = (out[:, 3:6] - images_np[:, 3:6]) / np.float32(2 / 255)
    assert np.abs(steps).max() <= 5 + 1e-3 and (np.abs(steps) > 0.5).mean() > 0.5
    # (no free-running comparison with the reference's colours: with dynamic graphs the 28-block network is chaotic -
    # every d-th neighbour of a sorted list changes wholesale when one near-tie moves, the edge overlap with the
    # reference's tables falls from 1.0 to 0.62 at block 5 and 0.08 at block 27 from rounding differences alone
    # (tools/gcn28_diag.py) - so parity is pinned with the reference's graphs teacher-forced, above)


def test_nb_attack_outcome_against_the_reference_runs(gcn28):
    """Free-running outcome at configs[3]'s own size (round 5).  tests/golden/gcn28_nb_outcome.npz and ..._t1.npz hold what the
    reference's experiment loop (sem_seg_dense/attacks.py:125-160) measured when ITS `NB_attack(eps=0.3, alpha=2/255, iters=50)`
    ran on four single-room batches with the fitted 28-block weights - twice: with eight intra-op threads and with one.  The
    two runs of the reference itself agree on 3 - 48 % of the clean per-point predictions, and their per-room accuracies differ
    by up to 0.12 clean and 0.04 adversarial: with dynamic graphs every d-th neighbour of a sorted list is re-selected when a
    near-tie moves (above), and the 70-step fit of the fixture weights leaves the predictions near chance (accuracy 0.08 - 0.2),
    so what ANY two evaluation orders agree on is the distribution, not the points.  The bar is therefore the reference's own
    spread: this implementation's distance from the eight-thread run may not exceed 2 x the one-thread run's distance from it
    (round 5 added 0.06 to that; round 6 does not), for the clean and the adversarial accuracy averaged over the rooms and room by room; what does not depend on the
    chaos is pinned tightly: the L2 distance of the returned room (eps / alpha / iteration count: within 2 %) and the
    invariants (only colours move)."""
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    g = dict(np.load(os.path.join(GOLDEN, "gcn28_nb_outcome.npz")))
    g1 = dict(np.load(os.path.join(GOLDEN, "gcn28_nb_outcome_t1.npz")))
    assert np.array_equal(g["seeds"], g1["seeds"])
    model, ws = gcn28
    ws.set_graphs(None)
    eps, alpha, iters = float(g["eps"]), float(g["alpha"]), int(g["iters"])
    side = torch.cuda.Stream()
    acc, adv_acc, adv_miou, dis, agree = [], [], [], [], []
    for si, seed in enumerate(g["seeds"]):
        r = make_rooms(1, int(seed))
        y = rule_labels(r)
        images_np = np.ascontiguousarray(r.transpose(0, 2, 1))
        labels = dev(y.astype(np.int32))
        pred = ws.forward(model, dev(r)).argmax(2)[0].cpu().numpy()
        agree.append(float((pred == g["r%d_clean_pred" % si]).mean()))
        with torch.cuda.stream(side):
            adv = ws.nb_attack(model, dev(images_np), labels, eps, alpha, iters)
        side.synchronize()
        apred = ws.forward(model, adv.permute(0, 2, 1).contiguous()).argmax(2)[0].cpu().numpy()
        inter = sum(int(((apred == c) & (y[0] == c)).sum()) for c in range(13))
        union = sum(int(((apred == c) | (y[0] == c)).sum()) for c in range(13))
        acc.append(float((pred == y[0]).mean()))
        adv_acc.append(float((apred == y[0]).mean()))
        adv_miou.append(inter / union)
        dis.append(float(torch.dist(adv, dev(images_np)).item()))
        out = adv.cpu().numpy()
        assert np.array_equal(out[:, :3], images_np[:, :3]) and np.array_equal(out[:, 6:], images_np[:, 6:])
        assert np.abs(out[:, 3:6] - images_np[:, 3:6]).max() <= eps + alpha + 1e-6
    acc, adv_acc, adv_miou, dis = (np.array(v) for v in (acc, adv_acc, adv_miou, dis))
    self_agree = [float((g["r%d_clean_pred" % i] == g1["r%d_clean_pred" % i]).mean()) for i in range(len(g["seeds"]))]
    print("ResGCN-28 NB outcome (this | reference 8 threads | reference 1 thread): clean predictions equal to the 8-thread run's on",
          np.round(agree, 3), "| the 1-thread run's on", np.round(self_agree, 3), "of the points; acc", np.round(acc, 3), g["acc"].round(3),
          g1["acc"].round(3), "| adv_acc", np.round(adv_acc, 3), g["adv_acc"].round(3), g1["adv_acc"].round(3), "| adv micro-IoU",
          np.round(adv_miou, 3), g["adv_miou"].round(3), g1["adv_miou"].round(3), "| L2", np.round(dis, 3), g["dis"].round(3), g1["dis"].round(3))
    for name, got in (("acc", acc), ("adv_acc", adv_acc), ("adv_miou", adv_miou)):
        ref, ref1 = g[name], g1[name]
        # (round 6: the "+ 0.06" of round 5 is gone - the bar is the reference's own spread alone; this implementation's attack is
        # bit-reproducible, so the margins below are the same on every box: mean 0.066 / 0.008 / 0.005 against bars of 0.110 /
        # 0.027 / 0.015, per room 0.099 / 0.062 / 0.035 against 0.236 / 0.086 / 0.048.  The test with teeth is the next one.)
        assert abs(got.mean() - ref.mean()) <= 2.0 * abs(ref1.mean() - ref.mean()), (name, got.mean(), ref.mean(), ref1.mean())
        assert np.abs(got - ref).max() <= 2.0 * np.abs(ref1 - ref).max(), (name, got, ref, ref1)
    assert np.abs(dis / g["dis"] - 1).max() <= 0.02 and np.abs(g1["dis"] / g["dis"] - 1).max() <= 0.02


def test_nb_attack_outcome_on_fitted_weights_against_the_reference_runs():
    """Free-running outcome at configs[3]'s own size on weights the network is RIGHT with (round 6; the round-5 test above
    keeps the 70-step weights, on which the reference predicts near chance and disagrees with itself, so its bars are the
    reference's own spread).  tests/golden/gcn28_weights_fit.npz: the reference's 28-block DenseDeepGCN fitted in eval mode
    with small residual branches (make_golden_big.py: fit_gcn28_frozen) - clean accuracy 0.91 on held-out 4096-point rooms,
    and stable: tests/golden/gcn28_nb_outcome_fit.npz / ..._fit_t1.npz hold what the reference's experiment loop
    (sem_seg_dense/attacks.py:125-160) measured when ITS `NB_attack(eps=0.3, alpha=2/255, iters=50)` ran on four single-room
    batches, with six intra-op threads and with one, and the two runs agree with each other on >= 0.95 of the clean
    per-point predictions (0.958 - 0.966), within 0.004 on every room's clean accuracy (0.0032) and 0.012 on its adversarial
    accuracy (0.0103) (asserted
    below, so the premise of the bars is part of the test).  The attack takes the accuracy from 0.91 to 0.47: nothing here
    is degenerate.  Bars, ABSOLUTE and per room (the ones test_nb_b8_statistical_parity_32_rooms uses for PointNet++): clean
    accuracy within 0.01 of the reference's, clean predictions equal on >= 0.95 of the points, adversarial accuracy and
    adversarial micro-IoU (sum I / sum U, attacks.py:159-160) within 0.03, L2 distance of the returned room within 2 %;
    only colours move, inside the eps ball (+ the un-projected last step)."""
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import gcn28_fit_state_dict, make_rooms, rule_labels
    g = dict(np.load(os.path.join(GOLDEN, "gcn28_nb_outcome_fit.npz")))
    g1 = dict(np.load(os.path.join(GOLDEN, "gcn28_nb_outcome_fit_t1.npz")))
    assert np.array_equal(g["seeds"], g1["seeds"]) and int(g["threads"]) != int(g1["threads"])
    n_rooms = len(g["seeds"])
    # ---- the premise: the reference agrees with itself on these weights
    self_agree = [float((g["r%d_clean_pred" % i] == g1["r%d_clean_pred" % i]).mean()) for i in range(n_rooms)]
    assert min(self_agree) >= 0.95, self_agree
    assert np.abs(g["acc"] - g1["acc"]).max() <= 0.004 and np.abs(g["adv_acc"] - g1["adv_acc"]).max() <= 0.012
    assert g["acc"].min() >= 0.8 and g["adv_acc"].max() <= 0.6
    model = runtime.GCNModel(gcn28_fit_state_dict(), 28)
    ws = runtime.GCNWorkspace(1, 4096, 28)
    eps, alpha, iters = float(g["eps"]), float(g["alpha"]), int(g["iters"])
    side = torch.cuda.Stream()
    acc, adv_acc, adv_miou, dis, agree = [], [], [], [], []
    for si, seed in enumerate(g["seeds"]):
        r = make_rooms(1, int(seed))
        y = rule_labels(r)
        images_np = np.ascontiguousarray(r.transpose(0, 2, 1))
        labels = dev(y.astype(np.int32))
        pred = ws.forward(model, dev(r)).argmax(2)[0].cpu().numpy()
        agree.append(float((pred == g["r%d_clean_pred" % si]).mean()))
        with torch.cuda.stream(side):
            adv = ws.nb_attack(model, dev(images_np), labels, eps, alpha, iters)
        side.synchronize()
        apred = ws.forward(model, adv.permute(0, 2, 1).contiguous()).argmax(2)[0].cpu().numpy()
        inter = sum(int(((apred == c) & (y[0] == c)).sum()) for c in range(13))
        union = sum(int(((apred == c) | (y[0] == c)).sum()) for c in range(13))
        acc.append(float((pred == y[0]).mean()))
        adv_acc.append(float((apred == y[0]).mean()))
        adv_miou.append(inter / union)
        dis.append(float(torch.dist(adv, dev(images_np)).item()))
        out = adv.cpu().numpy()
        assert np.array_equal(out[:, :3], images_np[:, :3]) and np.array_equal(out[:, 6:], images_np[:, 6:])
        assert np.abs(out[:, 3:6] - images_np[:, 3:6]).max() <= eps + alpha + 1e-6
    acc, adv_acc, adv_miou, dis = (np.array(v) for v in (acc, adv_acc, adv_miou, dis))
    print("ResGCN-28 NB outcome, fitted weights (this | reference %d threads | reference %d thread): clean predictions equal to the"
          " reference's on" % (int(g["threads"]), int(g1["threads"])), np.round(agree, 4), "(its own two runs:", np.round(self_agree, 4),
          ") | acc", np.round(acc, 4), g["acc"].round(4), g1["acc"].round(4), "| adv_acc", np.round(adv_acc, 4), g["adv_acc"].round(4),
          g1["adv_acc"].round(4), "| adv micro-IoU", np.round(adv_miou, 4), g["adv_miou"].round(4), g1["adv_miou"].round(4),
          "| L2", np.round(dis, 3), g["dis"].round(3), g1["dis"].round(3))
    assert min(agree) >= 0.95, agree
    assert np.abs(acc - g["acc"]).max() <= 0.01, (acc, g["acc"])
    assert np.abs(adv_acc - g["adv_acc"]).max() <= 0.03, (adv_acc, g["adv_acc"])
    assert np.abs(adv_miou - g["adv_miou"]).max() <= 0.03, (adv_miou, g["adv_miou"])
    assert np.abs(dis / g["dis"] - 1).max() <= 0.02, (dis, g["dis"])


def test_nb_attack_28_blocks_is_bit_reproducible(gcn28, g28):
    """Round 5: no float atomics are left in the ResGCN forward / input-gradient backward (the prediction head's column sums
    are added in chunk order, the global max-pool's transpose has one writer per point), so the free-running attack - chaotic
    as it is - returns the same bits every time."""
    model, ws = gcn28
    ws.set_graphs(None)
    images_np = np.ascontiguousarray(g28["room"][None].transpose(0, 2, 1))
    labels = dev(g28["labels"].astype(np.int32)[None])
    side = torch.cuda.Stream()
    outs = []
    for _ in range(2):
        with torch.cuda.stream(side):
            adv = ws.nb_attack(model, dev(images_np), labels, 0.3, 2 / 255, 6)
        side.synchronize()
        outs.append(adv.cpu().numpy())
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
