"""RandLA-Net network: the HIP path against the source-reading torch restatement
(oracle/randla_net.py; PARITY UNPINNED: the TensorFlow-1 reference cannot run here).  Tolerances: logits 2e-4 relative to
their magnitude, loss 1e-4 relative, colour gradient 1e-3 of its maximum on >= 99.5 % of entries (a max-pool or
leaky-ReLU sign that flips on a last-bit difference moves single entries visibly), BIM colours after teacher-forced
steps equal on >= 99.5 % of entries."""
import numpy as np
import pytest

from oracle import randla, randla_net
from pointsecguard_amd.synthetic import randla_layer_specs, randla_params

N = 8192


@pytest.fixture(scope="module")
def cloud():
    rng = np.random.default_rng(12)
    xyz = (rng.random((1, N, 3), dtype=np.float32) * np.array([4, 3, 3], np.float32)).astype(np.float32)
    rgb = rng.random((N, 3), dtype=np.float32)
    labels = rng.integers(0, 13, N)
    pts, neigh, pools, ups = randla.pyramid(xyz)
    pyr = ([p[0] for p in pts], [n[0] for n in neigh], [p[0] for p in pools], [u[0] for u in ups])
    return xyz[0], rgb, labels, pyr


def test_layer_specs_and_oracle_shapes(cloud):
    specs = randla_layer_specs()
    assert len(specs) == 55 and specs[0] == ("fc0", 6, 8, True) and specs[-1] == ("fc", 32, 13, False)
    assert [s for s in specs if s[0] == "Decoder_layer_0"][0][1:3] == (1536, 512)
    xyz, rgb, labels, pyr = cloud
    orc = randla_net.RandLAOracle(randla_params(3))
    loss, logits, g = randla_net.loss_and_grad(orc, xyz, rgb, labels, pyr)
    assert logits.shape == (N, 13) and g.shape == (N, 3) and np.isfinite(logits).all() and loss > 0
    # finite-difference check of the restatement's gradient along a random direction, on a float64 copy of the network
    import torch
    o64 = randla_net.RandLAOracle(randla_params(3), dtype=torch.float64)
    _, _, g64 = randla_net.loss_and_grad(o64, xyz, rgb, labels, pyr)
    assert np.abs(g64 - g).max() <= 1e-3 * np.abs(g64).max()
    rng = np.random.default_rng(0)
    dirn = rng.standard_normal(rgb.shape)
    h = 1e-6
    lp, _, _ = randla_net.loss_and_grad(o64, xyz, rgb.astype(np.float64) + h * dirn, labels, pyr)
    lm, _, _ = randla_net.loss_and_grad(o64, xyz, rgb.astype(np.float64) - h * dirn, labels, pyr)
    fd, an = (lp - lm) / (2 * h), float((g64 * dirn).sum())
    assert abs(fd - an) <= 1e-3 * max(abs(an), 1.0), (fd, an)


def test_bim_step_rules():
    xs = np.full((4, 3), 0.5, np.float32)
    g = np.array([[1, -1, 0], [2, 2, 2], [-3, 0, 1], [0, 0, 0]], np.float32)
    out = randla_net.bim_step(xs, xs, g, eps=0.05, alpha=0.1)
    assert np.allclose(out, xs + 0.05 * np.sign(g))
    out = randla_net.bim_step(xs, xs, g, eps=10.0, alpha=0.1, metric="l_2")
    assert np.isclose(np.linalg.norm(out - xs), 0.1, rtol=1e-5)
    out = randla_net.bim_step(xs, xs, g, eps=0.03, alpha=0.1, metric="l_2")
    assert np.isclose(np.linalg.norm(out - xs), 0.03, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ GPU
def dev(a, dt=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda().contiguous()


@pytest.fixture(scope="module")
def gpu(cloud):
    from pointsecguard_amd.randla import network
    params = randla_params(3)
    return network.RandLAModel(params), network.RandLAWorkspace(N), randla_net.RandLAOracle(params)


@pytest.mark.gpu
def test_gpu_forward_backward_vs_oracle(cloud, gpu):
    import torch
    from pointsecguard_amd.randla import network
    xyz, rgb, labels, pyr = cloud
    model, ws, orc = gpu
    ws.set_cloud(dev(xyz))
    for lvl in range(5):
        assert np.array_equal(ws.index(0, lvl).cpu().numpy(), pyr[1][lvl])
        assert np.array_equal(ws.index(1, lvl).cpu().numpy(), pyr[3][lvl][:, 0])
    feats = dev(np.concatenate([xyz, rgb], 1))
    logits = ws.forward(model, feats)
    loss_ref, logits_ref, g_ref = randla_net.loss_and_grad(orc, xyz, rgb, labels, pyr)
    assert np.abs(logits.cpu().numpy() - logits_ref).max() <= 2e-4 * max(1.0, np.abs(logits_ref).max())
    loss, dl = network.colper_grad(logits, dev(labels.astype(np.int32)))
    assert abs(loss.item() - loss_ref) <= 1e-4 * abs(loss_ref)
    df = ws.backward(model, dl).cpu().numpy()
    torch.cuda.synchronize()
    g = df[:, 3:6]
    ok = np.abs(g - g_ref) <= 1e-3 * np.abs(g_ref).max()
    assert ok.mean() >= 0.995, ok.mean()
    assert np.abs(g - g_ref).max() <= 0.05 * np.abs(g_ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("metric,eps,alpha", [("l_inf", 0.08, 0.02), ("l_2", 3.0, 1.0)])
def test_gpu_bim_attack_vs_oracle(cloud, gpu, metric, eps, alpha):
    """Three BIM updates, the oracle teacher-forced with the GPU's colours after every step."""
    import torch
    xyz, rgb, labels, pyr = cloud
    model, ws, orc = gpu
    feats0 = np.concatenate([xyz, rgb], 1)
    lab = dev(labels.astype(np.int32))
    cur = rgb.copy()
    # l_inf steps are sign steps: the oracle can be teacher-forced with the GPU's colours after every step.  An l_2 step
    # is proportional to the gradient VALUE and the GPU re-runs the earlier steps with float atomics, so only its first
    # step (which starts from the original colours on both sides) is compared entry by entry; later ones by invariants.
    for it in range(1, 4):
        adv = ws.bim_attack(model, dev(feats0), lab, eps, alpha, it, metric=metric).cpu().numpy()
        torch.cuda.synchronize()
        assert np.array_equal(adv[:, :3], xyz)
        if metric == "l_inf" or it == 1:
            _, _, g = randla_net.loss_and_grad(orc, xyz, cur, labels, pyr)
            want = randla_net.bim_step(rgb, cur, g, eps, alpha, metric)
            same = np.abs(adv[:, 3:6] - want) <= (1e-5 if metric == "l_inf" else 1e-4)
            assert same.mean() >= 0.995, (it, same.mean())
        cur = adv[:, 3:6].copy()
    if metric == "l_inf":
        assert np.abs(cur - rgb).max() <= eps + 1e-6
    else:
        assert np.linalg.norm((cur - rgb).astype(np.float64)) <= eps * (1 + 1e-5)
    assert cur.min() >= 0.0 and cur.max() <= 1.0


@pytest.mark.gpu
def test_gpu_full_size_cloud_and_bim_class():
    """The reference's size (40 960 points): forward and colour gradient against the restatement, then the BIM class
    (the reference's constructor / config / batch_attack call shape): colours stay in the eps ball and the loss grows."""
    import torch
    from pointsecguard_amd.randla import attack, network
    n = 40960
    rng = np.random.default_rng(77)
    xyz = (rng.random((1, n, 3), dtype=np.float32) * np.array([8, 6, 3], np.float32)).astype(np.float32)
    rgb = rng.random((n, 3), dtype=np.float32)
    labels = rng.integers(0, 13, n)
    params = randla_params(5)
    model, ws, orc = network.RandLAModel(params), network.RandLAWorkspace(n), randla_net.RandLAOracle(params)
    ws.set_cloud(dev(xyz[0]))
    pyr = ([xyz[0][:m] for m in (n, n // 4, n // 16, n // 64, n // 256)],
           [ws.index(0, l).cpu().numpy() for l in range(5)],
           [ws.index(0, l).cpu().numpy()[:m] for l, m in enumerate((n // 4, n // 16, n // 64, n // 256, n // 512))],
           [ws.index(1, l).cpu().numpy()[:, None] for l in range(5)])     # indices: pinned separately (test_randla_knn.py)
    feats = dev(np.concatenate([xyz[0], rgb], 1))
    logits = ws.forward(model, feats)
    loss_ref, logits_ref, g_ref = randla_net.loss_and_grad(orc, xyz[0], rgb, labels, pyr)
    assert np.abs(logits.cpu().numpy() - logits_ref).max() <= 2e-4 * max(1.0, np.abs(logits_ref).max())
    loss0, dl = network.colper_grad(logits, dev(labels.astype(np.int32)))
    assert abs(loss0.item() - loss_ref) <= 1e-4 * abs(loss_ref)
    g = ws.backward(model, dl).cpu().numpy()[:, 3:6]
    assert (np.abs(g - g_ref) <= 1e-3 * np.abs(g_ref).max()).mean() >= 0.995
    atk = attack.BIM(model, 1, "colper", "ut", "l_inf")
    atk.config(magnitude=0.05, alpha=0.01, iteration=5)
    adv = atk.batch_attack(feats, labels)
    torch.cuda.synchronize()
    assert tuple(adv.shape) == (n, 3) and float((adv - dev(rgb)).abs().max()) <= 0.05 + 1e-6
    ws2 = network.RandLAWorkspace(n)
    ws2.set_cloud(dev(xyz[0]))
    loss1, _ = network.colper_grad(ws2.forward(model, torch.cat([dev(xyz[0]), adv], 1)), dev(labels.astype(np.int32)))
    assert loss1.item() > loss0.item()


@pytest.mark.gpu
def test_gpu_cloud_batch_equals_single_clouds(cloud):
    """A workspace of 3 clouds (psg_rla_ws_create_batch: one launch of every kernel serves all of them) against three
    one-cloud workspaces: the index pyramids are the per-cloud ones shifted by the cloud's row offset, logits agree to
    fp32 rounding (row-wise kernels, no cross-row arithmetic; the GEMM tile shape, hence the summation order, is chosen
    by the row count), colour gradients to float-atomic order, the l_inf attack (sign steps) ends within the same
    eps ball with >= 98 % of the entries identical, and the l_2 attack (round 4: norms per cloud) moves every cloud as it
    moves alone, bit-reproducibly."""
    import torch
    from pointsecguard_amd.randla import network
    xyz, rgb, labels, _ = cloud
    rng = np.random.default_rng(9)
    B = 3
    xyzs = [xyz] + [np.ascontiguousarray(xyz[rng.permutation(N)] * s).astype(np.float32) for s in (0.7, 1.3)]
    rgbs = [rgb] + [rng.random((N, 3), dtype=np.float32) for _ in range(B - 1)]
    labs = [labels] + [rng.integers(0, 13, N) for _ in range(B - 1)]
    model = network.RandLAModel(randla_params(3))
    wsb, ws1 = network.RandLAWorkspace(N, batch=B), network.RandLAWorkspace(N)
    wsb.set_cloud(dev(np.concatenate(xyzs, 0)))
    feats_b = dev(np.concatenate([np.concatenate([x, c], 1) for x, c in zip(xyzs, rgbs)], 0))
    lab_b = dev(np.concatenate(labs).astype(np.int32))
    logits_b = wsb.forward(model, feats_b)
    _, dl_b = network.colper_grad(logits_b, lab_b)
    g_b = wsb.backward(model, dl_b).cpu().numpy()
    adv_b = wsb.bim_attack(model, feats_b, lab_b, 0.08, 0.02, 3).cpu().numpy()
    # l_2 (what the reference's tester selects, tester_S3DIS.py:37): every cloud is normalised by ITS OWN gradient / delta norm
    eps2, alpha2 = 3.0, 1.2
    adv2_b = wsb.bim_attack(model, feats_b, lab_b, eps2, alpha2, 3, metric="l_2").cpu().numpy()
    adv2_again = wsb.bim_attack(model, feats_b, lab_b, eps2, alpha2, 3, metric="l_2").cpu().numpy()
    assert np.array_equal(adv2_b.view(np.uint32), adv2_again.view(np.uint32))          # fixed-order norms: bit-reproducible
    idx_b = [[wsb.index(w, l).cpu().numpy() for l in range(5)] for w in (0, 1)]
    # no cross-talk: new colours in the LAST cloud leave the other clouds' logits bit-identical (same launch shapes)
    feats_c = feats_b.clone()
    feats_c[(B - 1) * N:, 3:] = dev(rng.random((N, 3), dtype=np.float32))
    logits_c = wsb.forward(model, feats_c)
    assert torch.equal(logits_c[:(B - 1) * N], logits_b[:(B - 1) * N]) and not torch.equal(logits_c[(B - 1) * N:], logits_b[(B - 1) * N:])
    torch.cuda.synchronize()
    ratios = (1, 4, 16, 64, 256)
    for b in range(B):
        ws1.set_cloud(dev(xyzs[b]))
        for l in range(5):
            n_l, n_s = N // ratios[l], N // ratios[l] // (4 if l < 4 else 2)
            assert np.array_equal(idx_b[0][l][b * n_l:(b + 1) * n_l], ws1.index(0, l).cpu().numpy() + b * n_l)
            assert np.array_equal(idx_b[1][l][b * n_l:(b + 1) * n_l], ws1.index(1, l).cpu().numpy() + b * n_s)
        f1 = dev(np.concatenate([xyzs[b], rgbs[b]], 1))
        y1 = dev(labs[b].astype(np.int32))
        logits1 = ws1.forward(model, f1)
        assert float((logits1 - logits_b[b * N:(b + 1) * N]).abs().max()) <= 1e-5 * max(1.0, float(logits1.abs().max()))
        _, dl1 = network.colper_grad(logits1, y1)          # (a sum over points, not a mean: no factor between the two)
        g1 = ws1.backward(model, dl1).cpu().numpy()
        scale = np.abs(g1).max()
        # (a 1e-6 difference in an activation can flip a max-pool argmax or a leaky-relu side: few entries, bounded)
        diff = np.abs(g_b[b * N:(b + 1) * N] - g1)
        assert (diff <= 2e-4 * scale).mean() >= 0.995 and diff.max() <= 0.05 * scale
        adv1 = ws1.bim_attack(model, f1, y1, 0.08, 0.02, 3).cpu().numpy()
        part = adv_b[b * N:(b + 1) * N]
        assert np.array_equal(part[:, :3], xyzs[b]) and np.abs(part[:, 3:] - rgbs[b]).max() <= 0.08 + 1e-6
        assert (part == adv1).mean() >= 0.98           # (sign flips of near-zero gradient entries, compounding over 3 steps)
        # l_2: the cloud inside the batch moves like the cloud alone (its own norms), inside its own eps ball
        adv2_1 = ws1.bim_attack(model, f1, y1, eps2, alpha2, 3, metric="l_2").cpu().numpy()
        part2 = adv2_b[b * N:(b + 1) * N]
        d_b, d_1 = part2[:, 3:] - rgbs[b], adv2_1[:, 3:] - rgbs[b]
        assert np.array_equal(part2[:, :3], xyzs[b]) and np.linalg.norm(d_b.astype(np.float64)) <= eps2 * (1 + 1e-5)
        assert np.linalg.norm(d_b.astype(np.float64)) >= 0.5 * min(eps2, alpha2)        # it did move
        assert np.linalg.norm((d_b - d_1).astype(np.float64)) <= 0.02 * np.linalg.norm(d_1.astype(np.float64))
    # the BIM class with the reference's batch_size argument drives the same cloud-batch workspace
    from pointsecguard_amd.randla import attack
    atk = attack.BIM(model, B, "colper", "ut", "l_inf")
    atk.config(magnitude=0.08, alpha=0.02, iteration=2)   # bim.py:204-232: one update before the loop -> 3 updates like adv_b
    rgb_b = atk.batch_attack(feats_b.reshape(B, N, 6), lab_b.reshape(B, N))
    assert tuple(rgb_b.shape) == (B, N, 3) and (rgb_b.reshape(-1, 3).cpu().numpy() == adv_b[:, 3:]).mean() >= 0.98
    atk2 = attack.BIM(model, B, "colper", "ut", "l_2")           # the tester's metric in the cloud-batch workspace
    atk2.config(magnitude=eps2, alpha=alpha2, iteration=2)
    rgb2_b = atk2.batch_attack(feats_b.reshape(B, N, 6), lab_b.reshape(B, N))
    assert np.array_equal(rgb2_b.reshape(-1, 3).cpu().numpy().view(np.uint32), adv2_b[:, 3:].copy().view(np.uint32))


@pytest.mark.gpu
def test_gpu_gradient_is_bit_reproducible(cloud, gpu):
    """The backward pass gathers through inverse lists built once per cloud (no float atomics): two runs give the same
    bits, and so do two 3-step attacks."""
    import torch
    from pointsecguard_amd.randla import network
    xyz, rgb, labels, _ = cloud
    model, ws, _ = gpu
    ws.set_cloud(dev(xyz))
    feats, lab = dev(np.concatenate([xyz, rgb], 1)), dev(labels.astype(np.int32))
    grads = []
    for _ in range(2):
        _, dl = network.colper_grad(ws.forward(model, feats), lab)
        grads.append(ws.backward(model, dl).clone())
    assert torch.equal(grads[0], grads[1])
    advs = [ws.bim_attack(model, feats, lab, 0.08, 0.02, 3).clone() for _ in range(2)]
    assert torch.equal(advs[0], advs[1])


# ---------------------------------------------------------------------------------------------------------------
# The other attacks of the reference's tester (tester_S3DIS.py:36-44): TBIM / tar_NBattack, NUattack, tar_NUattack, against
# the source-read restatements of oracle/randla_net.py (PARITY UNPINNED like the network itself).

def test_masked_hinge_and_nu_rules_cpu():
    """The restated loss / update rules on hand-made numbers (no GPU)."""
    import torch
    z = torch.tensor([[0.5, 2.0, -1.0], [3.0, 0.1, 0.2], [-1.0, -2.0, -3.0]])
    ys = torch.tensor([0, 0, 2])
    # point 0: other = max(0*, 2, -1) = 2 -> 1.5; point 1: real 3, other max(0*, .1, .2) = .2 -> 0; point 2: the masked
    # true logit counts as 0 in the max (bim.py:113): other = max(-1, -2, 0*) = 0, real = -3 -> 3
    assert abs(float(randla_net.colper_loss_masked(z, ys)) - 4.5) < 1e-6
    assert abs(float(randla_net.colper_loss_masked(z, ys, torch.tensor([1.0, 1.0, 0.0]))) - 1.5) < 1e-6
    xs = np.array([[0.2, 0.5, 0.9]], np.float32)
    adv, _ = randla_net.nu_color(xs, np.zeros_like(xs))
    assert np.abs(adv - (xs * (1 - 1e-6) + 0.5e-6)).max() < 1e-6        # tanh(atanh(t)) = t: the bound scaling only
    adv_m, _ = randla_net.nu_color(np.repeat(xs, 2, 0), np.ones((2, 3), np.float32), mask=np.array([True, False]))
    assert np.array_equal(adv_m[1], xs[0]) and (adv_m[0] > xs[0]).all()
    dws, m, v, dist = randla_net.nu_adam_step(xs, np.zeros_like(xs), np.zeros_like(xs), np.zeros_like(xs), 1,
                                              np.array([[1.0, -1.0, 0.0]], np.float32), 2.0, 0.01)
    # first Adam step: |update| = lr * m_hat / (sqrt(v_hat) + eps') = lr for every entry with a gradient
    assert np.allclose(np.abs(dws[0, :2]), 0.01, rtol=1e-3) and dws[0, 0] < 0 < dws[0, 1]
    assert randla_net.mean_iou([0, 0, 1, 1], [0, 1, 1, 1]) == pytest.approx((1 / 2 + 2 / 3) / 2)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,eps,alpha", [("l_inf", 0.08, 0.02), ("l_2", 3.0, 1.0)])
def test_gpu_tbim_steps_vs_oracle(cloud, gpu, metric, eps, alpha):
    """TBIM: masked hinge against the target labels, negated gradient (descent), BIM's update; teacher-forced like the BIM
    test.  Then the class: the reference's tuple, the `sr > 0.9` stop and an honest success rate."""
    import torch
    from pointsecguard_amd import _lib, runtime
    from pointsecguard_amd.randla import attack
    xyz, rgb, labels, pyr = cloud
    model, ws, orc = gpu
    ws.set_cloud(dev(xyz))
    ori, target = 2, 7
    mask_h = labels == ori
    ys_target = np.where(mask_h, target, labels)
    feat = dev(np.concatenate([xyz, rgb], 1))
    ys, mask = dev(ys_target.astype(np.int32)), dev(mask_h.astype(np.uint8))
    ori_rgb = dev(rgb)
    norms, delta = torch.zeros(2, device="cuda"), torch.empty(len(rgb), 3, device="cuda")
    cur = rgb.copy()
    for it in range(3):
        logits = ws.forward(model, feat)
        dl = torch.empty_like(logits)
        loss = torch.zeros(1, device="cuda")
        _lib.call("psg_rla_colper_grad_masked", runtime.ptr(logits), runtime.ptr(ys), runtime.ptr(mask), -1.0, len(rgb), runtime.ptr(dl),
                  runtime.ptr(loss), runtime.stream())
        g = ws.backward(model, dl)
        score_ref, _, g_ref = randla_net.score_and_grad(orc, xyz, cur, ys_target, pyr, mask_h)
        assert abs(loss.item() - score_ref) <= 1e-4 * max(1.0, abs(score_ref))
        got = g.cpu().numpy()[:, 3:6]
        assert (np.abs(got + g_ref) <= 1e-3 * np.abs(g_ref).max()).mean() >= 0.995         # sign -1: the negated gradient
        _lib.call("psg_rla_bim_step", runtime.ptr(feat), runtime.ptr(g), runtime.ptr(ori_rgb), len(rgb), eps, alpha,
                  1 if metric == "l_2" else 0, runtime.ptr(norms), runtime.ptr(delta), runtime.stream())
        adv = feat.cpu().numpy()
        assert np.array_equal(adv[:, :3], xyz)
        if metric == "l_inf" or it == 0:
            want = randla_net.bim_step(rgb, cur, -g_ref, eps, alpha, metric)
            assert (np.abs(adv[:, 3:6] - want) <= (1e-5 if metric == "l_inf" else 1e-4)).mean() >= 0.995, it
        cur = adv[:, 3:6].copy()
    atk = attack.tar_NBattack(model, 1, "colper", "t", metric)
    atk.config(magnitude=eps, alpha=alpha, iteration=4, rand_init_magnitude=0.5)
    points, sr, other_acc, ori_other_acc, dist, other_miou, ori_other_miou = atk.batch_attack(np.concatenate([xyz, rgb], 1), labels,
                                                                                            target=target, ori=ori)
    assert points == mask_h.sum() and 0.0 <= sr <= 1.0 and 0.0 <= other_acc <= 1.0 and 0.0 < other_miou <= 1.0
    adv = atk.last_adv.cpu().numpy()
    assert adv.min() >= 0.0 and adv.max() <= 1.0
    if metric == "l_inf":
        assert np.abs(adv - rgb).max() <= eps + 1e-6
    else:
        assert dist <= eps * (1 + 1e-5)
    # the masked score (what the attack minimises) did not grow
    s0, _, _ = randla_net.score_and_grad(orc, xyz, rgb, ys_target, pyr, mask_h)
    s1, _, _ = randla_net.score_and_grad(orc, xyz, adv, ys_target, pyr, mask_h)
    assert s1 <= s0 * (1 + 1e-6)
    with pytest.raises(ValueError):
        atk.batch_attack(np.concatenate([xyz, rgb], 1), np.zeros_like(labels), target=target, ori=ori)


@pytest.mark.gpu
@pytest.mark.parametrize("masked", [False, True])
def test_gpu_nu_steps_vs_oracle(cloud, gpu, masked):
    """NUattack / tar_NUattack: three Adam steps in tanh space, the oracle teacher-forced with the GPU's optimiser state."""
    import torch
    from pointsecguard_amd import _lib, runtime
    from pointsecguard_amd.randla import attack
    xyz, rgb, labels, pyr = cloud
    model, ws, orc = gpu
    ws.set_cloud(dev(xyz))
    n = len(rgb)
    ori, target, c, lr = 3, 9, 0.5, 0.01
    mask_h = (labels == ori) if masked else None
    ys_h = np.where(labels == ori, target, labels) if masked else labels
    feat = dev(np.concatenate([xyz, rgb], 1))
    xs, ys = dev(rgb), dev(ys_h.astype(np.int32))
    mask = dev(mask_h.astype(np.uint8)) if masked else None
    mptr = runtime.ptr(mask) if masked else None
    # start away from d_ws = 0: there the distance is the 1e-6 bound scaling alone and its gradient (adv - x) / |adv - x| a unit
    # vector of tanh / atanh rounding noise (the reference's first step has that property too; nothing to compare)
    dws0 = (0.05 * np.random.default_rng(8).standard_normal((n, 3))).astype(np.float32)
    dws = dev(dws0)
    m, v = (torch.zeros(n, 3, device="cuda") for _ in range(2))
    dist2 = torch.zeros(1, device="cuda")
    for t in range(1, 4):
        h_dws, h_m, h_v = dws.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()
        _lib.call("psg_rla_nu_color", runtime.ptr(xs), runtime.ptr(dws), mptr, n, runtime.ptr(feat), runtime.ptr(dist2), runtime.stream())
        adv_ref, _ = randla_net.nu_color(rgb, h_dws, mask_h)
        assert np.abs(feat.cpu().numpy()[:, 3:6] - adv_ref).max() <= 2e-6
        logits = ws.forward(model, feat)
        dl = torch.empty_like(logits)
        _lib.call("psg_rla_colper_grad_masked", runtime.ptr(logits), runtime.ptr(ys), mptr, 1.0, n, runtime.ptr(dl), None, runtime.stream())
        dfeat = ws.backward(model, dl)
        _lib.call("psg_rla_nu_adam_step", runtime.ptr(xs), runtime.ptr(dws), runtime.ptr(m), runtime.ptr(v), mptr, runtime.ptr(feat),
                  runtime.ptr(dfeat), runtime.ptr(dist2), n, c, lr, t, runtime.stream())
        _, _, g_ref = randla_net.score_and_grad(orc, xyz, adv_ref, ys_h, pyr, mask_h)
        w_dws, w_m, w_v, dist = randla_net.nu_adam_step(rgb, h_dws, h_m, h_v, t, g_ref, c, lr, mask_h)
        # (at t = 1 the distance is the 1e-6 bound scaling alone, 5e-5 in all: tanh / atanh rounding shows at the 1e-7 level)
        assert abs(float(dist2.item()) ** 0.5 - dist) <= max(1e-4 * dist, 5e-7)
        # Adam divides by sqrt(v): entries with a near-zero gradient amplify rounding; compare where the oracle's gradient is
        # not negligible, and the moments everywhere
        assert (np.abs(m.cpu().numpy() - w_m) <= 1e-3 * np.abs(w_m).max() + 1e-12).mean() >= 0.995
        big = np.abs(w_m) > 1e-2 * np.abs(w_m).max()
        assert (np.abs(dws.cpu().numpy() - w_dws)[big] <= 2e-4).mean() >= 0.99, t
        if masked:
            assert np.array_equal(dws.cpu().numpy()[~mask_h], dws0[~mask_h])            # the other points' variable never moves
    cls = attack.tar_NUattack(model) if masked else attack.NUattack(model, 1, "ut", "l_2")
    cls.config(cs=c, iteration=3)
    np.random.seed(0)
    out = cls.batch_attack(np.concatenate([xyz, rgb], 1), labels, target=target, ori=ori) if masked else \
        cls.batch_attack(np.concatenate([xyz, rgb], 1), labels)
    assert len(out) == 7 and all(np.isfinite(out))
    adv = cls.last_adv.cpu().numpy()
    assert adv.min() >= 0.0 and adv.max() <= 1.0
    if masked:
        assert np.abs(adv[~mask_h] - rgb[~mask_h]).max() == 0.0 and np.abs(adv[mask_h] - rgb[mask_h]).max() > 0.0
