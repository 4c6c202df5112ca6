"""GPU parity tests proper: the HIP path (through the C ABI) against the reference-generated golden
fixtures and against the CPU oracle on the same seeded inputs.

Bars (SURVEY.md section 7 "Stating the fp tolerance"):
  * grouping indices (FPS, ball query, 3-NN) and interpolation weights: BIT-EXACT;
  * log-probs / activations: abs <= 1e-4;
  * colour gradient: zero pattern identical, sign agreement >= 99.9 % of non-zero entries,
    every disagreement below 1e-3 of the gradient's max magnitude;
  * adversarial colours: bit-equal wherever gradient signs agree (>= 99.9 % of entries).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGP_TOL = 1e-4


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda().contiguous()


@pytest.fixture(scope="module")
def room_run(gpu_model, golden_room):
    """Plan + forward + backward of the golden room on the GPU (one workspace for the module)."""
    from pointsecguard_amd import runtime
    g = golden_room
    ws = runtime.PN2Workspace(1, 4096, 2)
    x0 = dev(g["room"][None])
    starts = dev(g["starts"].reshape(1, 4, 1), torch.int32)
    ws.plan_build(x0, starts, 1)
    logp = ws.forward(gpu_model, 0, x0)
    torch.cuda.synchronize()
    return ws, x0, logp


def test_geometry_bit_exact_vs_reference(room_run, golden_room):
    ws, _, _ = room_run
    g = golden_room
    for lvl in range(4):
        assert np.array_equal(ws.plan_tensor(0, lvl, 0, 0).cpu().numpy(), g["fps%d" % lvl].astype(np.int32)), lvl
        assert np.array_equal(ws.plan_tensor(1, lvl, 0, 0).cpu().numpy(), g["group%d" % lvl].astype(np.int32)), lvl
        assert np.array_equal(ws.plan_tensor(2, lvl, 0, 0).cpu().numpy(), g["nn_idx%d" % lvl].astype(np.int32)), lvl
        w = ws.plan_tensor(3, lvl, 0, 0).cpu().numpy()
        assert np.array_equal(w.view(np.uint32), g["nn_w%d" % lvl].view(np.uint32)), lvl


def test_forward_vs_reference(room_run, golden_room):
    ws, _, logp = room_run
    g = golden_room
    for which, name in enumerate(("sa1", "sa2", "sa3", "sa4", "fp4", "fp3", "fp2")):
        a = ws.activation(which)[0].cpu().numpy()
        assert np.abs(a - g["act_" + name]).max() <= LOGP_TOL, name
    assert np.abs(logp[0].cpu().numpy() - g["logp"]).max() <= LOGP_TOL


def check_grad(ours, ref):
    nz = ref != 0
    assert np.array_equal(ours != 0, nz), "zero pattern of the colour gradient differs"
    agree = np.sign(ours[nz]) == np.sign(ref[nz])
    assert agree.mean() >= 0.999
    if not agree.all():
        assert np.abs(ref[nz][~agree]).max() <= 1e-3 * np.abs(ref).max()
    rel = np.abs(ours - ref)[nz] / np.abs(ref[nz])
    assert np.median(rel) < 1e-4


def test_backward_vs_reference(room_run, golden_room, gpu_model):
    from pointsecguard_amd import _lib, runtime
    ws, x0, logp = room_run
    g = golden_room
    labels = dev(g["labels"].astype(np.int32)[None])
    dlogp = torch.empty_like(logp)
    cost = torch.zeros(1, device="cuda")
    _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(labels), 0, 4096, 4096, 13, 1.0 / 4096,
              runtime.ptr(dlogp), runtime.ptr(cost), runtime.stream())
    dx0 = ws.backward(gpu_model, 0, dlogp)
    torch.cuda.synchronize()
    assert abs(cost.item() - float(g["cost"])) < 1e-4
    check_grad(dx0[0, :, 3:6].cpu().numpy(), g["dcolor"])


def _set_color(x0, color_cn):
    """x0 [B,N,9] point-major device tensor <- colour state [B,3,N] (numpy, reference layout)."""
    x0[:, :, 3:6] = dev(np.ascontiguousarray(color_cn.transpose(0, 2, 1)))


PATHS = ("pieces", "timed")


def _one_step(ws, model, x0, ori, labels, slot, alpha, eps, last, mask=None, target=None, path="pieces"):
    """One attack iteration assembled from the C-ABI pieces.  path = "pieces": the general entry points (full forward with
    the module outputs, backward on all nine input channels with the first SA layer's transpose on the matrix pipe,
    psg_pgd_step as a launch of its own).  path = "timed": the launches psg_pn2_nb_attack itself runs per iteration and the
    benchmark therefore times (round 6) - the lean forward, the colour-only backward (level 0's first layer on the vector
    pipe: another summation order) with the PGD step applied by the gradient's last gather; the gradient for check_flips
    comes from the same colour-only backward run once more without the fused step (bit-reproducible: no float atomics)."""
    from pointsecguard_amd import _lib, runtime
    B, N = x0.shape[0], x0.shape[1]
    timed = path == "timed"
    logp = ws.forward(model, slot, x0, lean=timed)
    dlogp = torch.empty_like(logp)
    if target is None:
        _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(labels), 0, B * N, B * N, 13, 1.0 / N,
                  runtime.ptr(dlogp), None, runtime.stream())
    else:
        _lib.call("psg_ce_logp_grad", runtime.ptr(logp), None, int(target), B * N, N, 13, 1.0 / N,
                  runtime.ptr(dlogp), None, runtime.stream())
    if timed:
        dx0 = ws.backward(model, slot, dlogp, colour_only=True)
        ws.backward_pgd(model, slot, dlogp, x0, ori, alpha, eps, mask=mask, descent=target is not None, last=last)
    else:
        dx0 = ws.backward(model, slot, dlogp)
        _lib.call("psg_pgd_step", runtime.ptr(x0), runtime.ptr(dx0), runtime.ptr(ori), runtime.ptr(mask), B, N,
                  float(alpha), float(eps), -1.0 if target is not None else 1.0, 1 if last else 0, runtime.stream())
    torch.cuda.synchronize()
    _one_step.last_grad = dx0[:, :, 3:6].cpu().numpy().transpose(0, 2, 1)      # [B,3,N], for check_flips
    return x0[:, :, 3:6].cpu().numpy().transpose(0, 2, 1)


def check_flips(got, nxt, grad, bar=0.999):
    """Colours after one teacher-forced step: bit-equal on >= `bar` of the entries, and every entry that differs is a
    gradient whose sign is not determined at fp32: |g| below 3e-3 of the largest gradient magnitude (measured: the
    largest flipped |g| over the tested iterations is 1.6e-3 max|g|, i.e. 1.1e-5 against 6.8e-3; an entry moved by
    +-alpha where the reference moved it by -+alpha or not at all)."""
    diff = np.ascontiguousarray(got).view(np.uint32) != np.ascontiguousarray(nxt).view(np.uint32)
    same = 1.0 - diff.mean()
    assert same >= bar, same
    if diff.any():
        assert np.abs(grad[diff]).max() <= 3e-3 * np.abs(grad).max(), (np.abs(grad[diff]).max(), np.abs(grad).max())
    return same


@pytest.mark.parametrize("path", PATHS)
def test_nb_attack_steps_vs_reference(gpu_model, golden_nb, path):
    """Teacher-forced NB_attack iterations: start from the colour state the REFERENCE fed to its model
    at iteration t, run one HIP iteration, compare with the reference's state at t+1.  (A free-running
    40-iteration comparison cannot be bit-stable between ANY two fp32 implementations: one flipped
    sign of a ~1e-9 gradient is a 2*alpha colour change that the next iterations amplify; DESIGN.md.)"""
    from pointsecguard_amd import runtime
    g = golden_nb
    rooms, iters = g["rooms"], int(g["iters"])
    B = rooms.shape[0]
    labels = dev(g["labels"].astype(np.int32))
    x0 = dev(rooms)
    ori = x0[:, :, 3:6].contiguous()
    ws = runtime.PN2Workspace(B, 4096, iters)
    # golden RNG stream = [clean forward] + attack forwards + [adversarial forward]: attack draws start at 1
    ws.plan_build(x0, dev(g["starts"][1:1 + iters], torch.int32), iters)
    checked = 0
    for t in (0, 1, 2, 5, 10, 20, 39):
        nxt = g["adv_color_final"] if t == iters - 1 else g["state_it%d" % (t + 1)]
        _set_color(x0, g["state_it%d" % t])
        got = _one_step(ws, gpu_model, x0, ori, labels, t, g["alpha"], g["eps"], last=(t == iters - 1), path=path)
        check_flips(got, nxt, _one_step.last_grad)
        checked += 1
    assert checked == 7


def _project(adv_color, ori_color, eps):
    """nontarget.py:38-39 on the host, fp32 like torch: the projection the reference applies to the colours it feeds to its
    next iteration (NB_attack returns the UN-projected last step)."""
    eta = np.clip((adv_color - ori_color).astype(np.float32), np.float32(-eps), np.float32(eps))
    return np.clip((ori_color + eta).astype(np.float32), np.float32(0), np.float32(1))


@pytest.mark.parametrize("fixture", ["pn2_nb.npz", "pn2_nb_b8.npz"])
def test_fused_nb_attack_first_iterations_vs_reference(gpu_model, fixture):
    """psg_pn2_nb_attack ITSELF (the call the benchmark times: plan + lean forwards + colour-only backwards + fused PGD
    steps in one C call) for k = 1 and k = 3 (B = 8 fixture: 5) iterations from the reference's state at iteration 0 (= the clean colours:
    nontarget.py:20-23 starts from the images) against the reference's recorded state entering iteration k.  The call
    returns the un-projected last step; the host projects it.  k = 1: the check_flips bar of the teacher-forced tests
    (>= 99.9 % bit-equal, flips only where |g| < 3e-3 max|g|).  k = 3 runs free for three iterations: an entry flipped in
    iteration 0 or 1 is a 2 alpha colour change that moves the neighbouring gradients, so the bar is the bit-equal share
    alone (measured: 0.99870 after three iterations at B = 2, 0.99996 after five at B = 8 - the path has no float atomics, so
    these are the same bits on every box; every differing entry within 2 alpha k of the reference's)."""
    import os
    from conftest import GOLDEN
    from pointsecguard_amd import _lib, runtime
    g = dict(np.load(os.path.join(GOLDEN, fixture)))
    if fixture == "pn2_nb_b8.npz":
        rooms, lab = _b8_batch(g, 0)
        starts_all, free = g["b0_starts"][1:], (5, g["state_it5"])   # (the B = 8 fixture records iterations 0, 1, 5, 39)
    else:
        rooms, lab, starts_all, free = g["rooms"], g["labels"], g["starts"][1:], (3, g["state_it3"])
    B = rooms.shape[0]
    eps, alpha = float(g["eps"]), float(g["alpha"])
    images_np = np.ascontiguousarray(rooms.transpose(0, 2, 1))
    assert np.array_equal(g["state_it0"], images_np[:, 3:6])
    images, labels = dev(images_np), dev(lab.astype(np.int32))
    for k, want in ((1, g["state_it1"]), free):
        ws = runtime.PN2Workspace(B, 4096, k)
        adv = ws.nb_attack(gpu_model, images, labels, dev(starts_all[:k], torch.int32), eps, alpha, k)
        torch.cuda.synchronize()
        out = adv.cpu().numpy()
        assert np.array_equal(out[:, :3], images_np[:, :3]) and np.array_equal(out[:, 6:], images_np[:, 6:])
        got = _project(out[:, 3:6], images_np[:, 3:6], eps)
        if k == 1:
            # the gradient at iteration 0 for the flip criterion: the same colour-only backward, outside the fused call
            x0 = dev(rooms)
            ws.plan_build(x0, dev(starts_all[:1], torch.int32), 1)
            logp = ws.forward(gpu_model, 0, x0, lean=True)
            dlogp = torch.empty_like(logp)
            _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(labels), 0, B * 4096, B * 4096, 13, 1.0 / 4096,
                      runtime.ptr(dlogp), None, runtime.stream())
            grad = ws.backward(gpu_model, 0, dlogp, colour_only=True)[:, :, 3:6].cpu().numpy().transpose(0, 2, 1)
            check_flips(got, want, grad)
        else:
            diff = got.view(np.uint32) != want.view(np.uint32)
            print("fused k=%d: bit-equal share %.5f" % (k, 1.0 - diff.mean()))
            assert 1.0 - diff.mean() >= (0.998 if k == 3 else 0.995), 1.0 - diff.mean()
            assert np.abs(got - want).max() <= 2 * alpha * k + 1e-6


def test_nb_attack_free_run_vs_reference(gpu_model, golden_nb):
    """Fused 40-iteration attack: invariants + statistical parity with the reference's result."""
    from pointsecguard_amd import runtime
    g = golden_nb
    rooms, iters = g["rooms"], int(g["iters"])
    B = rooms.shape[0]
    images_np = np.ascontiguousarray(rooms.transpose(0, 2, 1))
    images, labels = dev(images_np), dev(g["labels"].astype(np.int32))
    ws = runtime.PN2Workspace(B, 4096, iters)
    adv = ws.nb_attack(gpu_model, images, labels, dev(g["starts"][1:1 + iters], torch.int32), float(g["eps"]),
                       float(g["alpha"]), iters)
    torch.cuda.synchronize()
    out = adv.cpu().numpy()
    assert np.array_equal(out[:, :3], images_np[:, :3]) and np.array_equal(out[:, 6:], images_np[:, 6:])
    # un-projected last step: within eps + alpha of the original, may leave [0,1] by alpha
    assert np.abs(out[:, 3:6] - images_np[:, 3:6]).max() <= float(g["eps"]) + float(g["alpha"]) + 1e-6
    ref = g["adv_color_final"]
    same = (out[:, 3:6].view(np.uint32) == ref.view(np.uint32)).mean()
    # Free-running bit equality after 40 iterations is a chaotic quantity, calibrated by the reference against ITSELF
    # (tests/golden/pn2_nb_threads.json: the same attack with 2 or 8 intra-op threads instead of 1 stays bit-identical
    # for the first 20+ iterations and ends at 0.968): this implementation's backward sums in another order than MKL's,
    # flips a handful of |g| < 1e-3 max|g| signs in EVERY iteration (test above), each one a 2 alpha colour change
    # that doubles per iteration, and ends lower: 0.78 measured.  Not a parity bar (that is the teacher-forced test and
    # the statistics below), a tripwire against gross divergence.
    assert same >= 0.65, same
    # adversarial accuracy / mIoU parity on the same RNG slot as the reference's adversarial forward
    ev = runtime.PN2Workspace(B, 4096, 1)
    x0 = adv.transpose(1, 2).contiguous()
    ev.plan_build(x0, dev(g["starts"][iters + 1:iters + 2], torch.int32), 1)
    counters, _ = runtime.seg_stats(ev.forward(gpu_model, 0, x0), labels)
    torch.cuda.synchronize()
    c = counters.cpu().numpy().astype(np.float64)
    adv_acc = c[1].sum() / c[0].sum()
    adv_miou = np.mean((c[1] / (c[2] + 1e-6))[c[0] != 0])
    assert abs(adv_acc - float(g["adv_acc"])) <= 0.01, (adv_acc, float(g["adv_acc"]))
    assert abs(adv_miou - float(g["adv_miou"])) <= 0.01, (adv_miou, float(g["adv_miou"]))
    # clean metrics with the reference's clean-forward RNG slot: counters must match EXACTLY
    x0c = dev(rooms)
    ev.plan_build(x0c, dev(g["starts"][0:1], torch.int32), 1)
    counters, _ = runtime.seg_stats(ev.forward(gpu_model, 0, x0c), labels)
    c = counters.cpu().numpy()
    assert np.abs(c[1] - g["inter"]).sum() <= 4 and np.abs(c[2] - g["union"]).sum() <= 8
    assert np.array_equal(c[0], g["seen"])


@pytest.mark.parametrize("path", PATHS)
def test_tar_nb_attack_steps_vs_reference(gpu_model, golden_tarnb, path):
    from pointsecguard_amd import runtime
    g = golden_tarnb
    rooms, iters = g["rooms"], int(g["iters"])
    B = rooms.shape[0]
    x0 = dev(rooms)
    ori = x0[:, :, 3:6].contiguous()
    mask = dev(g["mask"].astype(np.uint8))
    ws = runtime.PN2Workspace(B, 4096, iters)
    ws.plan_build(x0, dev(g["starts"][:iters], torch.int32), iters)
    for t in (0, 1, 5, 9):
        nxt = g["adv_color_final"] if t == iters - 1 else g["state_it%d" % (t + 1)]
        _set_color(x0, g["state_it%d" % t])
        got = _one_step(ws, gpu_model, x0, ori, None, t, g["alpha"], g["eps"], last=(t == iters - 1), mask=mask,
                        target=int(g["target"]), path=path)
        same = (np.ascontiguousarray(got).view(np.uint32) == nxt.view(np.uint32)).mean()
        assert same >= 0.999, (t, same)
        m = g["mask"]
        assert np.array_equal(got[:, :, ~m], rooms.transpose(0, 2, 1)[:, 3:6][:, :, ~m])
    # fused call: only masked colours may move, everything else bit-identical
    images_np = np.ascontiguousarray(rooms.transpose(0, 2, 1))
    adv = ws.nb_attack(gpu_model, dev(images_np), None, dev(g["starts"][:iters], torch.int32), float(g["eps"]),
                       float(g["alpha"]), iters, mask=mask, target=int(g["target"]))
    torch.cuda.synchronize()
    out = adv.cpu().numpy()
    assert np.array_equal(out[:, :, ~g["mask"]], images_np[:, :, ~g["mask"]])
    same = (out[:, 3:6].view(np.uint32) == g["adv_color_final"].view(np.uint32)).mean()
    assert same >= 0.99, same


@pytest.mark.parametrize("n,s", [(4096, 1024), (1024, 256), (256, 64), (64, 16), (1000, 100), (77, 5)])
def test_unit_geometry_vs_oracle(n, s):
    """Ragged / non-power-of-two sizes of the geometry unit ops, bit-exact against the oracle."""
    from oracle import pn2
    from pointsecguard_amd import runtime
    rng = np.random.default_rng(n * 7 + s)
    P = 3
    xyz = rng.random((P, n, 3), dtype=np.float32)
    xyz[..., 2] *= 3
    xyz[0, n // 2] = xyz[0, n // 3]  # a duplicated point (distance ties)
    start = rng.integers(0, n, P).astype(np.int32)
    d_xyz = dev(xyz)
    idx = runtime.fps(d_xyz, s, dev(start))
    new_xyz = runtime.gather_points(d_xyz, idx)
    grp = runtime.ball_query(0.2, 32, d_xyz, new_xyz)
    if s >= 3:
        nn_idx, nn_w = runtime.three_nn(d_xyz, new_xyz)
    torch.cuda.synchronize()
    for p in range(P):
        o_idx = pn2.fps(xyz[p], s, start[p])
        assert np.array_equal(idx[p].cpu().numpy(), o_idx)
        o_new = xyz[p][o_idx]
        assert np.array_equal(new_xyz[p].cpu().numpy(), o_new)
        assert np.array_equal(grp[p].cpu().numpy(), pn2.ball_query(0.2, 32, xyz[p], o_new))
        if s >= 3:
            oi, ow = pn2.three_nn(xyz[p], o_new)
            assert np.array_equal(nn_idx[p].cpu().numpy(), oi)
            assert np.array_equal(nn_w[p].cpu().numpy().view(np.uint32), ow.view(np.uint32))


def test_forward_backward_vs_oracle_batch(gpu_model, oracle_net):
    """B=3 fresh rooms, two plan slots: HIP forward/backward against the oracle on identical inputs."""
    from oracle import pn2
    from pointsecguard_amd import _lib, runtime
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    B = 3
    rooms = make_rooms(B, 99, structured=True)
    labels = rule_labels(rooms)
    rng = np.random.default_rng(5)
    starts = np.stack([rng.integers(0, n, (2, B)) for n in (4096, 1024, 256, 64)], axis=1).astype(np.int32)  # [2,4,B]
    ws = runtime.PN2Workspace(B, 4096, 2)
    x0 = dev(rooms)
    ws.plan_build(x0, dev(starts), 2)
    logp = ws.forward(gpu_model, 1, x0)
    dlogp = torch.empty_like(logp)
    _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(dev(labels.astype(np.int32))), 0, B * 4096, B * 4096,
              13, 1.0 / 4096, runtime.ptr(dlogp), None, runtime.stream())
    dx0 = ws.backward(gpu_model, 1, dlogp)
    torch.cuda.synchronize()
    for b in range(B):
        geom = oracle_net.geometry(rooms[b, :, :3], starts[1, :, b])
        o_logp, cache = oracle_net.forward(rooms[b], geom)
        assert np.abs(logp[b].cpu().numpy() - o_logp).max() <= LOGP_TOL
        o_dlogp, _ = pn2.nll_logp_grad(o_logp, labels[b], 1.0 / 4096)
        check_grad(dx0[b, :, 3:6].cpu().numpy(), oracle_net.backward_color(cache, o_dlogp))


def test_forward_backward_vs_oracle_clustered_room(gpu_model, oracle_net):
    """A room whose density is as uneven as it gets: 3 400 of the 4096 points inside a 2 cm ball, the rest spread over
    the room.  The few coarse points FPS leaves inside the ball are the 3-NN of thousands of fine points (the transposed
    interpolation lists of fp_bwd, dealt to waves by entry count, are then hundreds of entries long against a mean of 12),
    every ball-query group there is full, and distances tie.  Indices must still be bit-exact and the forward / colour
    gradient within the usual bars of the oracle."""
    from oracle import pn2
    from pointsecguard_amd import _lib, runtime
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    rooms = make_rooms(1, 321)
    rng = np.random.default_rng(9)
    sel = rng.permutation(4096)[:3400]
    ball = rng.normal(0, 0.005, (3400, 3)).astype(np.float32)
    rooms[0, sel, 0:3] = np.array([0.1, -0.2, 1.0], np.float32) + ball
    rooms[0, sel, 6] = rooms[0, sel, 0] + 0.5
    rooms[0, sel, 7] = rooms[0, sel, 1] + 0.5
    rooms[0, sel, 8] = rooms[0, sel, 2] / 3.0
    labels = rule_labels(rooms)
    starts = np.stack([rng.integers(0, n, (1, 1)) for n in (4096, 1024, 256, 64)], axis=1).astype(np.int32)   # [1,4,1]
    ws = runtime.PN2Workspace(1, 4096, 1)
    x0 = dev(rooms)
    ws.plan_build(x0, dev(starts), 1)
    logp = ws.forward(gpu_model, 0, x0)
    dlogp = torch.empty_like(logp)
    _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(dev(labels.astype(np.int32))), 0, 4096, 4096, 13, 1.0 / 4096,
              runtime.ptr(dlogp), None, runtime.stream())
    dx0 = ws.backward(gpu_model, 0, dlogp)
    torch.cuda.synchronize()
    geom = oracle_net.geometry(rooms[0, :, :3], starts[0, :, 0])
    for lvl in range(4):
        assert np.array_equal(ws.plan_tensor(0, lvl, 0, 0).cpu().numpy(), geom["fps"][lvl].astype(np.int32)), lvl
        assert np.array_equal(ws.plan_tensor(1, lvl, 0, 0).cpu().numpy(), geom["group"][lvl].astype(np.int32)), lvl
        assert np.array_equal(ws.plan_tensor(2, lvl, 0, 0).cpu().numpy(), geom["nn_idx"][lvl].astype(np.int32)), lvl
    indeg = np.bincount(geom["nn_idx"][0].reshape(-1), minlength=1024)
    assert indeg.max() >= 150, indeg.max()                       # the case this test is about (mean: 12)
    o_logp, cache = oracle_net.forward(rooms[0], geom)
    assert np.abs(logp[0].cpu().numpy() - o_logp).max() <= LOGP_TOL
    o_dlogp, _ = pn2.nll_logp_grad(o_logp, labels[0], 1.0 / 4096)
    check_grad(dx0[0, :, 3:6].cpu().numpy(), oracle_net.backward_color(cache, o_dlogp))


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] at its own batch: tests/golden/pn2_nb_b8.npz = the reference's NB_attack on B = 8 rooms (two
# batches, 16 rooms, make_golden_big.py: gen_nb_b8)

@pytest.fixture(scope="module")
def golden_nb_b8():
    import os
    from conftest import GOLDEN
    return dict(np.load(os.path.join(GOLDEN, "pn2_nb_b8.npz")))


def _b8_batch(g, bi):
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    rooms = make_rooms(8, int(g["seed_rooms"][bi]))
    return rooms, rule_labels(rooms)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("fixture", ["pn2_nb_b8.npz", "pn2_nb_dup.npz"])
def test_nb_b8_steps_vs_reference(gpu_model, fixture, path):
    """Teacher-forced PGD steps at configs[1]'s own batch against the reference's recorded states; the second fixture's rooms
    contain ~900 EXACT duplicates of points each (sampling with replacement, S3DISDataLoader.py:149-154): equal coordinates
    are where the reference's `sort` / `max` are least defined, so the tie rules (lowest index first) are exercised on
    FPS, the ball query and the 3-NN interpolation at once."""
    import os
    from conftest import GOLDEN
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import make_rooms_with_duplicates, rule_labels
    g = dict(np.load(os.path.join(GOLDEN, fixture)))
    iters = int(g["iters"])
    if fixture == "pn2_nb_dup.npz":
        rooms = make_rooms_with_duplicates(8, int(g["seed_room"]))
        lab = rule_labels(rooms)
        n_dup = [4096 - len(np.unique(r.view([("", r.dtype)] * 9))) for r in rooms]
        assert np.array_equal(n_dup, g["duplicates_per_room"]) and min(n_dup) > 500
        starts, kept = g["starts"][0:iters], (0, 1, 39)
    else:
        rooms, lab = _b8_batch(g, 0)
        starts, kept = g["b0_starts"][1:1 + iters], (0, 1, 5, 39)
    labels = dev(lab.astype(np.int32))
    x0 = dev(rooms)
    ori = x0[:, :, 3:6].contiguous()
    ws = runtime.PN2Workspace(8, 4096, iters)
    ws.plan_build(x0, dev(starts, torch.int32), iters)
    assert np.array_equal(g["state_it0"], rooms.transpose(0, 2, 1)[:, 3:6]) and 39 in kept
    dup_point = None
    if fixture == "pn2_nb_dup.npz":
        dup_point = np.zeros((8, 4096), bool)                 # points whose coordinates occur more than once in their room
        for b in range(8):
            _, inv, cnt = np.unique(rooms[b, :, :3], axis=0, return_inverse=True, return_counts=True)
            dup_point[b] = cnt[inv.reshape(-1)] > 1
    for t, nxt in ((0, g["state_it1"]), (39, g["adv_color_final"])):
        _set_color(x0, g["state_it%d" % t])
        got = _one_step(ws, gpu_model, x0, ori, labels, t, g["alpha"], g["eps"], last=(t == iters - 1), path=path)
        if dup_point is None or t == 0:
            # (iteration 0 of the duplicate fixture: twins are EXACT copies, every tie is exact, the tie rules decide - and
            # all 98 304 colours come out bit-equal to the reference's)
            same = check_flips(got, nxt, _one_step.last_grad)
            assert dup_point is None or same == 1.0
        else:
            # Iteration 39: the twins' colours have drifted apart by up to eps, their coordinates are still equal - they sit
            # in the same groups with almost equal activations, so every max-pool they share is a NEAR-tie that the last
            # bits decide, and the winner takes the whole gradient of that channel.  Entries of duplicated points may
            # therefore flip at any |g| (measured: 5 entries of 98 304, |g| up to 0.17 max|g|); everywhere else the usual bar.
            diff = got.view(np.uint32) != nxt.view(np.uint32)
            on_dup = np.broadcast_to(dup_point[:, None, :], diff.shape)
            assert 1.0 - diff.mean() >= 0.999 and (diff & on_dup).sum() <= 0.001 * on_dup.sum()
            grad = _one_step.last_grad
            rest = diff & ~on_dup
            assert not rest.any() or np.abs(grad[rest]).max() <= 3e-3 * np.abs(grad).max()


def test_nb_b8_statistical_parity_32_rooms(gpu_model, golden_nb_b8):
    """Fused 40-iteration attack on the reference's FOUR batches of 8 rooms (pn2_nb_b8.npz + round 4's pn2_nb_more.npz) with
    the reference's FPS draws: clean counters equal (up to the handful of arg-max ties), adversarial accuracy / mIoU of
    each batch within 0.01 of the reference's, of the 32 rooms together within 0.005, per-room adversarial accuracy within
    0.03, L2 distance within 1 %."""
    import os
    from conftest import GOLDEN
    from pointsecguard_amd import runtime
    more = dict(np.load(os.path.join(GOLDEN, "pn2_nb_more.npz")))
    iters, eps, alpha = int(golden_nb_b8["iters"]), float(golden_nb_b8["eps"]), float(golden_nb_b8["alpha"])
    assert int(more["iters"]) == iters and float(more["eps"]) == eps and float(more["alpha"]) == alpha
    tot_ref, tot_got = np.zeros((3, 13)), np.zeros((3, 13))
    for g, bi in ((golden_nb_b8, 0), (golden_nb_b8, 1), (more, 0), (more, 1)):
        rooms, lab = _b8_batch(g, bi)
        p = "b%d_" % bi
        starts = g[p + "starts"]
        images_np = np.ascontiguousarray(rooms.transpose(0, 2, 1))
        images, labels = dev(images_np), dev(lab.astype(np.int32))
        ws = runtime.PN2Workspace(8, 4096, iters)
        adv = ws.nb_attack(gpu_model, images, labels, dev(starts[1:1 + iters], torch.int32), eps, alpha, iters)
        torch.cuda.synchronize()
        out = adv.cpu().numpy()
        assert np.array_equal(out[:, :3], images_np[:, :3]) and np.array_equal(out[:, 6:], images_np[:, 6:])
        assert np.abs(out[:, 3:6] - images_np[:, 3:6]).max() <= eps + alpha + 1e-6
        l2 = float(np.sqrt(((out - images_np).astype(np.float64) ** 2).sum()))
        assert abs(l2 - float(g[p + "l2_dis"])) <= 0.01 * float(g[p + "l2_dis"])
        ev = runtime.PN2Workspace(8, 4096, 1)
        # clean forward on the reference's clean RNG slot
        x0c = dev(rooms)
        ev.plan_build(x0c, dev(starts[0:1], torch.int32), 1)
        logp = ev.forward(gpu_model, 0, x0c)
        pred = logp.argmax(2).cpu().numpy()
        if p + "clean_pred" in g:
            assert (pred != g[p + "clean_pred"]).mean() <= 2e-4
        assert abs((pred == lab).mean() - g[p + "room_acc"].mean()) <= 2e-4
        # adversarial forward on the reference's slot
        x0a = adv.transpose(1, 2).contiguous()
        ev.plan_build(x0a, dev(starts[iters + 1:iters + 2], torch.int32), 1)
        counters, _ = runtime.seg_stats(ev.forward(gpu_model, 0, x0a), labels)
        apred = ev.forward(gpu_model, 0, x0a).argmax(2).cpu().numpy()
        torch.cuda.synchronize()
        c = counters.cpu().numpy().astype(np.float64)
        r = g[p + "adv_counters"].astype(np.float64)
        tot_ref += r
        tot_got += c
        acc, racc = c[1].sum() / c[0].sum(), r[1].sum() / r[0].sum()
        miou = np.mean((c[1] / (c[2] + 1e-6))[c[0] != 0])
        rmiou = np.mean((r[1] / (r[2] + 1e-6))[r[0] != 0])
        assert abs(acc - racc) <= 0.01 and abs(miou - rmiou) <= 0.01, (bi, acc, racc, miou, rmiou)
        room_acc = (apred == lab).mean(1)
        assert np.abs(room_acc - g[p + "room_adv_acc"]).max() <= 0.03, np.abs(room_acc - g[p + "room_adv_acc"]).max()
    acc, racc = tot_got[1].sum() / tot_got[0].sum(), tot_ref[1].sum() / tot_ref[0].sum()
    assert abs(acc - racc) <= 0.005, (acc, racc)
