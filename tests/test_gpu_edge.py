"""Size-independent properties and edge cases of the HIP PointNet++ path at BASELINE's full sizes.

* run-to-run determinism: no fp32 atomics anywhere on the path, so gradients and whole attacks are bit-identical;
* exact linearity of the backward pass in the upstream gradient (scaling by 2 is exact in every fp32 step);
* room-permutation equivariance (rooms are independent: what multi-GPU sharding relies on);
* smallest / largest supported clouds (1024 / 8192 points), batch 1 and ragged batch 3 against the oracle;
* degenerate attack inputs: empty and full masks, alpha = 0, eps = 0;
* degenerate geometry: every point in the ball (first 32 by index), lonely points (group padded with itself).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS, ALPHA, ITERS = 0.05, 2.0 / 255.0, 40


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda().contiguous()


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


def draw_starts(seed, iters, batch, n_point):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, n, (iters, batch)) for n in (n_point, 1024, 256, 64)], axis=1).astype(np.int32)


@pytest.fixture(scope="module")
def full_batch():
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    rooms = make_rooms(8, 4242, structured=True)
    return rooms, rule_labels(rooms), draw_starts(17, ITERS, 8, 4096)


def test_backward_is_deterministic_and_exactly_linear(gpu_model, full_batch):
    from pointsecguard_amd import runtime
    rooms, _, starts = full_batch
    ws = runtime.PN2Workspace(8, 4096, 1)
    x0 = dev(rooms)
    ws.plan_build(x0, dev(starts[:1]), 1)
    logp = ws.forward(gpu_model, 0, x0)
    g = torch.Generator(device="cpu").manual_seed(3)
    dlogp = (torch.randn(logp.shape, generator=g) * 1e-3).cuda()
    d1 = ws.backward(gpu_model, 0, dlogp).clone()
    d2 = ws.backward(gpu_model, 0, dlogp).clone()
    assert np.array_equal(bits(d1), bits(d2)), "backward is not bit-reproducible"
    d3 = ws.backward(gpu_model, 0, dlogp * 2.0).clone()
    assert np.array_equal(bits(d3), bits(d1 * 2.0)), "backward(2 g) != 2 backward(g) bit for bit"
    assert torch.isfinite(d1).all() and float(d1.abs().max()) > 0.0
    # a second forward of the same input reproduces the log-probs bit for bit
    assert np.array_equal(bits(ws.forward(gpu_model, 0, x0)), bits(logp))


def test_full_size_attack_deterministic_and_bounded(gpu_model, full_batch):
    """BASELINE configs[1]: 8 rooms x 4096 points, eps = 0.05, alpha = 2/255, 40 iterations."""
    from pointsecguard_amd import runtime
    rooms, labels, starts = full_batch
    images = dev(rooms.transpose(0, 2, 1))
    ws = runtime.PN2Workspace(8, 4096, ITERS)
    outs = []
    for _ in range(2):
        out = torch.empty_like(images)
        ws.nb_attack(gpu_model, images, dev(labels.astype(np.int32)), dev(starts), EPS, ALPHA, ITERS, out=out)
        torch.cuda.synchronize()
        outs.append(out)
    assert np.array_equal(bits(outs[0]), bits(outs[1])), "two runs of the same attack differ"
    adv, img = outs[0].cpu().numpy(), images.cpu().numpy()
    assert np.array_equal(adv[:, :3], img[:, :3]) and np.array_equal(adv[:, 6:], img[:, 6:])   # only colour moves
    d = np.abs(adv[:, 3:6] - img[:, 3:6])
    assert d.max() <= EPS + ALPHA + 1e-6     # the returned tensor is the un-projected last step (nontarget.py:36-41)
    assert np.isfinite(adv).all() and d.max() > 0.5 * EPS
    # the attack does what it is for: accuracy on the attacked rooms drops
    ev = runtime.PN2Workspace(8, 4096, 1)
    accs = []
    for x in (images, outs[0]):
        x0 = x.transpose(1, 2).contiguous()
        ev.plan_build(x0, dev(starts[:1]), 1)
        pred = ev.forward(gpu_model, 0, x0).argmax(-1).cpu().numpy()
        accs.append((pred == labels).mean())
    assert accs[1] < accs[0] - 0.02, accs


def test_room_permutation_equivariance(gpu_model, full_batch):
    from pointsecguard_amd import runtime
    rooms, _, starts = full_batch
    perm = np.array([5, 2, 7, 0, 3, 6, 1, 4])
    ws = runtime.PN2Workspace(8, 4096, 1)
    res = []
    for r, s in ((rooms, starts[:1]), (rooms[perm], starts[:1][:, :, perm])):
        x0 = dev(r)
        ws.plan_build(x0, dev(s), 1)
        logp = ws.forward(gpu_model, 0, x0).clone()
        dx = ws.backward(gpu_model, 0, torch.full_like(logp, 1e-4)).clone()
        res.append((logp, dx))
    assert np.array_equal(bits(res[0][0])[perm], bits(res[1][0]))
    assert np.array_equal(bits(res[0][1])[perm], bits(res[1][1]))


@pytest.mark.parametrize("n_point,batch", [(1024, 1), (1024, 3), (2048, 2)])
def test_small_clouds_vs_oracle(gpu_model, oracle_net, n_point, batch):
    from oracle import pn2
    from pointsecguard_amd import _lib, runtime
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    rooms = make_rooms(batch, 31 + n_point, num_point=n_point, structured=True)
    labels = rule_labels(rooms)
    starts = draw_starts(n_point, 1, batch, n_point)
    ws = runtime.PN2Workspace(batch, n_point, 1)
    x0 = dev(rooms)
    ws.plan_build(x0, dev(starts), 1)
    logp = ws.forward(gpu_model, 0, x0)
    dlogp = torch.empty_like(logp)
    _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(dev(labels.astype(np.int32))), 0, batch * n_point,
              batch * n_point, 13, 1.0 / n_point, runtime.ptr(dlogp), None, runtime.stream())
    dx0 = ws.backward(gpu_model, 0, dlogp)
    torch.cuda.synchronize()
    for b in range(batch):
        geom = oracle_net.geometry(rooms[b, :, :3], starts[0, :, b])
        for lvl in range(4):
            assert np.array_equal(ws.plan_tensor(1, lvl, 0, b).cpu().numpy(), geom["group"][lvl].astype(np.int32)), lvl
        o_logp, cache = oracle_net.forward(rooms[b], geom)
        assert np.abs(logp[b].cpu().numpy() - o_logp).max() <= 1e-4
        o_dlogp, _ = pn2.nll_logp_grad(o_logp, labels[b], 1.0 / n_point)
        ref = oracle_net.backward_color(cache, o_dlogp)
        got = dx0[b, :, 3:6].cpu().numpy()
        nz = ref != 0
        assert (np.sign(got[nz]) == np.sign(ref[nz])).mean() >= 0.999
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-9


def test_largest_cloud_runs_and_is_deterministic(gpu_model):
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    n_point, batch, iters = 8192, 2, 3
    rooms = make_rooms(batch, 77, num_point=n_point, structured=True)
    labels = rule_labels(rooms)
    starts = draw_starts(9, iters, batch, n_point)
    images = dev(rooms.transpose(0, 2, 1))
    ws = runtime.PN2Workspace(batch, n_point, iters)
    a = torch.empty_like(images)
    b = torch.empty_like(images)
    ws.nb_attack(gpu_model, images, dev(labels.astype(np.int32)), dev(starts), EPS, ALPHA, iters, out=a)
    ws.nb_attack(gpu_model, images, dev(labels.astype(np.int32)), dev(starts), EPS, ALPHA, iters, out=b)
    torch.cuda.synchronize()
    assert np.array_equal(bits(a), bits(b))
    assert torch.isfinite(a).all() and float((a[:, 3:6] - images[:, 3:6]).abs().max()) <= iters * ALPHA + 1e-6


def test_degenerate_attack_inputs(gpu_model, full_batch):
    from pointsecguard_amd import runtime
    rooms, labels, starts = full_batch
    rooms, labels, starts = rooms[:2], labels[:2], np.ascontiguousarray(starts[:4, :, :2])
    images = dev(rooms.transpose(0, 2, 1))
    lab = dev(labels.astype(np.int32))
    ws = runtime.PN2Workspace(2, 4096, 4)
    out = torch.empty_like(images)
    # alpha = 0: nothing moves
    ws.nb_attack(gpu_model, images, lab, dev(starts), EPS, 0.0, 4, out=out)
    assert np.array_equal(bits(out), bits(images))
    # eps = 0: the projection pins the colours, only the un-projected last step (one alpha) shows
    ws.nb_attack(gpu_model, images, lab, dev(starts), 0.0, ALPHA, 4, out=out)
    assert float((out[:, 3:6] - images[:, 3:6]).abs().max()) <= ALPHA + 1e-7
    # targeted attack with an EMPTY mask leaves the cloud untouched; with a FULL mask only colour moves
    empty = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    ws.nb_attack(gpu_model, images, lab, dev(starts), EPS, ALPHA, 4, mask=empty, target=5, out=out)
    assert np.array_equal(bits(out), bits(images))
    full = torch.ones(4096, dtype=torch.uint8, device="cuda")
    ws.nb_attack(gpu_model, images, lab, dev(starts), EPS, ALPHA, 4, mask=full, target=5, out=out)
    assert np.array_equal(bits(out[:, :3]), bits(images[:, :3])) and np.array_equal(bits(out[:, 6:]), bits(images[:, 6:]))
    # tar_NB: the loss is batch row 0 only (target.py:36-39), so row 1 receives no gradient and never moves
    assert np.array_equal(bits(out[1]), bits(images[1])) and not np.array_equal(bits(out[0]), bits(images[0]))


def test_degenerate_geometry():
    """Ball query: a cloud inside one ball returns the first 32 indices for every centroid; a lonely centroid is
    padded with itself (pointnet_util.py:100-106).  FPS of all points returns a permutation."""
    from pointsecguard_amd import runtime
    g = torch.Generator().manual_seed(1)
    tight = (torch.rand(1, 256, 3, generator=g) * 0.01).cuda()
    cent = tight[:, :16].contiguous()
    idx = runtime.ball_query(0.5, 32, tight, cent).cpu().numpy()
    assert np.array_equal(idx, np.broadcast_to(np.arange(32), (1, 16, 32)))
    lonely = torch.zeros(1, 64, 3)
    lonely[0, :, 0] = torch.arange(64, dtype=torch.float32) * 10.0
    lonely = lonely.cuda()
    idx = runtime.ball_query(0.2, 32, lonely, lonely).cpu().numpy()
    assert np.array_equal(idx, np.broadcast_to(np.arange(64)[None, :, None], (1, 64, 32)))
    fps = runtime.fps(lonely, 64, torch.tensor([7], dtype=torch.int32).cuda()).cpu().numpy()
    assert sorted(fps[0].tolist()) == list(range(64)) and fps[0, 0] == 7


def test_fps_many_problems_path():
    """More than 512 problems take the 256-thread x 16-point FPS kernel: same indices as the oracle."""
    from oracle import pn2
    from pointsecguard_amd import runtime
    from pointsecguard_amd.synthetic import make_rooms
    P, N, S = 520, 4096, 1024
    xyz = np.ascontiguousarray(make_rooms(4, 5)[:, :, :3])
    rng = np.random.default_rng(0)
    start = rng.integers(0, N, P).astype(np.int32)
    clouds = torch.from_numpy(xyz[np.arange(P) % 4]).cuda().contiguous()
    idx = runtime.fps(clouds, S, torch.from_numpy(start).cuda()).cpu().numpy()
    for p in (0, 257, 519):
        assert np.array_equal(idx[p], pn2.fps(xyz[p % 4], S, int(start[p])).astype(np.int32)), p
    assert all(len(set(row.tolist())) == S for row in idx[::37])


def test_concurrent_attacks_on_two_streams_match_serial(gpu_model, full_batch):
    """bench.py keeps two attacks in flight on separate HIP streams with separate workspaces: the overlapped runs must
    return exactly what the same attacks return one after the other (the kernels share the model's weights only)."""
    from pointsecguard_amd import runtime
    rooms, labels, starts = full_batch
    iters = 6
    imgs = [dev(rooms[:4].transpose(0, 2, 1)), dev(rooms[4:].transpose(0, 2, 1))]
    labs = [dev(labels[:4].astype(np.int32)), dev(labels[4:].astype(np.int32))]
    sts = [dev(np.ascontiguousarray(starts[:iters, :, :4])), dev(np.ascontiguousarray(starts[:iters, :, 4:]))]
    wss = [runtime.PN2Workspace(4, 4096, iters) for _ in range(2)]
    serial = []
    for i in range(2):
        serial.append(wss[i].nb_attack(gpu_model, imgs[i], labs[i], sts[i], EPS, ALPHA, iters).clone())
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    outs = [torch.empty_like(imgs[0]), torch.empty_like(imgs[1])]
    for rep in range(3):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                wss[i].nb_attack(gpu_model, imgs[i], labs[i], sts[i], EPS, ALPHA, iters, out=outs[i])
    torch.cuda.synchronize()
    for i in range(2):
        assert np.array_equal(bits(outs[i]), bits(serial[i])), i
