"""GPU parity tests of the MSG network: the HIP path through the C ABI against the
fixtures generated from the reference's pointnet2_sem_seg_msg (tests/golden/make_golden_msg.py) and against the CPU
oracle (oracle/pn2_msg.py) on the same seeded inputs.  Same bars as test_gpu_parity.py: grouping indices bit-exact;
activations / log-probs within 1e-4 (relative to the tensor's magnitude where that exceeds 1); colour gradient with
identical zero pattern and >= 99.9 % sign agreement; teacher-forced attack steps bit-equal on >= 99.9 % of entries."""
import os

import numpy as np
import pytest
import torch

from pointsecguard_amd.synthetic import make_rooms, msg_state_dict, rule_labels

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda().contiguous()


@pytest.fixture(scope="module")
def room():
    return dict(np.load(os.path.join(GOLD, "pn2msg_room.npz")))


@pytest.fixture(scope="module")
def nb():
    return dict(np.load(os.path.join(GOLD, "pn2msg_nb.npz")))


@pytest.fixture(scope="module")
def sd(room):
    return msg_state_dict(int(room["msg_seed"]))


@pytest.fixture(scope="module")
def model(sd):
    from pointsecguard_amd import runtime
    return runtime.PN2Model(runtime.fold_state_dict(sd, msg=True), arch=runtime.ARCH_MSG)


@pytest.fixture(scope="module")
def oracle(sd):
    from oracle import pn2_msg
    return pn2_msg.PN2MsgOracle(sd)


@pytest.fixture(scope="module")
def room_run(model, room):
    from pointsecguard_amd import runtime
    ws = runtime.PN2Workspace(1, 4096, 2, arch=runtime.ARCH_MSG)
    x0 = dev(room["room"][None])
    ws.plan_build(x0, dev(room["starts"].reshape(1, 4, 1), torch.int32), 1)
    logp = ws.forward(model, 0, x0)
    torch.cuda.synchronize()
    return ws, x0, logp


def test_msg_geometry_bit_exact(room_run, room):
    ws, _, _ = room_run
    for lvl in range(4):
        assert np.array_equal(ws.plan_tensor(0, lvl, 0, 0).cpu().numpy(), room["fps%d" % lvl].astype(np.int32)), lvl
        assert np.array_equal(ws.plan_tensor(1, lvl, 0, 0).cpu().numpy(), room["group%d_0" % lvl].astype(np.int32)), lvl
        assert np.array_equal(ws.plan_tensor(5, lvl, 0, 0).cpu().numpy(), room["group%d_1" % lvl].astype(np.int32)), lvl


def test_msg_forward_vs_reference(room_run, room):
    ws, _, logp = room_run
    for which, name in enumerate(("sa1", "sa2", "sa3", "sa4", "fp4", "fp3", "fp2")):
        a = ws.activation(which)[0].cpu().numpy()
        ref = room["act_" + name]
        assert a.shape == ref.shape, name
        assert np.abs(a - ref).max() <= TOL * max(1.0, np.abs(ref).max()), name
    assert np.abs(logp[0].cpu().numpy() - room["logp"]).max() <= TOL


def check_grad(ours, ref):
    nz = ref != 0
    assert np.array_equal(ours != 0, nz), "zero pattern of the colour gradient differs"
    agree = np.sign(ours[nz]) == np.sign(ref[nz])
    assert agree.mean() >= 0.999
    if not agree.all():
        assert np.abs(ref[nz][~agree]).max() <= 1e-3 * np.abs(ref).max()
    rel = np.abs(ours - ref)[nz] / np.abs(ref[nz])
    assert np.median(rel) < 1e-4


def _ce_grad(logp, labels, n):
    from pointsecguard_amd import _lib, runtime
    rows = logp.shape[0] * logp.shape[1]
    dlogp = torch.empty_like(logp)
    cost = torch.zeros(1, device="cuda")
    _lib.call("psg_ce_logp_grad", runtime.ptr(logp), runtime.ptr(labels), 0, rows, rows, 13, 1.0 / n,
              runtime.ptr(dlogp), runtime.ptr(cost), runtime.stream())
    return dlogp, cost


def test_msg_backward_vs_reference(room_run, room, model):
    ws, x0, logp = room_run
    labels = dev(room["labels"].astype(np.int32)[None])
    dlogp, cost = _ce_grad(logp, labels, 4096)
    dx0 = ws.backward(model, 0, dlogp)
    torch.cuda.synchronize()
    assert abs(cost.item() - float(room["cost"])) < 1e-4
    check_grad(dx0[0, :, 3:6].cpu().numpy(), room["dcolor"])


def test_msg_l4_points(room_run, room, model):
    """the second output of get_model.forward: l4_points [1024 channels x 16 points]"""
    ws, x0, _ = room_run
    l4 = torch.empty(1, 16, 1024, device="cuda")
    ws.forward(model, 0, x0, l4=l4)
    torch.cuda.synchronize()
    assert np.abs(l4[0].cpu().numpy().T - room["l4"]).max() <= TOL * max(1.0, np.abs(room["l4"]).max())


TIE_TOL = 2e-5      # a max-pool gap below this (relative to max(1, |winner|)) is inside the two implementations' rounding difference


def _n_beyond(ours, ref, rel=1e-3):
    nz = ref != 0
    return int((np.abs(ours - ref)[nz] > rel * np.abs(ref[nz])).sum()) + int((ours != 0)[~nz].sum())


def prove_near_ties(oracle, cache, dl, ours, geom):
    """A room whose colour gradient misses the strict bars of check_grad is accepted ONLY with a proof that the deviation
    is a max-pool near-tie that fell the other way than in the oracle (round-5 advisor: the round-5 test accepted any room
    with the same zero pattern, >= 99.8 % sign agreement and a small median error, which a mis-routed pooled channel of the
    split first-layer kernels would also have shown).  The proof: (1) list every pooled (level, scale, group, channel) of
    the ORACLE's forward whose winner leads the best row of ANOTHER source point by less than TIE_TOL (rows of the same
    source point - ball-query padding - route the gradient to the same place whichever wins) and whose upstream gradient
    is not zero; (2) hand the oracle's arg-max of such a channel to the runner-up, re-run the oracle's backward, keep the
    flip iff the number of entries beyond 1e-3 relative falls (a tie moves one channel's whole gradient, so the flips are
    independent of each other); (3) with the kept flips the STRICT bars of check_grad must hold.  At least one flip must
    have been kept, every flip is a listed near-tie, and nothing else in the oracle is touched - a routing or mask bug in
    the library cannot be repaired by re-deciding ties of width 2e-5.  Returns the kept flips."""
    from oracle import pn2_msg
    ref = oracle.backward_color(cache, dl)                       # (fills cache["dsa"]: the upstream gradients per level)
    base = _n_beyond(ours, ref)
    cands = []
    for lvl in range(4):
        npoint, _, nsamples = pn2_msg.SA_CFG[lvl]
        c_off = 0
        for i, ns in enumerate(nsamples):
            acts, arg = cache["sa"][lvl][i]
            c = acts[-1].shape[1]
            h = acts[-1].reshape(npoint, ns, c)
            src = geom["group"][lvl][i].reshape(npoint, ns)
            win = np.take_along_axis(h, arg[:, None, :], axis=1)[:, 0, :]                      # [npoint, c]
            win_src = np.take_along_axis(src, arg.reshape(npoint, c), axis=1)                  # [npoint, c] source point of the winner
            other = np.where(src[:, :, None] == win_src[:, None, :], -np.inf, h)                # rows of other source points
            second = other.argmax(axis=1)
            gap = win - other.max(axis=1)
            up = cache["dsa"][lvl + 1][:, c_off:c_off + c]
            # (win > 0: the pooled activations are post-ReLU; a channel whose maximum is 0 passes no gradient whoever wins)
            near = np.isfinite(gap) & (gap <= TIE_TOL * np.maximum(1.0, np.abs(win))) & (up != 0) & (win > 0)
            for g, ch in zip(*np.nonzero(near)):
                cands.append((lvl, i, int(g), int(ch), int(second[g, ch]), float(gap[g, ch])))
            c_off += c
    assert 0 < len(cands) <= 24, "no max-pool near-tie in the oracle's forward explains the deviation (%d candidates)" % len(cands)
    kept = []
    for lvl, i, g, ch, second, gap in cands:
        arg = cache["sa"][lvl][i][1]
        old = int(arg[g, ch])
        arg[g, ch] = second
        n = _n_beyond(ours, oracle.backward_color(cache, dl))
        if n < base:
            base = n
            kept.append((lvl, i, g, ch, gap))
        else:
            arg[g, ch] = old
    assert kept, "re-deciding the %d near-ties of the oracle's forward does not explain the deviation" % len(cands)
    check_grad(ours, oracle.backward_color(cache, dl))           # the STRICT bars, with only proven ties re-decided
    return kept


def test_msg_batch_vs_oracle(model, oracle):
    """B = 3 rooms of 2048 points (another N than the fixture), full 9-channel gradient against the oracle: the strict bars
    of check_grad for every room; a room that misses them must be PROVEN a max-pool near-tie room (prove_near_ties: at most
    one room of the batch, and the strict bars must hold once the oracle's proven ties are decided the library's way)."""
    from oracle import pn2
    from pointsecguard_amd import runtime
    B, N = 3, 2048
    rooms = make_rooms(B, 515, num_point=N)
    labels = rule_labels(rooms)
    torch.manual_seed(3)
    starts = np.stack([torch.randint(0, n, (B,)).numpy() for n in (N, 1024, 256, 64)]).astype(np.int32)
    ws = runtime.PN2Workspace(B, N, 1, arch=runtime.ARCH_MSG)
    x0 = dev(rooms)
    ws.plan_build(x0, dev(starts.reshape(1, 4, B), torch.int32), 1)
    logp = ws.forward(model, 0, x0)
    dlogp, _ = _ce_grad(logp, dev(labels.astype(np.int32)), N)
    dx0 = ws.backward(model, 0, dlogp)
    torch.cuda.synchronize()
    n_tie_rooms = 0
    for b in range(B):
        geom = oracle.geometry(rooms[b, :, :3], starts[:, b])
        for lvl in range(4):
            assert np.array_equal(ws.plan_tensor(0, lvl, 0, b).cpu().numpy(), geom["fps"][lvl])
            assert np.array_equal(ws.plan_tensor(1, lvl, 0, b).cpu().numpy(), geom["group"][lvl][0])
            assert np.array_equal(ws.plan_tensor(5, lvl, 0, b).cpu().numpy(), geom["group"][lvl][1])
        lp, cache = oracle.forward(rooms[b], geom)
        assert np.abs(logp[b].cpu().numpy() - lp).max() <= TOL
        dl, _ = pn2.nll_logp_grad(lp, labels[b], 1.0 / N)
        ours, ref = dx0[b, :, 3:6].cpu().numpy(), oracle.backward_color(cache, dl)
        try:
            check_grad(ours, ref)
        except AssertionError:
            n_tie_rooms += 1
            assert n_tie_rooms <= 1, b
            kept = prove_near_ties(oracle, cache, dl, ours, geom)
            print("room %d: near-tie room, proven: %s" % (b, kept))


def test_msg_nb_attack_steps_vs_reference(model, nb):
    """Teacher-forced NB_attack iterations through the MSG network (see test_gpu_parity.py for the rationale)."""
    from pointsecguard_amd import _lib, runtime
    g = nb
    rooms, iters = g["rooms"], int(g["iters"])
    B = rooms.shape[0]
    labels = dev(g["labels"].astype(np.int32))
    x0 = dev(rooms)
    ori = x0[:, :, 3:6].contiguous()
    ws = runtime.PN2Workspace(B, 4096, iters, arch=runtime.ARCH_MSG)
    ws.plan_build(x0, dev(g["starts"][1:1 + iters], torch.int32), iters)
    for t in range(iters):
        nxt = g["adv_color_final"] if t == iters - 1 else g["state_it%d" % (t + 1)]
        x0[:, :, 3:6] = dev(np.ascontiguousarray(g["state_it%d" % t].transpose(0, 2, 1)))
        logp = ws.forward(model, t, x0)
        dlogp, _ = _ce_grad(logp, labels, 4096)
        dx0 = ws.backward(model, t, dlogp)
        _lib.call("psg_pgd_step", runtime.ptr(x0), runtime.ptr(dx0), runtime.ptr(ori), None, B, 4096,
                  float(g["alpha"]), float(g["eps"]), 1.0, 1 if t == iters - 1 else 0, runtime.stream())
        torch.cuda.synchronize()
        got = np.ascontiguousarray(x0[:, :, 3:6].cpu().numpy().transpose(0, 2, 1))
        same = (got.view(np.uint32) == nxt.view(np.uint32)).mean()
        assert same >= 0.999, (t, same)


def test_msg_fused_attack_and_module_api(nb, sd):
    """The reference-shaped module (models.pointnet2_sem_seg_msg.get_model) under the reference-shaped NB_attack:
    same result as the fused C-ABI loop fed the same FPS draws, deterministic, and inside the eps ball."""
    from pointsecguard_amd import runtime
    from pointsecguard_amd.attacks import torchattacks
    from pointsecguard_amd.models import pointnet2_sem_seg_msg as msg
    g = nb
    iters, eps, alpha = int(g["iters"]), float(g["eps"]), float(g["alpha"])
    net = msg.get_model(13).cuda().eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    images_np = np.ascontiguousarray(g["rooms"].transpose(0, 2, 1))
    images = dev(images_np)
    with torch.no_grad():
        torch.manual_seed(int(g["seed_rng"]))
        logp, l4 = net(images)
    assert tuple(l4.shape) == (2, 1024, 16)
    assert np.abs(logp.cpu().numpy() - g["clean_logp"]).max() <= TOL
    torch.manual_seed(int(g["seed_rng"]))
    net(images)   # consume the clean forward's draws like the fixture's run
    atk = torchattacks.NB_attack(net, eps=eps, alpha=alpha, iters=iters)
    adv = atk(images, g["labels"].astype(np.float64))
    torch.cuda.synchronize()
    out = adv.cpu().numpy()
    assert np.array_equal(out[:, :3], images_np[:, :3]) and np.array_equal(out[:, 6:], images_np[:, 6:])
    assert np.abs(out[:, 3:6] - images_np[:, 3:6]).max() <= eps + alpha + 1e-6
    ws = runtime.PN2Workspace(2, 4096, iters, arch=runtime.ARCH_MSG)
    fused = ws.nb_attack(net._packed(), images, dev(g["labels"].astype(np.int32)),
                         dev(g["starts"][1:1 + iters], torch.int32), eps, alpha, iters)
    torch.cuda.synchronize()
    assert np.array_equal(fused.cpu().numpy().view(np.uint32), out.view(np.uint32))
    same = (out[:, 3:6].view(np.uint32) == g["adv_color_final"].view(np.uint32)).mean()
    assert same >= 0.9, same   # 6 free-running iterations: a few amplified sign flips at most


def test_msg_arch_mismatch_is_refused(model):
    from pointsecguard_amd import _lib, runtime
    ws = runtime.PN2Workspace(1, 1024, 1)   # SSG workspace
    x0 = dev(make_rooms(1, 1, num_point=1024))
    ws.plan_build(x0, dev(np.zeros((1, 4, 1), np.int32)), 1)
    with pytest.raises(_lib.PsgError):
        ws.forward(model, 0, x0)


def test_msg_nu_step_vs_oracle(model, oracle):
    """One NU_attack optimiser step (tanh-space Adam, f-loss + Smooth + L2) through the MSG network against the CPU
    oracle's restatement of the same step (oracle/attacks.py: nu_step is generic in the network oracle)."""
    from oracle import attacks as oatk
    from pointsecguard_amd import runtime
    from test_gpu_nu import nu_step_gpu
    N, c, kappa, lr, nb = 2048, 0.1, 0.0, 0.01, 10
    rooms = make_rooms(1, 616, num_point=N)
    labels = rule_labels(rooms)
    torch.manual_seed(9)
    starts = np.stack([torch.randint(0, n, (1,)).numpy() for n in (N, 1024, 256, 64)]).astype(np.int32)[None]   # [1,4,1]
    images = np.ascontiguousarray(rooms.transpose(0, 2, 1))
    w0 = oatk.inverse_tanh_space(images[:, 3:6]).astype(np.float32)          # [1,3,N]
    zeros = np.zeros_like(w0)
    ref = oatk.nu_step(oracle, images, images.copy(), w0, zeros, zeros, 1, labels, starts[0], c, kappa, lr, nb)
    ws = runtime.PN2Workspace(1, N, 1, arch=runtime.ARCH_MSG)
    x0 = dev(rooms)
    ori = x0[:, :, 3:6].contiguous()
    ws.plan_build(x0, dev(starts, torch.int32), 1)
    w = dev(np.ascontiguousarray(w0.transpose(0, 2, 1)))
    r = nu_step_gpu(ws, model, x0, ori, w, torch.zeros_like(w), torch.zeros_like(w), 1, dev(labels.astype(np.int32)), None,
                    None, c, kappa, lr, nb, 0)
    assert abs(r["cost"] - ref["cost"]) <= 1e-4 * abs(ref["cost"]) + 0.02
    got_g, ref_g = r["grad"].transpose(0, 2, 1), ref["grad_w"]
    rel = np.abs(got_g - ref_g) / (np.abs(ref_g) + 1e-12)
    assert np.median(rel) < 1e-3
    assert (np.abs(got_g - ref_g) <= 1e-2 * np.abs(ref_g).max()).mean() >= 0.99
    assert (np.abs(r["w"].transpose(0, 2, 1) - ref["w"]) <= 1e-4).mean() >= 0.99


def test_msg_whole_scene_harness(tmp_path, sd):
    """The whole-scene evaluation loop with the MSG classifier under NB_attack: runs, is reproducible,
    and sees every scene point once per vote pool."""
    from pointsecguard_amd import harness
    from pointsecguard_amd.attacks import torchattacks
    from pointsecguard_amd.models import pointnet2_sem_seg_msg as msg
    from test_gpu_harness import synth_scene
    ds = harness.ScannetDatasetWholeScene(None, block_points=1024, scenes={"Area_5_a.npy": synth_scene(21, 4000, 1.4, 1.2)})
    net = msg.get_model(13).cuda().eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    res = []
    for rep in range(2):
        np.random.seed(4)
        torch.manual_seed(4)
        res.append(harness.evaluate_whole_scene(net, ds, lambda m: torchattacks.NB_attack(m, eps=0.1, alpha=0.05, iters=2),
                                                batch_size=4, num_votes=1, log_path=str(tmp_path / ("l%d.txt" % rep)),
                                                log=lambda *_: None))
    assert np.array_equal(res[0]["counters"], res[1]["counters"])
    assert res[0]["counters"][0][0].sum() == 4000 and res[0]["counters"][1][0].sum() == 4000
    assert (tmp_path / "l0.txt").read_text() == (tmp_path / "l1.txt").read_text()


def test_msg_bit_reproducible_and_stream_safe(model, nb):
    """No floating-point atomics on this path either: two fused attacks give bit-identical results, also when a second
    attack runs concurrently on another stream with its own workspace."""
    from pointsecguard_amd import runtime
    g = nb
    iters = 4
    images = dev(np.ascontiguousarray(g["rooms"].transpose(0, 2, 1)))
    labels = dev(g["labels"].astype(np.int32))
    starts = dev(g["starts"][1:1 + iters], torch.int32)
    ws = [runtime.PN2Workspace(2, 4096, iters, arch=runtime.ARCH_MSG) for _ in range(2)]
    ref = ws[0].nb_attack(model, images, labels, starts, 0.05, 2 / 255, iters).clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = []
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            outs.append(ws[i].nb_attack(model, images, labels, starts, 0.05, 2 / 255, iters))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


@pytest.mark.parametrize("n_point,batch", [(1024, 2), (8192, 1)])
def test_msg_smallest_and_largest_clouds(model, oracle, n_point, batch):
    """Workspace limits (1024 / 8192 points): geometry bit-exact, log-probs and colour gradient against the oracle."""
    from oracle import pn2
    from pointsecguard_amd import runtime
    rooms = make_rooms(batch, 900 + n_point, num_point=n_point)
    labels = rule_labels(rooms)
    torch.manual_seed(n_point)
    starts = np.stack([torch.randint(0, n, (batch,)).numpy() for n in (n_point, 1024, 256, 64)]).astype(np.int32)
    ws = runtime.PN2Workspace(batch, n_point, 1, arch=runtime.ARCH_MSG)
    x0 = dev(rooms)
    ws.plan_build(x0, dev(starts.reshape(1, 4, batch), torch.int32), 1)
    logp = ws.forward(model, 0, x0)
    dlogp, _ = _ce_grad(logp, dev(labels.astype(np.int32)), n_point)
    dx0 = ws.backward(model, 0, dlogp)
    torch.cuda.synchronize()
    for b in range(batch):
        geom = oracle.geometry(rooms[b, :, :3], starts[:, b])
        for lvl in range(4):
            assert np.array_equal(ws.plan_tensor(1, lvl, 0, b).cpu().numpy(), geom["group"][lvl][0])
            assert np.array_equal(ws.plan_tensor(5, lvl, 0, b).cpu().numpy(), geom["group"][lvl][1])
        lp, cache = oracle.forward(rooms[b], geom)
        assert np.abs(logp[b].cpu().numpy() - lp).max() <= TOL
        dl, _ = pn2.nll_logp_grad(lp, labels[b], 1.0 / n_point)
        check_grad(dx0[b, :, 3:6].cpu().numpy(), oracle.backward_color(cache, dl))
