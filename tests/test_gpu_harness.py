"""GPU tests of the whole-scene harness: vote pool / vote statistics / L2 kernels against the oracle
on the reference-generated fixture, and the end-to-end evaluation loop on synthetic S3DIS-format scenes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness.npz")


@pytest.fixture(scope="module")
def g():
    return dict(np.load(GOLDEN))


def dev(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda().contiguous()


def test_vote_kernels_vs_reference_fixture(g):
    from oracle import harness as oh
    from pointsecguard_amd import harness
    for si in range(len(g["file_list"])):
        n_pts = g["scene%d" % si].shape[0]
        idx, pred, w = g["index_room%d" % si], g["pred%d" % si], g["weight%d" % si]
        pool = torch.zeros(n_pts, 13, dtype=torch.int32, device="cuda")
        harness.add_vote(pool, dev(idx, torch.int32), dev(pred, torch.int32), dev(w, torch.float32))
        assert np.array_equal(pool.cpu().numpy().astype(np.float64), g["pool%d" % si])      # the reference's add_vote
        # the same votes from log-probs: argmax with the first index on ties
        logp = np.full(pred.shape + (13,), -5.0, np.float32)
        np.put_along_axis(logp, pred[..., None], -0.1, axis=2)
        logp[0, 0, :] = -1.0                                        # an all-tie row votes for class 0
        pred2 = pred.copy(); pred2[0, 0] = 0
        pool2 = torch.zeros(n_pts, 13, dtype=torch.int32, device="cuda")
        harness.add_vote(pool2, dev(idx, torch.int32), dev(logp, torch.float32), dev(w, torch.float32))
        assert np.array_equal(pool2.cpu().numpy().astype(np.float64), oh.add_vote(np.zeros((n_pts, 13)), idx, pred2, w))
        labels = g["scene%d" % si][:, 6]
        c, vp = harness.vote_stats(pool, dev(labels, torch.int32), want_pred=True)
        c = c.cpu().numpy()
        assert np.array_equal(vp.cpu().numpy(), g["vote_pred%d" % si])
        assert np.array_equal(c[0], g["seen%d" % si]) and np.array_equal(c[1], g["correct%d" % si])
        assert np.array_equal(c[2], g["deno%d" % si])
        assert harness._miou(torch.from_numpy(c)) == pytest.approx(float(g["miou%d" % si]), abs=1e-12)
    # weight = None counts every row; an index past the pool is an error, not a silent write
    pool = torch.zeros(10, 13, dtype=torch.int32, device="cuda")
    idx = torch.tensor([[0, 3, 3, 9]], dtype=torch.int32, device="cuda")
    harness.add_vote(pool, idx, torch.tensor([[1, 2, 2, 12]], dtype=torch.int32, device="cuda"), None)
    assert pool.sum().item() == 4 and pool[3, 2].item() == 2 and pool[9, 12].item() == 1
    with pytest.raises(IndexError):
        harness.add_vote(pool, torch.tensor([[10]], dtype=torch.int32, device="cuda"),
                         torch.tensor([[0]], dtype=torch.int32, device="cuda"), None)
    with pytest.raises(Exception):
        harness.add_vote(pool.cpu(), idx, idx, None)                # no CPU path


def test_l2_distance_vs_oracle():
    from oracle import harness as oh
    from pointsecguard_amd import harness
    gen = torch.Generator().manual_seed(5)
    a = torch.rand(8, 9, 4096, generator=gen)
    b = a + 0.05 * torch.rand(8, 9, 4096, generator=gen)
    got = harness.l2_distance(a.cuda(), b.cuda())
    ref = oh.l2_dist(a.numpy(), b.numpy())
    assert abs(float(got.item()) - ref) <= 1e-6 * ref
    assert abs(float(got.item()) - float(torch.dist(a, b))) <= 1e-4 * ref          # what the reference prints with %.3f
    assert float(harness.l2_distance(a.cuda(), a.cuda()).item()) == 0.0


def synth_scene(seed, n, size_x, size_y):
    rng = np.random.default_rng(seed)
    xyz = rng.random((n, 3)) * np.array([size_x, size_y, 2.8])
    rgb = np.floor(rng.random((n, 3)) * 256.0)
    label = rng.integers(0, 13, n).astype(np.float64)
    return np.concatenate([xyz, rgb, label[:, None]], axis=1)


def test_whole_scene_loop_end_to_end(tmp_path, weights_sd):
    from pointsecguard_amd import harness
    from pointsecguard_amd.attacks import torchattacks
    from pointsecguard_amd.models.pointnet2_sem_seg import get_model
    scenes = {"Area_5_a.npy": synth_scene(11, 5000, 1.6, 1.2), "Area_5_b.npy": synth_scene(12, 3000, 1.0, 1.4)}
    ds = harness.ScannetDatasetWholeScene(None, block_points=1024, scenes=scenes)
    net = get_model(13).cuda()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights_sd.items()})
    net.eval()
    results, logs = [], []
    for rep in range(3):
        np.random.seed(3)
        torch.manual_seed(3)
        path = tmp_path / ("log%d.txt" % rep)
        lines = []
        # (round 5) reps 0, 1: the default - three batches of a scene in flight on separate streams and network replicas;
        # rep 2: one batch at a time on the caller's stream, as rounds 1-4 ran it: same rows, same counters
        results.append(harness.evaluate_whole_scene(
            net, ds, lambda m: torchattacks.NB_attack(m, eps=0.1, alpha=0.05, iters=3), batch_size=4, num_votes=1,
            log_path=str(path), log=lines.append, **({"streams": 1} if rep == 2 else {})))
        logs.append(path.read_text())
    assert logs[0] == logs[1] and np.array_equal(results[0]["counters"], results[1]["counters"])   # reproducible
    assert logs[0] == logs[2] and np.array_equal(results[0]["counters"], results[2]["counters"])   # streams do not change results
    rows = logs[0].splitlines()
    assert rows[0] == "index\tL2_dis\tadv_acc\tacc\tadv_miou\tmiou"
    n_batches = sum(-(-ds[i][0].shape[0] // 4) for i in range(2))
    assert len(rows) == 1 + n_batches
    for r in rows[1:]:
        f = r.split("\t")
        assert len(f) == 7 and f[5] == ""                     # the reference's format has a doubled tab before miou
        assert 0.0 <= float(f[2]) <= 1.0 and 0.0 <= float(f[3]) <= 1.0 and float(f[1]) > 0.0
    c = results[0]["counters"]
    assert c[0][0].sum() == 8000 and c[1][0].sum() == 8000    # every scene point is seen exactly once per pool
    assert (c[0][1] <= c[0][0]).all() and (c[0][2] >= c[0][0]).all()
    assert len(results[0]["scenes"]) == 2 and 0.0 <= results[0]["miou"] <= 1.0
    assert any(l.startswith("eval whole scene point accuracy") for l in lines)
    # clean evaluation (test_semseg.py): no attack object, adversarial columns repeat the clean ones
    np.random.seed(3)
    torch.manual_seed(3)
    clean = harness.evaluate_whole_scene(net, ds, None, batch_size=4, log=lambda *_: None)
    assert np.array_equal(clean["counters"][0], clean["counters"][1])
    assert np.array_equal(clean["counters"][0][0], c[0][0])


def test_whole_scene_sharded_over_ranks(tmp_path):
    """Scenes dealt round-robin to two ranks (gloo group, both on this GPU) + one all-reduce of the counters give the
    totals of the single-process run (the vote pool of a scene never leaves its rank)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    one, two = tmp_path / "one.json", tmp_path / "two.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, os.path.join(here, "_harness_rank.py"), str(one)], check=True, env=env, timeout=300)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", "29533", os.path.join(here, "_harness_rank.py"), str(two)], check=True,
                   env=env, timeout=300)
    a, b = json.load(open(one)), json.load(open(two))
    assert a["counters"] == b["counters"]
    assert a["miou"] == b["miou"] and a["adv_miou"] == b["adv_miou"]


def test_whole_scene_targeted_protocol(tmp_path, weights_sd):
    """The targeted scripts' protocol (NB_target_test_semseg.py): per-batch mask of the origin class, batches without it
    are not attacked and not logged, extra columns; colours outside the first block's mask never move."""
    from pointsecguard_amd import harness
    from pointsecguard_amd.attacks import torchattacks
    from pointsecguard_amd.models.pointnet2_sem_seg import get_model
    ds = harness.ScannetDatasetWholeScene(None, block_points=1024, scenes={"Area_5_a.npy": synth_scene(31, 5000, 1.6, 1.2)})
    net = get_model(13).cuda().eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights_sd.items()})
    labels = ds.semantic_labels_list[0]
    origin = int(np.bincount(labels.astype(np.int64)).argmax())
    np.random.seed(5)
    torch.manual_seed(5)
    path = tmp_path / "tar.txt"
    seen = []

    def make(m, target, mask):
        seen.append(int(mask.sum()))
        return torchattacks.tar_NB_attack(m, eps=0.5, alpha=0.1, iters=3, target=target, mask=mask)

    res = harness.evaluate_whole_scene(net, ds, make, batch_size=2, log_path=str(path), log=lambda *_: None,
                                       targeted=dict(origin=origin, target=(origin + 1) % 13))
    rows = path.read_text().splitlines()
    assert rows[0] == "ori\tindex\tL2_dis\tcount\t target acc\tadv_acc\tacc\tadv_miou\tmiou"
    assert len(rows) - 1 == len(seen) and len(seen) >= 1
    for r in rows[1:]:
        f = r.split("\t")
        assert len(f) == 10 and int(f[0]) == origin and int(f[3]) > 0 and 0.0 <= float(f[4]) <= 1.0
    assert res["counters"][0][0].sum() == 5000
