"""GPU tests of the drop-in Python surface (pointsecguard_amd.models / .attacks.torchattacks) used the way
the reference harness uses it (PointNet/NB_nontarget_test_semseg.py:100-106,163-173)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net(weights_sd):
    from pointsecguard_amd.models.pointnet2_sem_seg import get_model
    m = get_model(13)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in weights_sd.items()})
    assert not missing.missing_keys and not missing.unexpected_keys   # reference checkpoints load as-is
    return m.cuda().eval()


def test_harness_sequence_matches_reference_stream(net, golden_nb):
    """torch.manual_seed -> clean forward -> NB_attack: the FPS draws come from the CPU generator in the
    reference's order, so the clean log-probs and the first attack iterations match the golden run."""
    from pointsecguard_amd.attacks import torchattacks
    g = golden_nb
    x = torch.from_numpy(np.ascontiguousarray(g["rooms"].transpose(0, 2, 1))).cuda()
    torch.manual_seed(int(g["seed_rng"]))
    logp, l4 = net(x)
    assert logp.shape == (2, 4096, 13) and l4.shape == (2, 512, 16)
    assert np.abs(logp.detach().cpu().numpy() - g["clean_logp"]).max() <= 1e-4
    atk = torchattacks.NB_attack(net, eps=float(g["eps"]), alpha=float(g["alpha"]), iters=2)
    adv = atk(x, g["labels"].astype(np.float64))           # labels as float64 numpy, like the harness
    assert adv.shape == x.shape and not adv.requires_grad
    ori = x[:, 3:6].cpu().numpy()
    got = adv[:, 3:6].cpu().numpy()
    proj = np.clip(ori + np.clip(got - ori, -np.float32(g["eps"]), np.float32(g["eps"])), 0, 1).astype(np.float32)
    assert (proj.view(np.uint32) == g["state_it2"].view(np.uint32)).mean() >= 0.999
    assert str(atk).startswith("NB_attack(")
    assert net.training is False


def test_autograd_colour_leaf(net, golden_room):
    """A leaf on the colour channels gets its .grad from the HIP backward (nontarget.py:29-35 pattern)."""
    g = golden_room
    x = torch.from_numpy(np.ascontiguousarray(g["room"].T[None])).cuda()
    color = x[:, 3:6].clone().requires_grad_(True)
    adv = x.clone()
    adv[:, 3:6] = color
    torch.manual_seed(0)
    outputs, _ = net(adv)
    y = torch.from_numpy(g["labels"].astype(np.int64)).cuda()
    cost = torch.nn.CrossEntropyLoss(reduction="sum")(outputs.reshape(-1, 13), y.view(-1)) / outputs.size(1)
    net.zero_grad()
    cost.backward()
    assert abs(cost.item() - float(g["cost"])) < 1e-4
    got = color.grad[0].T.cpu().numpy()
    ref = g["dcolor"]
    nz = ref != 0
    assert np.array_equal(got != 0, nz)
    assert (np.sign(got[nz]) == np.sign(ref[nz])).mean() >= 0.999


def test_tar_nb_attack_api(net, golden_tarnb):
    from pointsecguard_amd.attacks import torchattacks
    g = golden_tarnb
    x = torch.from_numpy(np.ascontiguousarray(g["rooms"].transpose(0, 2, 1))).cuda()
    torch.manual_seed(int(g["seed_rng"]))
    atk = torchattacks.tar_NB_attack(net, eps=float(g["eps"]), alpha=float(g["alpha"]), iters=2,
                                     target=int(g["target"]), mask=g["mask"])
    adv = atk(x, g["labels"].astype(np.float64)).cpu().numpy()
    ori = x[:, 3:6].cpu().numpy()
    proj = np.clip(ori + np.clip(adv[:, 3:6] - ori, -np.float32(g["eps"]), np.float32(g["eps"])), 0, 1).astype(np.float32)
    assert (proj.view(np.uint32) == g["state_it2"].view(np.uint32)).mean() >= 0.999


def test_errors_are_loud(net):
    from pointsecguard_amd import _lib
    with pytest.raises(_lib.PsgError):
        net(torch.zeros(1, 9, 4096))                      # CPU tensor: no fallback
    net.train()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 9, 4096, device="cuda"))
    net.eval()
    with pytest.raises(ValueError):
        net(torch.zeros(1, 6, 4096, device="cuda"))


def test_pointnet_util_functions(golden_room):
    """Public helpers of models/pointnet_util.py on CUDA tensors, against the reference's outputs."""
    from pointsecguard_amd.models import pointnet_util as pu
    g = golden_room
    xyz = torch.from_numpy(g["room"][None, :, :3].copy()).cuda()
    torch.manual_seed(0)
    fi = pu.farthest_point_sample(xyz, 1024)
    assert fi.dtype == torch.int64 and np.array_equal(fi[0].cpu().numpy(), g["fps0"].astype(np.int64))
    new_xyz = pu.index_points(xyz, fi)
    gi = pu.query_ball_point(0.1, 32, xyz, new_xyz)
    assert np.array_equal(gi[0].cpu().numpy(), g["group0"].astype(np.int64))
    d = pu.square_distance(torch.from_numpy(g["sqd_src"][None]).cuda(), torch.from_numpy(g["sqd_dst"][None]).cuda())
    assert np.array_equal(d[0].cpu().numpy().view(np.uint32), g["sqd_bits"])
    torch.manual_seed(0)
    nx, npnts = pu.sample_and_group(1024, 0.1, 32, xyz, torch.from_numpy(g["room"][None]).cuda())
    assert nx.shape == (1, 1024, 3) and npnts.shape == (1, 1024, 32, 12)
    assert torch.equal(npnts[0, :, 0, 3:], torch.from_numpy(g["room"]).cuda()[gi[0, :, 0]])


def test_sample_and_group_full(golden_room):
    """sample_and_group (pointnet_util.py:110-143) element for element: [grouped_xyz - new_xyz, grouped features] for every
    (centroid, sample), built in numpy from the reference's own FPS / ball-query tables."""
    from pointsecguard_amd.models import pointnet_util as pu
    g = golden_room
    room = g["room"]
    xyz = torch.from_numpy(room[None, :, :3].copy()).cuda()
    torch.manual_seed(0)
    nx, npnts, gxyz, fidx = pu.sample_and_group(1024, 0.1, 32, xyz, torch.from_numpy(room[None]).cuda(), returnfps=True)
    fps, grp = g["fps0"].astype(np.int64), g["group0"].astype(np.int64)
    assert np.array_equal(fidx[0].cpu().numpy(), fps)
    assert np.array_equal(nx[0].cpu().numpy(), room[fps, :3])
    assert np.array_equal(gxyz[0].cpu().numpy(), room[grp][:, :, :3])
    expect = np.concatenate([room[grp][:, :, :3] - room[fps, :3][:, None, :], room[grp]], axis=-1)
    assert np.array_equal(npnts[0].cpu().numpy(), expect)


def test_gcn_lib_helpers(golden_gcn_room):
    """batched_index_select (torch_nn.py:82-98), DenseDilated (torch_edge.py:19-29) and dense_knn_matrix (:45-59) called
    directly, like the reference's own modules call them."""
    from oracle import resgcn
    from pointsecguard_amd.resgcn.gcn_lib.dense.torch_edge import DenseDilated, dense_knn_matrix
    from pointsecguard_amd.resgcn.gcn_lib.dense.torch_nn import batched_index_select
    g = golden_gcn_room
    f = torch.from_numpy(g["feat0"]).cuda()                               # [1024, 64]
    x = f.t()[None, :, :, None].contiguous()                              # [1, 64, 1024, 1]
    edge, d = dense_knn_matrix(x, 16 * 3)
    assert d == 3 and edge.shape == (2, 1, 1024, 16) and edge.dtype == torch.int64
    assert np.array_equal(edge[0, 0].cpu().numpy(), resgcn.knn_dilated(g["feat0"], 3))
    assert torch.equal(edge[1, 0], torch.arange(1024, device="cuda")[:, None].expand(1024, 16))
    sel = batched_index_select(x, edge[0])
    assert sel.shape == (1, 64, 1024, 16)
    assert torch.equal(sel[0], f[edge[0, 0]].permute(2, 0, 1))            # x[:, :, idx]
    full = torch.arange(2 * 1 * 8 * 48).view(2, 1, 8, 48)
    dd = DenseDilated(16, 3, stochastic=True, epsilon=0.0).eval()
    torch.manual_seed(1)
    before = torch.rand(1)
    torch.manual_seed(1)
    out = dd(full)
    assert torch.equal(out, full[:, :, :, ::3]) and torch.equal(torch.rand(1), torch.rand(1)) is False
    torch.manual_seed(1)
    torch.rand(1)
    after_ref = torch.rand(1)
    torch.manual_seed(1)
    dd(full)
    assert torch.equal(torch.rand(1), after_ref)                          # exactly one CPU-generator draw, like torch_edge.py:21


def test_global_max():
    """psg_global_max = torch.max_pool2d(x, [N, 1]) of DenseDeepGCN.forward (architecture.py:64) on point-major rows: value and
    first arg-max row per room and channel, incl. ties, negatives and N not a multiple of the row chunk."""
    from pointsecguard_amd import ops
    rng = np.random.default_rng(4)
    for B, N, C in ((1, 4096, 1024), (3, 1000, 64), (2, 65, 128)):
        x = rng.standard_normal((B, N, C)).astype(np.float32)
        x[:, :, 0] = -np.abs(x[:, :, 0]) - 1.0                 # an all-negative channel
        x[:, N // 3, 1] = 9.0
        x[:, N // 2, 1] = 9.0                                  # a tie: the lower row wins
        mx, arg = ops.global_max(torch.from_numpy(x).cuda())
        assert np.array_equal(mx.cpu().numpy(), x.max(axis=1))
        assert np.array_equal(arg.cpu().numpy(), x.argmax(axis=1).astype(np.int32))


def test_nb_attack_whole_attack_graph_equals_eager():
    """Round 5 (opt-in, PSG_PN2_GRAPH=1: measured slower than the eager launches): psg_pn2_nb_attack on a
    batch of at most 16 rooms replays the whole attack (plan + iterations) as a hipGraph from its third call with the same
    settings on (first call eager, second captured on the workspace's own stream - the harness calls from the legacy default
    stream, which cannot capture).  Same bits as the eager launches, on the default stream and on a side stream; the
    bookkeeping counts one capture, no failure.  (The switch is read once per process: a child interpreter.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from pointsecguard_amd import _lib, runtime
from pointsecguard_amd.synthetic import make_rooms, rule_labels
sd = dict(np.load(os.path.join(%r, "tests", "golden", "pn2_weights.npz")))
model = runtime.PN2Model(runtime.fold_state_dict(sd))
B, iters = 2, 3
rng = np.random.default_rng(5)
starts = torch.from_numpy(np.stack([rng.integers(0, n, (iters, B)) for n in (4096, 1024, 256, 64)], axis=1).astype(np.int32)).cuda()
rooms = [make_rooms(B, 300 + i) for i in range(4)]
imgs = [torch.from_numpy(np.ascontiguousarray(r.transpose(0, 2, 1))).cuda() for r in rooms]
labs = [torch.from_numpy(rule_labels(r).astype(np.int32)).cuda() for r in rooms]
before = _lib.capture_stats()
ws = runtime.PN2Workspace(B, 4096, iters)
outs = [ws.nb_attack(model, imgs[i], labs[i], starts, 0.05, 2 / 255, iters).cpu().numpy() for i in range(4)]   # eager, capture, replay, replay
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    out_side = ws.nb_attack(model, imgs[3], labs[3], starts, 0.05, 2 / 255, iters)
side.synchronize()
after = _lib.capture_stats()
for i in (0, 3):                                                 # a fresh workspace's first call is eager: the reference bits
    eager_out = runtime.PN2Workspace(B, 4096, iters).nb_attack(model, imgs[i], labs[i], starts, 0.05, 2 / 255, iters).cpu().numpy()
    assert np.array_equal(outs[i].view(np.uint32), eager_out.view(np.uint32)), i
    if i == 3:
        assert np.array_equal(out_side.cpu().numpy().view(np.uint32), eager_out.view(np.uint32))
d = {k: after[k] - before[k] for k in after}
assert d["captures_tried"] == 1 and d["captures_failed"] == 0 and d["replays"] == 4 and d["eager"] == 1, d
print("graph == eager", d)
""" % (root, root)
    env = dict(os.environ)
    env["PSG_PN2_GRAPH"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "graph == eager" in r.stdout, r.stderr[-3000:]
