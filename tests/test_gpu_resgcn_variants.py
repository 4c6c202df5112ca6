"""GPU parity of the ResGCN configuration switches: block = plain / dense / res with
conv = edge / mr, through the C ABI, against fixtures generated from the reference's DenseDeepGCN
(tests/golden/gcn_variants.npz) and against the CPU oracle.  With the reference's graphs teacher-forced: last block
output and logits within 2e-4 (relative to the tensor's magnitude), cost within 1e-4, input gradient equal to rounding
except for max-tie flips (see tests/test_oracle_resgcn_variants.py: check_dx).  Free-running: every dynamic graph is
exactly the oracle's kNN of the features the GPU itself produced (64-wide inputs) or >= 99.9 % of it (dense blocks:
128 / 192-wide distance products accumulate in another order)."""
import os

import numpy as np
import pytest
import torch

from pointsecguard_amd.synthetic import gcn_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
VARIANTS = (("plain", "edge"), ("dense", "edge"), ("res", "mr"), ("plain", "mr"), ("dense", "mr"))


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda().contiguous()


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(os.path.join(GOLD, "gcn_variants.npz")))


def build(fx, block, conv):
    from pointsecguard_amd import runtime
    from pointsecguard_amd.resgcn.sem_seg_dense.architecture import BLOCKS, CONVS
    nb = int(fx["n_blocks"])
    sd = gcn_state_dict(int(fx["seed"]), nb, block, conv)
    model = runtime.GCNModel(sd, nb, block=BLOCKS[block], conv=CONVS[conv])
    ws = runtime.GCNWorkspace(1, 1024, nb, block=BLOCKS[block], conv=CONVS[conv])
    return sd, model, ws, nb


def check_dx(dx, ref):
    scale = np.abs(ref).max()
    off = np.abs(dx - ref) > 1e-4 * scale
    assert off.mean() < 0.005, off.mean()
    assert np.abs(dx - ref).max() <= 0.02 * scale
    assert np.abs(dx - ref)[~off].max() <= 1e-4 * scale


@pytest.mark.parametrize("block,conv", VARIANTS)
def test_variant_with_reference_graphs(fx, block, conv):
    from pointsecguard_amd import _lib, runtime
    _, model, ws, nb = build(fx, block, conv)
    tag = "%s_%s_" % (block, conv)
    ws.set_graphs(dev(np.stack([fx[tag + "nbr%d" % e].astype(np.int32)[None] for e in range(nb)])))
    x0 = dev(fx["room"][None])
    logits = ws.forward(model, x0)
    last = ws.feats()[0].cpu().numpy()[:, -64:]
    ref = fx[tag + "logits"]
    assert np.abs(logits[0].cpu().numpy() - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(last - fx[tag + "last"]).max() <= 2e-4 * max(1.0, np.abs(fx[tag + "last"]).max())
    labels = dev(fx["labels"].astype(np.int32)[None])
    dl = torch.empty_like(logits)
    cost = torch.zeros(1, device="cuda")
    _lib.call("psg_ce_logp_grad", runtime.ptr(logits), runtime.ptr(labels), 0, 1024, 1024, 13, 1.0 / 1024,
              runtime.ptr(dl), runtime.ptr(cost), runtime.stream())
    dx = ws.backward(model, dl)[0].cpu().numpy()
    ws.set_graphs(None)
    assert abs(cost.item() - float(fx[tag + "cost"])) <= 1e-4 * max(1.0, float(fx[tag + "cost"]))
    check_dx(dx, fx[tag + "dx"])


@pytest.mark.parametrize("block,conv", VARIANTS)
def test_variant_free_running_vs_oracle(fx, block, conv):
    """Dynamic graphs in situ + the whole forward / backward against the oracle run on the GPU's own graphs."""
    from oracle import resgcn
    from pointsecguard_amd import _lib, runtime
    sd, model, ws, nb = build(fx, block, conv)
    x0 = dev(fx["room"][None])
    logits = ws.forward(model, x0)
    torch.cuda.synchronize()
    feats = ws.feats()[0].cpu().numpy()
    graphs = [ws.edges(e)[0].cpu().numpy() for e in range(nb)]
    for e in range(nb):
        if e == 0:
            src, d = fx["room"][:, :3], 1
        elif block == "dense":
            src, d = np.ascontiguousarray(feats[:, :64 * e]), e
        else:
            src, d = np.ascontiguousarray(feats[:, 64 * (e - 1):64 * e]), (1 if block == "plain" else e)
        want = resgcn.knn_dilated(src, d)
        same = (graphs[e] == want).mean()
        assert same == 1.0 if src.shape[1] <= 64 else same >= 0.999, (e, same)
    orc = resgcn.GCNOracle(sd, nb, block=block, conv=conv)
    lo, cache = orc.forward(fx["room"], graphs=graphs)
    assert np.abs(logits[0].cpu().numpy() - lo).max() <= 2e-4 * max(1.0, np.abs(lo).max())
    # the oracle's `feats` is the reference's concatenation; its last 64 * n_blocks columns are y_0 .. y_{n-1} for a
    # dense backbone (= the last block's output) and cur_0 .. cur_{n-1} otherwise: the workspace layout either way
    assert np.abs(feats - cache["feats"][:, -feats.shape[1]:]).max() <= 2e-4 * max(1.0, np.abs(feats).max())
    dl_np, _ = resgcn.ce_mean_grad(lo, fx["labels"].astype(np.int64))
    dx = ws.backward(model, dev(dl_np[None]))[0].cpu().numpy()
    check_dx(dx, orc.backward(cache, dl_np))


def test_variant_module_api_and_attack(fx):
    """DenseDeepGCN(opt) with block='dense', conv='mr' behind the reference's constructor: state_dict layout, forward
    equal to the C-ABI path, and the NB attack loop (hipGraph replay included) stays inside its eps ball."""
    from types import SimpleNamespace
    from pointsecguard_amd.resgcn.sem_seg_dense.architecture import DenseDeepGCN
    from pointsecguard_amd.resgcn.sem_seg_dense.attacks import torchattacks
    nb = int(fx["n_blocks"])
    opt = SimpleNamespace(n_filters=64, k=16, act="relu", norm="batch", bias=True, epsilon=0.0, stochastic=True,
                          conv="mr", n_blocks=nb, block="dense", in_channels=9, dropout=0.0, n_classes=13)
    net = DenseDeepGCN(opt).cuda().eval()
    sd = gcn_state_dict(int(fx["seed"]), nb, "dense", "mr")
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = dev(np.ascontiguousarray(fx["room"].T)[None, :, :, None])
    out = net(x)
    assert tuple(out.shape) == (1, 13, 1024)
    ref = fx["dense_mr_logits"]
    # free-running graphs: near-tie flips allowed, so compare predictions statistically
    assert np.abs(out[0].T.cpu().numpy() - ref).mean() <= 1e-2 * np.abs(ref).mean()
    atk = torchattacks.NB_attack(net, eps=0.1, alpha=0.02, iters=6)
    adv = atk(x, dev(fx["labels"].astype(np.int64)[None]))
    torch.cuda.synchronize()
    a, o = adv.cpu().numpy(), x.cpu().numpy()
    assert np.array_equal(a[:, :3], o[:, :3]) and np.array_equal(a[:, 6:], o[:, 6:])
    assert np.abs(a[:, 3:6] - o[:, 3:6]).max() <= 0.1 + 0.02 + 1e-6
    assert np.abs(a[:, 3:6] - o[:, 3:6]).max() > 0.02


def test_random_noise_baseline(fx, tmp_path):
    """The `random` attack of the reference's test.py: noise of L2 norm 1 on the colours; metrics from the device
    counters equal a numpy recomputation from the same predictions; log lines in the reference's format."""
    from types import SimpleNamespace
    from pointsecguard_amd.resgcn.sem_seg_dense.architecture import DenseDeepGCN
    from pointsecguard_amd.resgcn.sem_seg_dense.test import random_noise
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    nb = int(fx["n_blocks"])
    opt = SimpleNamespace(n_filters=64, k=16, act="relu", norm="batch", bias=True, epsilon=0.0, stochastic=True,
                          conv="edge", n_blocks=nb, block="plain", in_channels=9, dropout=0.0, n_classes=13,
                          device="cuda", res_dir=str(tmp_path))
    net = DenseDeepGCN(opt).cuda()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gcn_state_dict(5, nb, "plain", "edge").items()})
    rooms = make_rooms(3, 41)
    loader = [SimpleNamespace(pos=torch.from_numpy(r[None, :, :3].copy()), x=torch.from_numpy(r[None, :, 3:].copy()),
                              y=torch.from_numpy(rule_labels(r[None]))) for r in rooms]
    torch.manual_seed(0)
    res = random_noise(net, loader, opt)
    assert np.allclose(res["dis"], 1.0, atol=1e-4)          # |noise|_2 = 1 by construction
    assert np.all((res["acc"] >= 0) & (res["acc"] <= 1)) and np.all(res["Us"].sum(1) >= 4096)
    assert np.allclose(res["mious"], res["Is"].sum(1) / res["Us"].sum(1))
    lines = open(res["log"]).read().splitlines()
    assert lines[0].startswith("index\tl2dis\tadv_acc\tacc") and len(lines) == 4
    assert lines[1].split("\t")[0] == "0" and len(lines[1].split("\t")) == 7


def test_attack_experiment_loops(fx, tmp_path):
    """The four experiment loops of the reference's sem_seg_dense/attacks.py through the test.py dispatcher: protocol
    (skip rules of the targeted loops), log format, and metrics equal to a numpy recomputation."""
    from types import SimpleNamespace
    from pointsecguard_amd.resgcn.sem_seg_dense.architecture import DenseDeepGCN
    from pointsecguard_amd.resgcn.sem_seg_dense.test import attack
    from pointsecguard_amd.synthetic import make_rooms, rule_labels
    nb = int(fx["n_blocks"])
    opt = SimpleNamespace(n_filters=64, k=16, act="relu", norm="batch", bias=True, epsilon=0.0, stochastic=True,
                          conv="edge", n_blocks=nb, block="res", in_channels=9, dropout=0.0, n_classes=13,
                          device="cuda", res_dir=str(tmp_path), target=1, origin=0, left_ratio=1.0, att_type="Color")
    net = DenseDeepGCN(opt).cuda()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gcn_state_dict(5, nb, "res", "edge").items()})
    net.eval()
    rooms = make_rooms(2, 43)
    loader = []
    for r in rooms:
        x = dev(np.ascontiguousarray(r.T)[None, :, :, None])
        y = net(x).argmax(1)                                # labels = the clean predictions: clean accuracy 1
        loader.append(SimpleNamespace(pos=torch.from_numpy(r[None, :, :3].copy()), x=torch.from_numpy(r[None, :, 3:].copy()),
                                      y=y.cpu()))
    opt.attack, opt.attack_kwargs = "NB_attack", dict(iters=4)
    res = attack(net, loader, opt)
    assert np.allclose(res["acc"], 1.0) and np.all(res["other_acc"] <= 1.0) and np.all(res["dis"] > 0)
    lines = open(res["log"]).read().splitlines()
    assert lines[0] == "index\tL2_dis\tother_acc\tacc\tadv_miou\tmiou" and len(lines) == 3
    opt.attack, opt.attack_kwargs = "NU_attack", dict(steps=3)
    res = attack(net, loader, opt)
    assert np.allclose(res["acc"], 1.0) and len(open(res["log"]).read().splitlines()) == 3
    # targeted: attack the most frequent predicted class of room 0; a room with <= 500 such points is skipped
    cls, cnt = torch.unique(loader[0].y, return_counts=True)
    opt.origin = int(cls[cnt.argmax()])
    opt.target = (opt.origin + 1) % 13
    for name, kw in (("tar_NB_attack", dict(iters=3)), ("tar_NU_attack", dict(steps=3))):
        opt.attack, opt.attack_kwargs = name, kw
        res = attack(net, loader, opt)
        lines = open(res["log"]).read().splitlines()
        assert lines[0] == "left_ratio=1.0" and lines[1].startswith("index\tcount\tL2_dis\ttarget_acc")
        assert len(lines) == 2 + int((~res["skipped"]).sum())
        for i in np.where(~res["skipped"])[0]:
            assert 0.0 <= res["target_acc"][i] <= 1.0 and res["dis"][i] > 0
    opt.attack = "bogus"
    with pytest.raises(ValueError):
        attack(net, loader, opt)
