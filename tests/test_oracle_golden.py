"""CPU tests: the oracle (oracle/) against the fixtures generated from the reference itself.

These pin the oracle: every integer output bit-exact, floating-point outputs within the stated
tolerance, and teacher-forced attack iterations bit-equal except where the reference's own unstable
sort broke an exact distance tie differently (tests/golden/make_golden.py; DESIGN.md "ties")."""
import numpy as np
import pytest

from oracle import attacks as oatk
from oracle import pn2


def test_square_distance_bits(golden_room):
    g = golden_room
    d = pn2.square_distance(g["sqd_src"], g["sqd_dst"])
    assert np.array_equal(d.view(np.uint32), g["sqd_bits"])


def test_geometry_bit_exact(golden_room, oracle_net):
    g = golden_room
    geom = oracle_net.geometry(g["room"][:, :3], g["starts"])
    for lvl in range(4):
        assert np.array_equal(geom["fps"][lvl], g["fps%d" % lvl].astype(np.int32))
        assert np.array_equal(geom["group"][lvl], g["group%d" % lvl].astype(np.int32))
        assert np.array_equal(geom["nn_idx"][lvl], g["nn_idx%d" % lvl].astype(np.int32))
        assert np.array_equal(geom["nn_w"][lvl].view(np.uint32), g["nn_w%d" % lvl].view(np.uint32))


def test_ball_query_known_answers():
    # first-K-in-index-order within r, padded with the first hit
    xyz = np.zeros((40, 3), np.float32)
    xyz[:, 0] = np.arange(40) * 0.012
    q = xyz[[0, 39]]
    idx = pn2.ball_query(0.1, 32, xyz, q)
    assert idx[0].tolist() == list(range(9)) + [0] * 23
    assert idx[1].tolist() == list(range(31, 40)) + [31] * 23
    idx = pn2.ball_query(1.0, 8, xyz, q)
    assert idx[0].tolist() == list(range(8)) and idx[1].tolist() == list(range(8))


def test_fps_ties_pick_lowest_index():
    xyz = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0]], np.float32)
    assert pn2.fps(xyz, 3, 0).tolist() == [0, 1, 2]


def test_forward_and_gradient(golden_room, oracle_net):
    g = golden_room
    geom = oracle_net.geometry(g["room"][:, :3], g["starts"])
    logp, cache = oracle_net.forward(g["room"], geom)
    assert np.abs(logp - g["logp"]).max() <= 1e-4
    for i, name in enumerate(("sa1", "sa2", "sa3", "sa4")):
        assert np.abs(cache["sa_out"][i + 1] - g["act_" + name]).max() <= 1e-4
    for lvl, name in ((3, "fp4"), (2, "fp3"), (1, "fp2")):
        assert np.abs(cache["fp_out"][lvl] - g["act_" + name]).max() <= 1e-4
    assert np.abs(cache["fp_out"][0][::16] - g["act_fp1"]).max() <= 1e-4
    dlogp, cost = pn2.nll_logp_grad(logp, g["labels"].astype(np.int64), 1.0 / 4096)
    assert abs(cost - float(g["cost"])) < 1e-5
    dc = oracle_net.backward_color(cache, dlogp)
    ref = g["dcolor"]
    nz = ref != 0
    assert np.array_equal(dc != 0, nz)
    assert (np.sign(dc[nz]) == np.sign(ref[nz])).mean() >= 0.999
    assert np.median(np.abs(dc - ref)[nz] / np.abs(ref[nz])) < 1e-4


def _step(orc, rooms, state, labels, starts_t, alpha, eps, last, mask=None, target=None):
    """One teacher-forced attack iteration from the reference's colour state."""
    images = np.ascontiguousarray(rooms.transpose(0, 2, 1)).copy()
    ori = images[:, 3:6].copy()
    images[:, 3:6] = state
    B, _, N = images.shape
    logp, caches = oatk.model_forward(orc, images, starts_t)
    grad = np.zeros((B, 3, N), np.float32)
    for b in range(B if target is None else 1):
        y = labels[b] if target is None else np.full(N, target, np.int64)
        dlogp, _ = pn2.nll_logp_grad(logp[b], y, 1.0 / N)
        grad[b] = orc.backward_color(caches[b], dlogp).T
    sgn = np.sign(grad) * (np.float32(-1.0) if target is not None else np.float32(1.0))
    stepped = state + np.float32(alpha) * sgn
    proj = np.clip(ori + np.clip(stepped - ori, -np.float32(eps), np.float32(eps)), 0, 1).astype(np.float32)
    out = stepped if last else proj
    if mask is not None:
        out = np.where(mask[None, None, :], out, state)
    return out


@pytest.mark.parametrize("t", [0, 5, 39])
def test_nb_attack_step(golden_nb, oracle_net, t):
    g = golden_nb
    iters = int(g["iters"])
    nxt = g["adv_color_final"] if t == iters - 1 else g["state_it%d" % (t + 1)]
    got = _step(oracle_net, g["rooms"], g["state_it%d" % t], g["labels"].astype(np.int64), g["starts"][1 + t],
                float(g["alpha"]), float(g["eps"]), last=(t == iters - 1))
    assert (got.view(np.uint32) == nxt.view(np.uint32)).mean() >= 0.999


@pytest.mark.parametrize("t", [0, 9])
def test_tar_nb_attack_step(golden_tarnb, oracle_net, t):
    g = golden_tarnb
    iters = int(g["iters"])
    nxt = g["adv_color_final"] if t == iters - 1 else g["state_it%d" % (t + 1)]
    got = _step(oracle_net, g["rooms"], g["state_it%d" % t], None, g["starts"][t], float(g["alpha"]), float(g["eps"]),
                last=(t == iters - 1), mask=g["mask"], target=int(g["target"]))
    assert (got.view(np.uint32) == nxt.view(np.uint32)).mean() >= 0.999
    assert np.array_equal(got[:, :, ~g["mask"]], g["rooms"].transpose(0, 2, 1)[:, 3:6][:, :, ~g["mask"]])


def test_nb_attack_loop_matches_steps(golden_nb, oracle_net):
    """oracle/attacks.py nb_attack (the loop used as CPU baseline and GPU checker) == chained steps."""
    g = golden_nb
    images = np.ascontiguousarray(g["rooms"].transpose(0, 2, 1))
    adv = oatk.nb_attack(oracle_net, images, g["labels"].astype(np.int64), float(g["eps"]), float(g["alpha"]), 2,
                         g["starts"][1:3])
    assert np.abs(adv[:, 3:6] - images[:, 3:6]).max() <= float(g["eps"]) + float(g["alpha"]) + 1e-6
    # iteration 1 of the loop must reproduce the reference state after 1 step where projected
    rec = {}
    oatk.nb_attack(oracle_net, images, g["labels"].astype(np.int64), float(g["eps"]), float(g["alpha"]), 1,
                   g["starts"][1:2], record=lambda it, a, c, gr, lp: rec.update(color=c.copy()))
    assert (rec["color"].view(np.uint32) == g["state_it1"].view(np.uint32)).mean() >= 0.999


def _nu_check(orc, g, t, nb, mask, target):
    images = np.ascontiguousarray(g["rooms"].transpose(0, 2, 1))
    w = g["s%d_w_before" % t]
    if t == 0:
        m, v = np.zeros_like(w), np.zeros_like(w)
    else:
        m, v = g["s%d_m" % (t - 1)], g["s%d_v" % (t - 1)]
    r = oatk.nu_step(orc, images, images, w, m, v, int(g["s%d_t" % t]), g["labels"].astype(np.int64), g["starts"][t],
                     float(g["c"]), float(g["kappa"]), float(g["s%d_lr" % t]), nb, mask=mask, target=target)
    ref = g["s%d_grad" % t]
    assert abs(r["cost"] - g["costs"][t]) <= 1e-4 * abs(g["costs"][t]) + 0.02
    assert np.median(np.abs(r["grad_w"] - ref) / (np.abs(ref) + 1e-12)) < 1e-3
    assert (np.abs(r["grad_w"] - ref) <= 1e-2 * np.abs(ref).max()).mean() >= 0.99
    assert (np.abs(r["w"] - g["s%d_w_after" % t]) <= 1e-4).mean() >= 0.995
    assert (np.abs(r["m"] - g["s%d_m" % t]) <= 1e-3 * np.abs(ref).max()).mean() >= 0.99


@pytest.mark.parametrize("t", [0, 1, 5])
def test_nu_attack_step(golden_nu, oracle_net, t):
    """NU_attack (nontarget.py:52-135): f-loss + c*Smooth + c*L2, tanh space, torch Adam -- one
    teacher-forced optimiser step from the state recorded inside the reference."""
    _nu_check(oracle_net, golden_nu, t, 10, None, None)


@pytest.mark.parametrize("t", [0, 20, 30])
def test_tar_nu_attack_step(golden_tarnu, oracle_net, t):
    """tar_NU_attack (target.py:62-175), masked colours, lr halved + fresh Adam after step 50 not reached;
    steps 20/30 sit on the restart check boundary."""
    g = golden_tarnu
    _nu_check(oracle_net, g, t, 5, g["mask"], int(g["target"]))


def test_tar_nu_restart_clamps_xyz(golden_tarnu):
    """Known answer from the recorded reference run: the restart after step 40 (cost[40] >= cost[30])
    clamps ALL channels to [0,1]; the L2 term then carries sum((clamp(xyz)-xyz)^2) (cost jump)."""
    g = golden_tarnu
    c = g["costs"]
    assert c[40] >= c[30] and c[20] < c[10] and c[30] < c[20]
    x = g["rooms"][0]
    moved = ((np.clip(x, 0, 1) - x) ** 2).sum()
    assert abs((c[41] - c[39]) - float(g["c"]) * moved) <= 0.03 * moved  # + the f/Smooth change of the moved geometry
