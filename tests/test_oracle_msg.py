"""CPU tests: the MSG oracle (oracle/pn2_msg.py) against fixtures generated from the reference's own
pointnet2_sem_seg_msg network (tests/golden/make_golden_msg.py).  Integer outputs bit-exact; floating point within
the tolerance written at each assert."""
import os

import numpy as np
import pytest

from oracle import pn2, pn2_msg
from pointsecguard_amd.synthetic import msg_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def room():
    return dict(np.load(os.path.join(GOLD, "pn2msg_room.npz")))


@pytest.fixture(scope="module")
def net(room):
    return pn2_msg.PN2MsgOracle(msg_state_dict(int(room["msg_seed"])))


def test_msg_geometry_bit_exact(room, net):
    geom = net.geometry(room["room"][:, :3], room["starts"])
    for lvl in range(4):
        assert np.array_equal(geom["fps"][lvl], room["fps%d" % lvl].astype(np.int32))
        for i in range(2):
            assert np.array_equal(geom["group"][lvl][i], room["group%d_%d" % (lvl, i)].astype(np.int32))


def test_msg_forward_and_gradient(room, net):
    geom = net.geometry(room["room"][:, :3], room["starts"])
    logp, cache = net.forward(room["room"], geom)
    assert np.abs(logp - room["logp"]).max() <= 1e-4
    for i, name in enumerate(("sa1", "sa2", "sa3", "sa4")):
        ref = room["act_" + name]
        assert cache["sa_out"][i + 1].shape == ref.shape
        assert np.abs(cache["sa_out"][i + 1] - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    for lvl, name in ((3, "fp4"), (2, "fp3"), (1, "fp2")):
        ref = room["act_" + name]
        assert np.abs(cache["fp_out"][lvl] - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(cache["sa_out"][4] - room["l4"].T).max() <= 1e-4 * max(1.0, np.abs(room["l4"]).max())
    dlogp, cost = pn2.nll_logp_grad(logp, room["labels"].astype(np.int64), 1.0 / 4096)
    assert abs(cost - float(room["cost"])) < 1e-5
    dc = net.backward_color(cache, dlogp)
    ref = room["dcolor"]
    assert np.array_equal(dc != 0, ref != 0) or ((dc != 0) != (ref != 0)).mean() < 1e-3
    assert np.abs(dc - ref).max() <= 2e-3 * np.abs(ref).max()


def test_msg_module_has_the_reference_state_dict_layout(room):
    """models.pointnet2_sem_seg_msg.get_model accepts (strict) the state_dict the reference's get_model accepted when
    the fixture was generated: same keys and shapes, so reference checkpoints load unchanged.  (Construction only;
    its forward needs the GPU.)"""
    import torch
    from pointsecguard_amd.models import pointnet2_sem_seg_msg as msg
    net = msg.get_model(13)
    sd = msg_state_dict(int(room["msg_seed"]))
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    with pytest.raises(NotImplementedError):
        net.train()(torch.zeros(1, 9, 1024))
