import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights_sd():
    return dict(np.load(os.path.join(GOLDEN, "pn2_weights.npz")))


@pytest.fixture(scope="session")
def golden_room():
    return dict(np.load(os.path.join(GOLDEN, "pn2_room.npz")))


@pytest.fixture(scope="session")
def golden_nb():
    return dict(np.load(os.path.join(GOLDEN, "pn2_nb.npz")))


@pytest.fixture(scope="session")
def golden_tarnb():
    return dict(np.load(os.path.join(GOLDEN, "pn2_tarnb.npz")))


@pytest.fixture(scope="session")
def oracle_net(weights_sd):
    from oracle import pn2
    return pn2.PN2Oracle(weights_sd)


@pytest.fixture(scope="session")
def gpu_model(weights_sd):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from pointsecguard_amd import runtime
    return runtime.PN2Model(runtime.fold_state_dict(weights_sd))


@pytest.fixture(scope="session")
def golden_nu():
    return dict(np.load(os.path.join(GOLDEN, "pn2_nu.npz")))


@pytest.fixture(scope="session")
def golden_tarnu():
    return dict(np.load(os.path.join(GOLDEN, "pn2_tarnu.npz")))


@pytest.fixture(scope="session")
def gcn_weights_sd():
    return dict(np.load(os.path.join(GOLDEN, "gcn_weights.npz")))


@pytest.fixture(scope="session")
def golden_gcn_room():
    return dict(np.load(os.path.join(GOLDEN, "gcn_room.npz")))


@pytest.fixture(scope="session")
def golden_gcn_nb():
    return dict(np.load(os.path.join(GOLDEN, "gcn_nb.npz")))


@pytest.fixture(scope="session")
def gcn_oracle(gcn_weights_sd):
    from oracle import resgcn
    return resgcn.GCNOracle(gcn_weights_sd, n_blocks=5)


@pytest.fixture(scope="session")
def golden_gcn_nu():
    return dict(np.load(os.path.join(GOLDEN, "gcn_nu.npz")))


@pytest.fixture(scope="session")
def golden_gcn_tarnu():
    return dict(np.load(os.path.join(GOLDEN, "gcn_tarnu.npz")))
